def compute_J(dataset, gamma=1.0):
    raise NotImplementedError("stub")
