import copy


class MDPInfo:
    def __init__(self, observation_space, action_space, gamma, horizon):
        self.observation_space = observation_space
        self.action_space = action_space
        self.gamma = gamma
        self.horizon = horizon

    def copy(self):
        return copy.deepcopy(self)


class Environment:
    def __init__(self, mdp_info):
        self._mdp_info = mdp_info

    @property
    def info(self):
        return self._mdp_info

    def seed(self, seed):
        pass

    def stop(self):
        pass


class Core:          # names only need to exist (circle_terminated.py imports them)
    pass


class Agent:
    pass
