"""CPU-baseline legs of bench.py -- ORACLE CODE, test / measurement
infrastructure only: imported by bench.py's `cpu_baseline` leg and nothing else.  Kept free of torch so that the
all-cores leg's worker processes (multiprocessing 'spawn') start in about a second each."""
import os
import time

import numpy as np

IIWA_INIT_Q = np.array([0.0, 0.7135214629060707, 0.0, -0.5024756033561426, 0.0, 1.9256631778550268])


def _spec(env_name):
    from . import atacom_scalar as osc
    return {'circle': osc.circle_spec, 'planar': osc.planar_spec, 'iiwa': osc.iiwa_spec}[env_name]()


def scalar_leg(args):
    """R1: the float64 oracle in the reference's algorithmic shape -- one LAPACK SVD + one RREF per environment per
    physics sub-step (atacom.py:123-139), one environment at a time -- for `budget_s` seconds on ONE core.
    args = (env_name, budget_s, seed).  Returns {'steps', 'seconds'}."""
    env_name, budget_s, seed = args
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[v] = '1'
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:  # noqa: BLE001
        pass
    from . import atacom_scalar as osc
    spec = _spec(env_name)
    rng = np.random.default_rng(seed)
    env = osc.ScalarAtacomEnv(spec, init_q=IIWA_INIT_Q if env_name == 'iiwa' else None)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(8):
            env.step(rng.uniform(-1, 1, spec.n_null))
            n += 1
            if env.t >= spec.horizon:
                env.reset()
    return {'steps': n, 'seconds': time.perf_counter() - t0}


def batched_leg(env_name, init_rows, actions):
    """R2: the batch-vectorised float64 restatement, free-running over actions [T, n, k] from the given initial states
    (rows [n, 2 nq + 6] = [q, dq, puck] as handed to the engine's reset, or None for the default reset state), with
    auto-reset at `last` like the engine.  Returns the constraint statistics (atacom.py:201-216) of the run."""
    from . import atacom_batched as ob
    spec = _spec(env_name)
    T, n, _ = actions.shape
    nq = spec.dim_q
    if init_rows is None:
        env = ob.BatchedAtacomEnv(spec, n, init_q=IIWA_INIT_Q if env_name == 'iiwa' else None)
    else:
        r = np.asarray(init_rows, dtype=np.float64)
        puck = r[:, 2 * nq:2 * nq + 6] if r.shape[1] >= 2 * nq + 6 else None
        env = ob.BatchedAtacomEnv(spec, n, init_q=r[:, :nq], init_dq=r[:, nq:2 * nq], init_puck=puck)
    for t in range(T):
        _, _, ab, _ = env.step(actions[t])
        last = ab | (env.t >= spec.horizon)
        if last.any():
            env.reset(last)
    c_avg, c_max, c_dq = env.get_constraints_logs()
    return {'c_avg': c_avg, 'c_max': c_max, 'c_dq_max': c_dq}
