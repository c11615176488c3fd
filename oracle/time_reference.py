#!/usr/bin/env python3
"""R0 -- time the ACTUAL reference on this container's host cores (build container only:
/root/reference does not exist on the GPU box, so this never runs there and bench.py never calls it).

    python oracle/time_reference.py [--seconds 10] [--procs 8]

What is timed (env-steps/s; 1 env-step = one `step()` of one environment):
  circle   the reference's CircleEnvAtacom end to end (circle_atacom.py / circle_base.py / atacom.py), unchanged,
           through the duck-typed MushroomRL stub of oracle/_mushroom_stub;
  planar   the reference's generic AtacomEnvWrapper (atacom.py:106-216), ViabilityConstraint / ConstraintsSet,
  iiwa     pinv_null and rref, unchanged, at the 6 x 9 / 12 x 17 shapes with 4 sub-steps per step -- driven by THIS
           build's numpy kinematics callables and kinematic base env (oracle/gen_golden.py: _GenericAtacom), because
           the reference's own planar / iiwa environments need Pinocchio, PyBullet and MushroomRL.  It is therefore an
           UPPER bound for the real reference per core (which adds ~10 Pinocchio passes, Bullet inverse dynamics and
           stepSimulation per sub-step).
           Two variants: `*_cached` memoises this build's (slow, pure-numpy) kinematics per distinct (q, dq), which is
           the closest stand-in for Pinocchio's microsecond-scale C++ calls -- the time left is the reference's own
           Python / numpy / LAPACK work; the plain variant pays the numpy kinematics on each of the ~30 callable
           evaluations per sub-step.
Each on 1 core and on `--procs` independent processes.  Prints one JSON object; BASELINE.md section 2 quotes it."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.dont_write_bytecode = True                   # /root/reference is read-only for this build: no __pycache__ into it
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(args):
    name, seconds, seed = args
    cached = name.endswith('_cached')
    name = name.replace('_cached', '')
    for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[v] = '1'
    import numpy as np
    import gen_golden as gg                      # imports the reference + stub, our robots / specs
    rng = np.random.default_rng(seed)
    if name == 'circle':
        env = gg.CircleEnvAtacom(horizon=500, random_init=False)
        k, horizon = 1, 500
    else:
        spec = {'planar': gg.osc.planar_spec, 'iiwa': gg.osc.iiwa_spec}[name]()
        init_q = gg.robots.PLANAR_INIT_Q if name == 'planar' else gg.iiwa_init_q()
        if cached:
            raw, memo = gg.osc.constraint_terms, {}

            def memoised(sp, q, dq):
                key = (np.asarray(q).tobytes(), np.asarray(dq).tobytes())
                if key not in memo:
                    if len(memo) > 64:
                        memo.clear()
                    memo[key] = raw(sp, q, dq)
                return memo[key]
            gg.osc.constraint_terms = memoised
        env = gg._GenericAtacom(spec, init_q)
        k, horizon = spec.n_null, spec.horizon
    env.reset()
    n, t_ep = 0, 0
    for _ in range(20):                           # warm-up
        env.step(rng.uniform(-1, 1, k))
    env.reset()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            env.step(rng.uniform(-1, 1, k))
            n += 1
            t_ep += 1
            if t_ep >= horizon:
                env.reset()
                t_ep = 0
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    out = {'host_cpus': os.cpu_count(), 'procs': a.procs, 'seconds_per_leg': a.seconds, 'unit': 'env-steps/s'}
    ctx = mp.get_context('spawn')
    for name in ('circle', 'planar_cached', 'iiwa_cached', 'planar', 'iiwa'):
        with ctx.Pool(1) as pool:                 # a fresh process per leg (the cached variant patches a module)
            one = pool.map(run, [(name, a.seconds, 0)])[0]
        with ctx.Pool(a.procs) as pool:
            rates = pool.map(run, [(name, a.seconds, i + 1) for i in range(a.procs)])
        out[name] = {'1_core': one, '%d_procs' % a.procs: sum(rates)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
