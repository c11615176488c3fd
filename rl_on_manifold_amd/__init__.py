"""rl_on_manifold_amd -- MI355X-native batched ATACOM environment-step engine.

The hot path (constraint Jacobian assembly, null-space basis, pseudo-inverse solve, error correction,
slack integration, acceleration truncation, forward kinematics / Jacobians, dynamics, reward,
termination, constraint statistics) is hand-written HIP for gfx950 in `csrc/`, behind the C ABI of
`include/atacom_hip.h`; the Python here mirrors the reference's env surface and moves pointers only.
"""
from ._lib import AtacomError, LIB_PATH  # noqa: F401


def __getattr__(name):
    # torch is imported lazily so that `import rl_on_manifold_amd` (and the build) stay cheap
    if name in ('BatchedAtacomEnv', 'nullspace', 'constraint_terms', 'MlpPolicy', 'inverse_dynamics', 'forward_dynamics',
                'GraphedRollout', 'canonical_mu'):
        from . import engine
        return getattr(engine, name)
    if name in ('CircleEnvAtacom', 'AirHockeyPlanarAtacom', 'AirHockeyIiwaAtacom', 'CircleEnvErrorCorrection',
                'CircleEnvTerminated', 'VectorizedAtacomEnv'):
        from . import envs
        return getattr(envs, name)
    if name in ('RolloutCollector', 'RecordLayout'):
        from . import rollout
        return getattr(rollout, name)
    raise AttributeError(name)
