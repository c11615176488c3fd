"""ctypes binding of libatacom_hip.so (include/atacom_hip.h).  No numerics here.

The library is the product: if it is missing or cannot be loaded this module raises -- there is no
CPU / PyTorch fallback anywhere in the package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# ATACOM_LIB lets kernel-tuning experiments point at an alternative build of the same library
LIB_PATH = os.environ.get('ATACOM_LIB') or os.path.join(HERE, 'libatacom_hip.so')

ENV_CIRCLE, ENV_PLANAR, ENV_IIWA, ENV_CIRCLE_EC, ENV_CIRCLE_T = 0, 1, 2, 3, 4
F32, F64 = 0, 1
OK, E_INVALID, E_HIP, E_UNSUPPORTED = 0, -1, -2, -3
MAX_C, MAX_Q = 12, 6

EXPORTS = ['atacom_snapshot_bytes', 'atacom_snapshot_save', 'atacom_snapshot_restore', 'atacom_rollout_mlp', 'atacom_rollout_packed', 'atacom_get_aux_state', 'atacom_set_aux_state',
           'atacom_inverse_dynamics', 'atacom_forward_dynamics', 'atacom_default_config', 'atacom_get_dims', 'atacom_create', 'atacom_destroy', 'atacom_reset',
           'atacom_step', 'atacom_rollout', 'atacom_get_stats', 'atacom_get_state', 'atacom_set_state',
           'atacom_nullspace', 'atacom_constraint_terms', 'atacom_step_masked', 'atacom_canonical_mu', 'atacom_last_error', 'atacom_version', 'atacom_get_lanes',
           'atacom_get_filter_state', 'atacom_set_filter_state', 'atacom_set_seed', 'atacom_get_policy_lanes']


class AtacomConfig(C.Structure):
    """Mirror of `atacom_config` (include/atacom_hip.h)."""
    _fields_ = [('struct_size', C.c_int32), ('env_id', C.c_int32), ('batch', C.c_int32), ('dtype', C.c_int32),
                ('substeps', C.c_int32), ('horizon', C.c_int32), ('hold_q', C.c_int32), ('bias_mode', C.c_int32),
                ('auto_reset', C.c_int32), ('lanes_per_env', C.c_int32),
                ('dt', C.c_double), ('rref_tol', C.c_double), ('action_penalty', C.c_double), ('gamma', C.c_double),
                ('K', C.c_double * MAX_C), ('Kc', C.c_double * MAX_C), ('vel_max', C.c_double * MAX_Q),
                ('acc_max', C.c_double * MAX_Q), ('Kq', C.c_double * MAX_Q), ('pos_limit', C.c_double * MAX_Q),
                ('base_xy', C.c_double * 2), ('link', C.c_double * 3), ('term_tol', C.c_double), ('random_init', C.c_int32), ('seed', C.c_int32),
                ('dynamics_mode', C.c_int32), ('chart_mode', C.c_int32), ('task', C.c_int32), ('reserved0', C.c_int32),
                ('dt_base', C.c_double),
                ('obs_noise', C.c_int32), ('obs_delay', C.c_int32), ('env_noise', C.c_int32), ('reserved1', C.c_int32),
                ('puck_mass', C.c_double)]


class AtacomMlp(C.Structure):
    """Mirror of `atacom_mlp` (include/atacom_hip.h)."""
    _fields_ = [('struct_size', C.c_int32), ('n_in', C.c_int32), ('hidden', C.c_int32), ('n_out', C.c_int32),
                ('activation', C.c_int32), ('reserved', C.c_int32),
                ('W1', C.c_void_p), ('b1', C.c_void_p), ('W2', C.c_void_p), ('b2', C.c_void_p),
                ('W3', C.c_void_p), ('b3', C.c_void_p), ('obs_shift', C.c_void_p), ('obs_scale', C.c_void_p),
                ('std', C.c_void_p),
                ('sW1', C.c_void_p), ('sb1', C.c_void_p), ('sW2', C.c_void_p), ('sb2', C.c_void_p),
                ('sW3', C.c_void_p), ('sb3', C.c_void_p), ('log_std_min', C.c_double), ('log_std_max', C.c_double),
                ('squash', C.c_int32), ('reserved1', C.c_int32)]


class AtacomDims(C.Structure):
    _fields_ = [('dim_q', C.c_int32), ('n_f', C.c_int32), ('n_g', C.c_int32), ('n_null', C.c_int32),
                ('obs_dim', C.c_int32), ('state_dim', C.c_int32), ('init_state_dim', C.c_int32),
                ('record_dim', C.c_int32)]


class AtacomError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the shared library with argtypes set.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # The process must use ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; if libatacom_hip.so were
    # dlopen'ed first it would pull in /opt/rocm's copy and the two runtimes would fight over the device (observed:
    # hipGetDeviceCount -> "no ROCm-capable device").  Importing torch first makes the .so bind to torch's runtime.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001  (a pure-C consumer of the ABI does not need torch)
        pass
    if not os.path.exists(LIB_PATH):
        raise AtacomError("libatacom_hip.so is not built (%s). Run `python -m rl_on_manifold_amd.build` -- "
                          "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u8p = C.c_void_p, C.c_int32, C.c_void_p
    lib.atacom_default_config.argtypes = [i32, C.POINTER(AtacomConfig)]
    lib.atacom_get_dims.argtypes = [i32, C.POINTER(AtacomDims)]
    lib.atacom_create.argtypes = [C.POINTER(AtacomConfig), C.c_int, C.POINTER(vp)]
    lib.atacom_destroy.argtypes = [vp]
    lib.atacom_reset.argtypes = [vp, u8p, vp, vp, vp]
    lib.atacom_step.argtypes = [vp, vp, vp, vp, u8p, u8p, vp]
    lib.atacom_step_masked.argtypes = [vp, u8p, vp, vp, vp, u8p, u8p, vp]
    lib.atacom_canonical_mu.argtypes = [i32, i32, i32, vp, vp, vp, vp, C.c_double, vp, vp]
    lib.atacom_rollout.argtypes = [vp, i32, vp, vp, vp, vp, u8p, u8p, vp]
    lib.atacom_rollout_mlp.argtypes = [vp, i32, C.POINTER(AtacomMlp), vp, vp, vp, vp, vp, u8p, u8p, vp]
    lib.atacom_rollout_packed.argtypes = [vp, i32, vp, C.POINTER(AtacomMlp), vp, vp, i32, vp]
    lib.atacom_get_stats.argtypes = [vp, C.POINTER(C.c_double * 3), i32, vp]
    lib.atacom_get_lanes.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.atacom_get_policy_lanes.argtypes = [vp, C.POINTER(i32)]
    lib.atacom_set_seed.argtypes = [vp, i32]
    lib.atacom_get_state.argtypes = [vp, vp, vp]
    lib.atacom_set_state.argtypes = [vp, vp, vp]
    lib.atacom_get_aux_state.argtypes = [vp, vp, vp]
    lib.atacom_snapshot_bytes.argtypes = [vp]
    lib.atacom_snapshot_save.argtypes = [vp, vp, vp]
    lib.atacom_snapshot_restore.argtypes = [vp, vp, vp]
    lib.atacom_set_aux_state.argtypes = [vp, vp, vp]
    lib.atacom_get_filter_state.argtypes = [vp, vp, vp]
    lib.atacom_set_filter_state.argtypes = [vp, vp, vp]
    lib.atacom_inverse_dynamics.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp]
    lib.atacom_forward_dynamics.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.atacom_nullspace.argtypes = [i32, i32, i32, i32, vp, vp, C.c_double, vp, vp, vp, vp]
    lib.atacom_constraint_terms.argtypes = [C.POINTER(AtacomConfig), i32, vp, vp, vp, vp, vp, vp]
    lib.atacom_last_error.restype = C.c_char_p
    lib.atacom_version.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ('atacom_last_error', 'atacom_version', 'atacom_snapshot_bytes'):
            getattr(lib, name).restype = C.c_int
    lib.atacom_snapshot_bytes.restype = C.c_int64
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise AtacomError(load().atacom_last_error().decode())


def default_config(env_id):
    cfg = AtacomConfig()
    check(load().atacom_default_config(env_id, C.byref(cfg)))
    return cfg


def get_dims(env_id):
    d = AtacomDims()
    check(load().atacom_get_dims(env_id, C.byref(d)))
    return d
