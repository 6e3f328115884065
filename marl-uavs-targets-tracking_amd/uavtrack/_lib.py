"""ctypes binding of include/uavtrack.h.  Loads libuavtrack.so from this directory and
fails loudly if it is missing -- there is no Python or CPU stand-in for the kernels."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UAVTRACK_LIB points at another build of the same ABI (A/B timing of kernel variants)
LIB_PATH = os.environ.get("UAVTRACK_LIB") or os.path.join(_HERE, "libuavtrack.so")

ABI_VERSION = 1
OBS_DIM = 12
MAX_CLIMB = 8

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class UavtrackConfig(C.Structure):
    """Mirror of `struct uavtrack_config` (include/uavtrack.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_envs", C.c_int32), ("n_uav", C.c_int32), ("m_targets", C.c_int32),
        ("dim", C.c_int32), ("na", C.c_int32), ("nc", C.c_int32),
        ("norm_n_uav", C.c_int32), ("norm_m_targets", C.c_int32),
        ("reward_mode", C.c_int32), ("horizon", C.c_int32), ("device_id", C.c_int32),
        ("env_offset", C.c_int64),
        ("x_max", C.c_double), ("y_max", C.c_double), ("z_max", C.c_double),
        ("dt", C.c_double), ("u_v_max", C.c_double), ("u_h_max", C.c_double), ("u_g_max", C.c_double),
        ("dc", C.c_double), ("dp", C.c_double), ("t_v_max", C.c_double),
        ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
        ("cooperative", C.c_double),
    ]


class HostStep(C.Structure):
    """Mirror of `struct uavtrack_host_step` (include/uavtrack.h): host pointers into the library's pinned block."""
    _fields_ = [(k, C.c_void_p) for k in ("obs", "reward", "terms", "raw", "covered", "done",
                                          "ux", "uy", "uz", "uh", "ua", "tx", "ty", "tz", "th", "step_count")]


ACTOR_SAMPLE, ACTOR_ARGMAX = 0, 1   # enum in include/uavtrack.h
PROF_CLASSES = ("rollout", "scorer", "mix", "ep_sums")   # UAVTRACK_PROF_* in include/uavtrack.h
PMI_SCHEMES = ("auto", "f16x3", "bf16x6", "fp32")         # enum uavtrack_pmi_scheme

# name -> (restype, argtypes); every symbol declared in include/uavtrack.h
SIGNATURES = {
    "uavtrack_version": (C.c_int, []),
    "uavtrack_last_error": (C.c_char_p, []),
    "uavtrack_create": (C.c_int, [C.POINTER(UavtrackConfig), C.POINTER(C.c_void_p)]),
    "uavtrack_destroy": (C.c_int, [C.c_void_p]),
    "uavtrack_reset": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "uavtrack_set_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10 + [C.c_void_p]),
    "uavtrack_get_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10 + [C.c_void_p]),
    "uavtrack_set_episodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "uavtrack_get_episodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "uavtrack_set_pmi_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "uavtrack_set_pmi_scheme": (C.c_int, [C.c_void_p, C.c_int32]),
    "uavtrack_pmi_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "uavtrack_pmi_inference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "uavtrack_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_void_p]),
    "uavtrack_step_accumulate": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7 + [C.c_void_p]),
    "uavtrack_step_many": (C.c_int, [C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p]),
    "uavtrack_step_many_autoreset": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64] + [C.c_void_p] * 7 + [C.c_void_p]),
    "uavtrack_run_greedy": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64] + [C.c_void_p] * 7 + [C.c_void_p]),
    "uavtrack_greedy_actions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "uavtrack_set_actor_weights": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p]),
    "uavtrack_actor_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uavtrack_run_actor": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32] + [C.c_void_p] * 8 + [C.c_void_p]),
    "uavtrack_set_target_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "uavtrack_set_raw_reward_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "uavtrack_step_host": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(HostStep), C.c_void_p]),
    "uavtrack_pmi_pairs_scored": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    "uavtrack_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "uavtrack_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p]),
    "uavtrack_kernel_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "uavtrack_launch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or "
                f"`make -C marl-uavs-targets-tracking_amd/csrc`.  uavtrack has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("UAVTRACK_LIB_OLDER_OK") and not hasattr(lib, name):
                continue              # A/B timing against an older build of the library (tools/sweep.py --lib)
            fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.uavtrack_version() != ABI_VERSION:
            raise RuntimeError(f"libuavtrack ABI {lib.uavtrack_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().uavtrack_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'uavtrack'} failed: {msg}")
