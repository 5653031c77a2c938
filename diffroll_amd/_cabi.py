"""ctypes binding of the C-ABI: include/diffroll_amd.h (the boundary) and include/diffroll_amd_debug.h (measurement /
checker / test entry points of the same library) - the only way Python reaches the kernels.

There is deliberately no fallback: if the shared library is missing or no MI355X is visible the
calls raise - a silent CPU path would void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DR_LIB") or os.path.join(_HERE, "lib", "libdiffroll_amd.so")   # DR_LIB: measurement builds

DR_ABI_VERSION = 10
DR_OK, DR_EINVAL, DR_ESTATE, DR_EHIP, DR_ENOMEM, DR_ENAME, DR_ETIMEOUT = 0, -1, -2, -3, -4, -5, -6

SAMPLERS = {
    "ddpm_x0": 0,
    "cfdg_ddpm_x0": 1,
    "generation_ddpm_x0": 2,
    "inpainting_ddpm_x0": 3,
    "ddim_x0": 4,
    "cfdg_ddim_x0": 5,
    "ddpm": 6,          # epsilon prediction (task/diffusion.py:804-829)
    "ddim": 7,          # epsilon prediction (:877-892)
    "ddim2ddpm": 8,     # epsilon prediction (:894-911)
}
COND_SPEC, COND_UNCOND = 0, 1
PRECISIONS = {"f32": 0, "bf16x3": 1}
NORM_MODES = {"imagewise": 0, "framewise": 1}

# every symbol include/diffroll_amd.h declares (the boundary) ...
EXPORTS = [
    "dr_abi_version", "dr_create", "dr_destroy", "dr_last_error", "dr_set_param", "dr_set_tables", "dr_set_frontend_tables",
    "dr_commit", "dr_frontend", "dr_forward", "dr_forward_steps", "dr_step", "dr_sample", "dr_sample_checked", "dr_finish",
    "dr_pending_timeout", "dr_launch_state", "dr_note_runs", "dr_frame_counts", "dr_q_sample", "dr_extract_x0",
    "dr_set_spec_norm", "dr_set_precision", "dr_set_option",
    "dr_comm_unique_id", "dr_comm_create", "dr_comm_destroy", "dr_comm_info", "dr_comm_last_error", "dr_gather",
]
# ... and every symbol of include/diffroll_amd_debug.h (the lab)
DEBUG_EXPORTS = [
    "dr_debug_stft_power", "dr_debug_bounds", "dr_debug_tenants", "dr_debug_kfd_root", "dr_debug_set_option", "dr_debug_ticks",
    "dr_stack_status", "dr_cold_times", "dr_profile_enable", "dr_profile_read", "dr_profile_read_ex",
    "dr_bench_layer", "dr_bench_pointwise",
]
# the options dr_set_option knows; every other name goes to dr_debug_set_option (Engine.set_option)
PUBLIC_OPTIONS = ("blocked_accumulation", "fused_rearm", "fused_stack", "fused_tail")
MODES = {0: "none", 1: "per_phase", 2: "fused_stack", 3: "fused_stack+tail"}


class DrConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("residual_channels", C.c_int32), ("residual_layers", C.c_int32),
        ("kernel_size", C.c_int32), ("dilation_base", C.c_int32), ("dilation_bound", C.c_int32),
        ("n_mels", C.c_int32), ("timesteps", C.c_int32), ("sample_rate", C.c_int32),
        ("n_fft", C.c_int32), ("hop_length", C.c_int32),
        ("f_min", C.c_float), ("f_max", C.c_float), ("beta_start", C.c_float), ("beta_end", C.c_float),
    ]


class DrLaunchInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("fused_enabled", C.c_int32), ("fallbacks", C.c_int64), ("yields", C.c_int64),
                ("rearms", C.c_int64), ("stack_launches", C.c_int64), ("tail_launches", C.c_int64)]


_lib = None


def bounds_violations(reset: bool = False):
    """Checker builds (DR_LIB pointing at a -DDR_BOUNDS library): (code of the first violated check, detail, detail,
    count); None for a production library."""
    lib = load_library()
    out = (C.c_int64 * 4)()
    if lib.dr_debug_bounds(out, 1 if reset else 0) != 0:
        return None
    return tuple(int(v) for v in out)


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the engine and declare the prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first (python -m diffroll_amd.build). "
            "diffroll_amd has no CPU fallback.")
    lib = C.CDLL(path)
    vp, f32p = C.c_void_p, C.POINTER(C.c_float)
    lib.dr_abi_version.restype = C.c_int
    lib.dr_abi_version.argtypes = []
    lib.dr_create.restype = C.c_int
    lib.dr_create.argtypes = [C.POINTER(vp), C.POINTER(DrConfig)]
    lib.dr_destroy.restype = None
    lib.dr_destroy.argtypes = [vp]
    lib.dr_last_error.restype = C.c_char_p
    lib.dr_last_error.argtypes = [vp]
    lib.dr_set_param.restype = C.c_int
    lib.dr_set_param.argtypes = [vp, C.c_char_p, f32p, C.c_size_t]
    lib.dr_set_tables.restype = C.c_int
    lib.dr_set_tables.argtypes = [vp, f32p, f32p]
    lib.dr_set_frontend_tables.restype = C.c_int
    lib.dr_set_frontend_tables.argtypes = [vp, f32p, C.c_float, f32p]
    lib.dr_commit.restype = C.c_int
    lib.dr_commit.argtypes = [vp, vp]
    lib.dr_frontend.restype = C.c_int
    lib.dr_frontend.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.dr_forward.restype = C.c_int
    lib.dr_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.dr_forward_steps.restype = C.c_int
    lib.dr_forward_steps.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, vp, vp]
    lib.dr_step.restype = C.c_int
    lib.dr_step.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, vp]
    lib.dr_sample.restype = C.c_int
    lib.dr_sample.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_int, vp]
    lib.dr_finish.restype = C.c_int
    lib.dr_finish.argtypes = [vp, vp]
    lib.dr_sample_checked.restype = C.c_int
    lib.dr_sample_checked.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_int,
                                      C.POINTER(C.c_int32), vp]
    lib.dr_launch_state.restype = C.c_int
    lib.dr_launch_state.argtypes = [vp, C.POINTER(DrLaunchInfo)]
    lib.dr_pending_timeout.restype = C.c_int
    lib.dr_pending_timeout.argtypes = [vp, vp]
    lib.dr_cold_times.restype = C.c_int
    lib.dr_cold_times.argtypes = [vp, C.POINTER(C.c_double)]
    lib.dr_frame_counts.restype = C.c_int
    lib.dr_frame_counts.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.POINTER(C.c_int64), vp]
    lib.dr_note_runs.restype = C.c_int
    lib.dr_note_runs.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
    for fn in (lib.dr_q_sample, lib.dr_extract_x0):
        fn.restype = C.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp]
    lib.dr_debug_stft_power.restype = C.c_int
    lib.dr_debug_stft_power.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.dr_debug_bounds.restype = C.c_int
    lib.dr_debug_bounds.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.dr_debug_tenants.restype = C.c_int
    lib.dr_debug_tenants.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    lib.dr_debug_kfd_root.restype = C.c_int
    lib.dr_debug_kfd_root.argtypes = [C.c_char_p]
    lib.dr_debug_set_option.restype = C.c_int
    lib.dr_debug_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.dr_set_spec_norm.restype = C.c_int
    lib.dr_set_spec_norm.argtypes = [vp, C.c_int]
    lib.dr_set_precision.restype = C.c_int
    lib.dr_set_precision.argtypes = [vp, C.c_int]
    lib.dr_profile_enable.restype = C.c_int
    lib.dr_profile_enable.argtypes = [vp, C.c_int]
    lib.dr_profile_read.restype = C.c_int
    lib.dr_profile_read.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int]
    lib.dr_profile_read_ex.restype = C.c_int
    lib.dr_profile_read_ex.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.c_char_p, C.c_size_t, C.c_int]
    lib.dr_set_option.restype = C.c_int
    lib.dr_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    lib.dr_stack_status.restype = C.c_int
    lib.dr_stack_status.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    lib.dr_bench_layer.restype = C.c_int
    lib.dr_bench_layer.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.dr_bench_pointwise.restype = C.c_int
    lib.dr_bench_pointwise.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    lib.dr_debug_ticks.restype = C.c_int
    lib.dr_debug_ticks.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dr_comm_unique_id.restype = C.c_int
    lib.dr_comm_unique_id.argtypes = [C.c_char_p]
    lib.dr_comm_create.restype = C.c_int
    lib.dr_comm_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.dr_comm_destroy.restype = None
    lib.dr_comm_destroy.argtypes = [vp]
    lib.dr_comm_info.restype = C.c_int
    lib.dr_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.dr_comm_last_error.restype = C.c_char_p
    lib.dr_comm_last_error.argtypes = []
    lib.dr_gather.restype = C.c_int
    lib.dr_gather.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp]
    if lib.dr_abi_version() != DR_ABI_VERSION:
        raise RuntimeError("libdiffroll_amd.so ABI version mismatch: rebuild it")
    _lib = lib
    return lib
