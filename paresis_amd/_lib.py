"""ctypes binding of libparesis_hip.so (the C ABI declared in include/paresis_hip.h).

The library is built in-tree by `make -C paresis_amd/csrc` (or __graft_entry__.build()).  There is NO fallback: if the
shared object is missing or a call fails, a PsxError is raised -- the product never routes through a CPU path.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libparesis_hip.so")

PSX_MAX_MAT = 8
PSX_MAX_DIST = 8
PSX_MAX_POISSON = 8
PSX_MAX_DETECT = 4
PSX_MAX_SRC = 16
PSX_SUM_SLOTS, PSX_SUM_STRIDE = 32, 16
ENGINE_AUTO, ENGINE_ROCFFT, ENGINE_LDS = 0, 1, 2
STATUS_NONFINITE = 1
ABI_VERSION = 10


class PsxError(RuntimeError):
    pass


_vp = c_void_p
_dp = POINTER(c_double)
_fp = POINTER(c_float)
_vpp = POINTER(c_void_p)

# name -> (restype, argtypes); one entry per symbol declared in include/paresis_hip.h
PROTOTYPES = {
    "psx_abi_version": (c_int, []),
    "psx_last_error": (c_char_p, []),
    "psx_device_ok": (c_int, []),
    "psx_clock_probe": (c_int, [POINTER(c_float), c_void_p]),
    "psx_transmit_wave_c64": (c_int, [_vp, c_float, _vpp, _dp, _dp, c_int, _vp, c_int64, _vp]),
    "psx_transmit_rt_f32": (c_int, [_vp, c_float, _vpp, _dp, _dp, c_int, _vp, _vp, _vp, c_int64, _vp]),
    "psx_accumulate_f32": (c_int, [_vp, _vp, c_float, _vpp, _dp, c_int, c_int, c_int64, _vp]),
    "psx_accumulate_sum_f32": (c_int, [_vp, _vp, c_float, _vpp, _dp, c_int, c_int, c_int64, _vp, c_double, _vp]),
    "psx_accumulate_many_f32": (c_int, [_vp, _vpp, _fp, c_int, _vpp, _dp, c_int, c_int, c_int64, _vp, _dp, _vp]),
    "psx_refract_workspace_bytes": (c_size_t, [c_int, c_int]),
    "psx_refract_set_halo": (c_int, [c_int]),
    "psx_set_deterministic": (c_int, [c_int]),
    "psx_get_deterministic": (c_int, []),
    "psx_set_deterministic_scale": (c_int, [c_float]),
    "psx_get_deterministic_scale": (c_float, []),
    "psx_refract_f32": (c_int, [_vp, c_float, _vpp, _dp, _dp, c_int, _vp, _vp, c_float, c_int, _vp, _vp, _vp, c_int,
                                c_int, c_int, c_double, c_double, c_double, _vp, _vp, _vp]),
    "psx_refract_split_f32": (c_int, [_vp, _vp, c_float, _vpp, _dp, _dp, c_int, _vp, _vp, _vp, c_float, c_int, c_int, c_int,
                                      c_int, c_double, c_double, c_double, _vp, _vp, _vp]),
    "psx_refract_multi_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "psx_refract_multi_f32": (c_int, [_vp, c_float, _vpp, _dp, _dp, c_int, _vp, _vpp, c_float, c_int, _vp, _vp, _vp,
                                      c_int, c_int, c_int, _dp, c_int, c_double, c_double, _vp, _vp, _vp]),
    "psx_refract_batch_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "psx_refract_batch_f32": (c_int, [c_int, _vpp, _fp, _vpp, _dp, _dp, c_int, _vpp, c_float, c_int, c_int, c_int, c_int, _dp,
                                      c_double, c_double, _vp, _vp, _vp]),
    "psx_fastloop_f32": (c_int, [_vp, _vp, _vp, _vp, c_int, c_int, _vp]),
    "psx_fresnel_plan_create": (c_int, [c_int, c_int, c_int, c_int, c_int, _vpp]),
    "psx_fresnel_plan_destroy": (c_int, [_vp]),
    "psx_fresnel_plan_engine": (c_int, [_vp]),
    "psx_fresnel_plan_bytes": (c_size_t, [_vp]),
    "psx_fresnel_plan_work_queue": (c_int, [_vp, c_int]),
    "psx_fresnel_propagate": (c_int, [_vp, _vp, c_float, _vpp, _dp, _dp, c_int, c_int, _dp, _dp, c_double, c_double,
                                      _vpp, _vpp, _fp, c_int, _vp]),
    "psx_fresnel_propagate_sources": (c_int, [_vp, c_int, c_int, _vpp, _fp, _vpp, _dp, _dp, c_int, _dp, _dp, c_double, c_double,
                                              _vpp, _vpp, _fp, _vp]),
    "psx_detector_plan_create": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_double, _vpp]),
    "psx_detector_plan_destroy": (c_int, [_vp]),
    "psx_detect_f32": (c_int, [_vp, _vp, _vp, _vp]),
    "psx_detect_multi_f32": (c_int, [_vp, _vpp, _vpp, c_int, _vp]),
    "psx_detector_operator_host": (c_int, [c_int, c_int, c_int, c_int, c_double, c_double, POINTER(c_int), _fp, c_int,
                                           POINTER(c_int)]),
    "psx_resize_f32": (c_int, [_vp, c_int, c_int, _vp, c_int, c_int, _vp]),
    "psx_poisson_f32": (c_int, [_vp, _vp, c_int64, c_uint64, _vp]),
    "psx_poisson_multi_f32": (c_int, [_vpp, POINTER(c_uint64), c_int, c_int64, _vp]),
    "psx_pack_counts_u16": (c_int, [_vp, _vp, c_int64, c_int64, _vp, _vp, c_int, _vp, _vp]),
    "psx_unpack_counts_u16": (c_int, [_vp, _vp, c_int64, _vp, _vp, c_int, _vp]),
    "psx_status_scan_f32": (c_int, [_vp, c_int64, _vp, _vp]),
    "psx_darkfield_workspace_bytes": (c_size_t, [c_int, c_int]),
    "psx_darkfield_blur_f32": (c_int, [_vp, _vp, _vp, _vp, c_int, c_int, c_int, _vp, _vp]),
    "psx_darkfield_split_f32": (c_int, [_vp, _vp, c_double, c_double, c_double, _vp, _vp, _vp, _vp, _vp, c_int, c_int, _vp]),
    "psx_darkfield_blur_prepared_f32": (c_int, [_vp, _vp, _vp, _vp, _vp, c_int, c_int, c_int, _vp, c_int, _vp]),
    "psx_darkfield_merge_f32": (c_int, [_vp, _vp, _vp, c_int64, _vp]),
    "psx_repad_f32": (c_int, [_vp, c_int, _vp, c_int, c_int, c_int, _vp]),
    "psx_membrane_f32": (c_int, [_dp, _dp, _dp, c_int64, c_int, c_int, c_int, c_int, c_double, c_int, _vp, _vp]),
    "psx_membrane_plan_create": (c_int, [_dp, _dp, _dp, c_int64, _vpp]),
    "psx_membrane_plan_destroy": (c_int, [_vp]),
    "psx_membrane_layer_f32": (c_int, [_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_int, _vp, _vp]),
    "psx_membrane_layers_f32": (c_int, [_vp, c_int, POINTER(c_int), POINTER(c_int), c_int, c_int, c_int, c_int, c_double, c_int,
                                        _vp, _vp, c_float, _vp]),
    "psx_debug_stamps": (c_int, [_vp]),
    "psx_debug_switch": (c_int, [c_char_p, c_int]),
    "psx_debug_switches_active": (c_int, [ctypes.c_char_p, c_size_t]),
    "psx_profile_enable": (c_int, [c_int]),
    "psx_profile_summary": (c_int, [ctypes.c_char_p, c_size_t]),
}

_LIB = None


def lib():
    """Load (once) and return the bound library; raises PsxError when it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PsxError("libparesis_hip.so is not built (%s missing): run `make -C paresis_amd/csrc` or "
                       "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise PsxError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise PsxError("libparesis_hip.so does not export %s (stale build?)" % name) from exc
        fn.restype = res
        fn.argtypes = args
    if handle.psx_abi_version() != ABI_VERSION:
        raise PsxError("libparesis_hip.so ABI %d != binding ABI %d (rebuild)" % (handle.psx_abi_version(), ABI_VERSION))
    _LIB = handle
    return _LIB


def check(rc, what):
    """Turn a non-zero return code into a PsxError carrying psx_last_error()."""
    if rc != 0:
        msg = lib().psx_last_error()
        raise PsxError("%s failed (rc=%d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else "?"))
