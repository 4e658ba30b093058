"""ctypes binding of ``libvgpa_hip.so`` (declared in ``include/vgpa.h``).

The library is built in-tree by ``__graft_entry__.build()`` (or
``python -m vgpa_amd.build``).  There is no CPU implementation behind these
entry points: if the shared object is missing the import fails, and if no HIP
device is present every compute call raises.
"""
import ctypes
import os

# PyTorch ships its own HIP runtime (libamdhip64.so.7).  It has to be in the process
# before libvgpa_hip.so is loaded so that the dynamic loader binds the kernels to that
# same runtime instance: device pointers and streams are only meaningful within one.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGPA_HIP_LIBRARY names another build of the same library (experimental builds, profiling builds)
LIB_PATH = os.environ.get("VGPA_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libvgpa_hip.so")

ABI_VERSION = 1

# enums of include/vgpa.h
EULER, HEUN, RK2, RK4 = 0, 1, 2, 3
METHODS = {"EULER": EULER, "HEUN": HEUN, "RK2": RK2, "RK4": RK4}
SYS_OU, SYS_DW, SYS_L63, SYS_L63_UT, SYS_L96_UT = 0, 1, 2, 3, 4
SYSTEMS = {"ou": SYS_OU, "dw": SYS_DW, "l63": SYS_L63, "l63_ut": SYS_L63_UT, "l96_ut": SYS_L96_UT}
ST_CHOL_FALLBACK, ST_CHOL_FAILED, ST_UNSUPPORTED = 1, 2, 4
GRAD_OUTER, GRAD_SIGINV_DIAG = 1, 2
SCAN_SYMMETRIC = 1
SCAN_STATE_SYMMETRIC = 2
SCAN_PSI_ROW0 = 4
SCAN_PACKED = 8
SCAN_PACKED_DEDS = 16
PACKED_S, PACKED_DEDS = 1, 2
SYM_PACKED_40 = 1152     # doubles of a packed symmetric 40 x 40 matrix (vgpa_sym_packed_size(40))
MAX_D = 60
OBS_STATIC_MAX = 1 << 17

_c_int, _c_dbl, _ptr = ctypes.c_int, ctypes.c_double, ctypes.c_void_p

_ESDE_ARGS = [_c_int, _c_int, _c_int] + [_ptr] * 4 + [_ptr, _c_int, _ptr, _c_int] + [_ptr] * 8 + [_ptr]

# name -> argtypes; every function returns int except the two noted below
SIGNATURES = {
    "vgpa_fwd_scan": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_dbl, _c_int,
                      _ptr, _ptr, _ptr],
    "vgpa_bwd_scan": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_dbl,
                      _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "vgpa_bwd_scan_grad": [_c_int] * 4 + [_ptr] * 9 + [_ptr, _c_int, _ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int,
                           _c_dbl, _c_int, _c_int] + [_ptr] * 7,
    "vgpa_bwd_scan_grad_is_fused": [_c_int] * 8,
    "vgpa_esde": [_c_int] + _ESDE_ARGS,
    "vgpa_esde_ou": _ESDE_ARGS,
    "vgpa_esde_dw": _ESDE_ARGS,
    "vgpa_esde_l63": _ESDE_ARGS,
    "vgpa_esde_l63_ut": _ESDE_ARGS,
    "vgpa_esde_l96_ut": _ESDE_ARGS,
    "vgpa_esde_l96_ut_packed": [_c_int, _c_int, _c_int] + [_ptr] * 4 + [_ptr, _c_int, _ptr, _c_int] + [_ptr] * 7 + [_ptr],
    "vgpa_obs_terms_packed": [_c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr,
                              _c_dbl, _ptr, _ptr, _ptr, _ptr, _ptr],
    "vgpa_sym_pack": [ctypes.c_longlong, _c_int, _ptr, _ptr, _ptr],
    "vgpa_sym_unpack": [ctypes.c_longlong, _c_int, _ptr, _ptr, _ptr],
    "vgpa_sym_packed_size": [_c_int],
    "vgpa_reduce_energy": [_c_int, _c_int, _c_int, _ptr, _ptr, _c_dbl, _ptr, _ptr],
    "vgpa_obs_terms": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr,
                       _c_dbl, _ptr, _ptr, _ptr, _ptr, _ptr],
    "vgpa_obs_terms_static": [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr,
                              _c_dbl, _ptr, _ptr, _ptr, _ptr],
    "vgpa_kl0": [_c_int, _c_int, _ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _c_int, _ptr, _ptr],
    "vgpa_grad_assemble": [_c_int, _c_int, _c_int] + [_ptr] * 8 + [_c_int, _ptr, _c_int, _ptr, _c_int, _c_dbl, _c_int,
                           _ptr, _ptr, _ptr],
    "vgpa_ut_sigma_points": [_c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "vgpa_ut_moments": [_c_int, _c_int, _c_int, _ptr, _ptr, _ptr, _ptr],
    "vgpa_grad_esde_ms": [_c_int, _c_int] + [_ptr] * 8 + [_ptr],
    "vgpa_mom_gauss": [ctypes.c_longlong, _ptr, _ptr, _c_int, _c_int, _ptr, _ptr],
    "vgpa_sde_path": [_c_int] * 4 + [_ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _c_int, ctypes.c_double, ctypes.c_double,
                      _ptr, _ptr, _ptr],
    "vgpa_init_ab0": [_c_int] * 3 + [_ptr, _ptr, _c_int, _ptr, _c_int, ctypes.c_double, _ptr, _ptr, _ptr],
    "vgpa_scg_dots": [_c_int, ctypes.c_longlong, ctypes.c_longlong, _ptr, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr],
    "vgpa_scg_step": [_c_int, ctypes.c_longlong, ctypes.c_longlong] + [_ptr] * 6 + [_ptr],
    "vgpa_scg_curvature": [_c_int, ctypes.c_longlong, ctypes.c_longlong] + [_ptr] * 4 + [_c_int, _ptr, _ptr, _ptr],
    "vgpa_scg_trial_stats": [_c_int, ctypes.c_longlong, ctypes.c_longlong] + [_ptr] * 5 + [_c_int, _ptr, _ptr, _ptr],
    "vgpa_scg_update": [_c_int, ctypes.c_longlong, ctypes.c_longlong] + [_ptr] * 7 + [_c_int, _ptr, _ptr, _ptr],
}
EXPORTS = ["vgpa_version", "vgpa_last_error_string"] + list(SIGNATURES)


class VgpaError(RuntimeError):
    """A libvgpa_hip.so entry point returned a negative status."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vgpa_version.restype = _c_int
    lib.vgpa_version.argtypes = []
    lib.vgpa_last_error_string.restype = ctypes.c_char_p
    lib.vgpa_last_error_string.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = _c_int
        fn.argtypes = args
    if lib.vgpa_version() != ABI_VERSION:
        raise ImportError(f"libvgpa_hip.so ABI {lib.vgpa_version()} != expected {ABI_VERSION}: rebuild")
    return lib


lib = _load()


def call(name, *args):
    """Call an entry point and raise VgpaError (ValueError for a bad integration
    method, as the reference does) on a negative return code."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.vgpa_last_error_string().decode("utf-8", "replace")
        if "Unknown method of integration" in msg:
            raise ValueError("Unknown method of integration")
        if "Unknown Higher Order Gaussian Moment" in msg:
            raise ValueError("Unknown Higher Order Gaussian Moment!")
        raise VgpaError(f"{name} failed ({rc}): {msg}")
