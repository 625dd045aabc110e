"""ctypes loader for libgridhip.so (the C ABI declared in include/gridhip.h).

There is no CPU fallback: if the HIP library has not been built this module raises, and every
compute entry point needs a gfx950 device.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(_PKG, "..", ".."))          # ska-sdp-accelerate-gridding_amd/
# GRIDHIP_LIB: another build of the same ABI (tools/ use lib/libgridhip_tuning.so, which has the "dbg" option)
LIB_PATH = os.environ.get("GRIDHIP_LIB") or os.path.join(ROOT, "lib", "libgridhip.so")

i64 = C.c_int64
vp = C.c_void_p
ci = C.c_int

OK = 0
EINVAL, ENOMEM, EHIP, ENODEV, EUNSUPPORTED = -1, -2, -3, -4, -5

# name -> (restype, argtypes); mirrors include/gridhip.h one to one
_GRID_DEV = [vp, i64, i64, vp, i64, vp, vp, i64, vp]
_CONV_DEV = [vp, i64, i64, vp, i64, i64, i64, i64, vp, vp, vp, i64, vp]
_CONV2_DEV = [vp, i64, i64, vp, i64, i64, i64, i64, i64, vp, vp, vp, i64, vp, vp]
SIGNATURES = {
    "gridhip_version": (ci, []),
    "gridhip_strerror": (C.c_char_p, [ci]),
    "gridhip_device_count": (ci, [C.POINTER(ci)]),
    "gridhip_create": (ci, [ci, C.POINTER(vp)]),
    "gridhip_destroy": (ci, [vp]),
    "gridhip_last_error": (C.c_char_p, [vp]),
    "gridhip_set_stream": (ci, [vp, vp]),
    "gridhip_reset_stream": (ci, [vp]),
    "gridhip_get_stream": (vp, [vp]),
    "gridhip_synchronize": (ci, [vp]),
    "gridhip_set_option": (ci, [vp, C.c_char_p, i64]),
    "gridhip_get_option": (ci, [vp, C.c_char_p, C.POINTER(i64)]),
    "gridhip_last_dropped": (ci, [vp, C.POINTER(i64)]),
    "gridhip_grid": (ci, _GRID_DEV),
    "gridhip_convgrid": (ci, _CONV_DEV),
    "gridhip_convgrid2": (ci, _CONV2_DEV),
    "gridhip_degrid2": (ci, _CONV2_DEV),
    "gridhip_grid_dev": (ci, _GRID_DEV),
    "gridhip_convgrid_dev": (ci, _CONV_DEV),
    "gridhip_convgrid2_dev": (ci, _CONV2_DEV),
    "gridhip_degrid2_dev": (ci, _CONV2_DEV),
    "gridhip_plan_create_dev": (ci, [vp, i64, i64, i64, i64, i64, i64, i64, vp, vp, i64, vp, C.POINTER(vp)]),
    "gridhip_plan_grid_dev": (ci, [vp, vp, vp, vp]),
    "gridhip_plan_degrid_dev": (ci, [vp, vp, vp, vp]),
    "gridhip_plan_destroy": (ci, [vp]),
    "gridhip_image_size": (i64, [C.c_double, i64]),
    "gridhip_wbins": (ci, [vp, i64, vp, i64, vp, C.POINTER(i64), C.POINTER(i64)]),
    "gridhip_find_closest": (ci, [vp, i64, vp, i64, vp, vp]),
    "gridhip_mirror_uvw": (ci, [vp, i64, vp, vp, vp, vp]),
    "gridhip_doweight": (ci, [vp, C.c_double, i64, i64, vp, vp, vp]),
    "gridhip_make_grid_hermitian": (ci, [vp, i64, vp]),
    "gridhip_fft2_centered": (ci, [vp, i64, vp, vp, ci]),
    "gridhip_w_kernel": (ci, [vp, C.c_double, C.c_double, i64, i64, i64, vp]),
    "gridhip_simple_imaging": (ci, [vp, C.c_double, i64, i64, vp, vp, i64, vp, vp]),
    "gridhip_conv_imaging": (ci, [vp, i64, i64, i64, vp, C.c_double, i64, i64, vp, vp, i64, vp, vp]),
    "gridhip_w_cache_imaging": (ci, [vp, i64, i64, i64, i64, C.c_double, i64, i64, vp, vp, vp, i64, vp, vp]),
    "gridhip_awgrid": (ci, [vp, i64, i64, vp, i64, i64, i64, i64, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "gridhip_awgrid_dev": (ci, [vp, i64, i64, vp, i64, i64, i64, i64, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "gridhip_aw_last_stats": (ci, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "gridhip_aw_imaging": (ci, [vp, C.c_double, i64, i64, i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp,
                                vp, vp]),
    "gridhip_do_imaging": (ci, [vp, ci, i64, i64, i64, i64, i64, vp, C.c_double, i64, i64, vp, vp, vp, i64, vp, vp,
                                vp, C.POINTER(C.c_double)]),
    "gridhip_do_imaging_dev": (ci, [vp, ci, i64, i64, i64, i64, i64, vp, C.c_double, i64, i64, vp, vp, vp, i64, vp, vp,
                                    vp, C.POINTER(C.c_double)]),
    "gridhip_w_cache_imaging_dev": (ci, [vp, i64, i64, i64, i64, C.c_double, i64, i64, vp, vp, vp, i64, vp, vp]),
    "gridhip_comm_create": (ci, [ci, C.POINTER(ci), C.POINTER(vp)]),
    "gridhip_comm_unique_id": (ci, [vp]),
    "gridhip_comm_create_rank": (ci, [vp, ci, ci, vp, C.POINTER(vp)]),
    "gridhip_comm_destroy": (ci, [vp]),
    "gridhip_comm_last_error": (C.c_char_p, [vp]),
    "gridhip_comm_ndev": (ci, [vp]),
    "gridhip_comm_nranks": (ci, [vp]),
    "gridhip_comm_ctx": (vp, [vp, ci]),
    "gridhip_comm_allreduce_grids": (ci, [vp, i64, C.POINTER(vp)]),
    "gridhip_comm_allreduce_grid": (ci, [vp, i64, vp]),
    "gridhip_comm_allreduce_rows": (ci, [vp, i64, i64, i64, C.POINTER(vp)]),
    "gridhip_comm_allreduce_grid_rows": (ci, [vp, i64, i64, i64, vp]),
    "gridhip_comm_set_option": (ci, [vp, C.c_char_p, i64]),
    "gridhip_comm_get_option": (ci, [vp, C.c_char_p, C.POINTER(i64)]),
    "gridhip_comm_set_stream": (ci, [vp, ci, vp]),
    "gridhip_comm_reset_stream": (ci, [vp, ci]),
    "gridhip_comm_convgrid2": (ci, [vp] + _CONV2_DEV[1:]),
    "gridhip_malloc": (ci, [vp, C.POINTER(vp), i64]),
    "gridhip_free": (ci, [vp, vp]),
    "gridhip_memcpy_h2d": (ci, [vp, vp, vp, i64]),
    "gridhip_memcpy_d2h": (ci, [vp, vp, vp, i64]),
    "gridhip_memset": (ci, [vp, vp, ci, i64]),
    "gridhip_last_timing": (ci, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gridhip_timing": (ci, [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gridhip_enable_timing": (ci, [vp, ci]),
}

_lib = None


class GridHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gridhip error {code}: {msg}")
        self.code = code


def load():
    """Load libgridhip.so and attach prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ska-sdp-accelerate-gridding_amd/csrc`). There is no CPU fallback."
        )
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 and loads
    # it by path, so a system copy loaded first would leave torch a second runtime that finds
    # no GPUs.  Importing torch first makes libgridhip's NEEDED libamdhip64.so.7 resolve (by
    # SONAME) to the copy torch uses.  Callers without torch (C, Haskell) get the system one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
