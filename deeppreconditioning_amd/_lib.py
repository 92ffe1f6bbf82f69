"""ctypes binding of libdpcg.so (the C ABI in include/dpcg.h).

The HIP library is the product: if it is missing or cannot be loaded this module raises, there is
no CPU or PyTorch fallback for any operation of the solve path.
"""

from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import os

_CSRC = pathlib.Path(__file__).resolve().parent / "csrc"
# DPCG_LIBRARY (development): load another build of the same ABI, e.g. an older one for an A/B measurement
LIB_PATH = pathlib.Path(os.environ["DPCG_LIBRARY"]) if os.environ.get("DPCG_LIBRARY") else _CSRC / "libdpcg.so"

# status codes (include/dpcg.h)
OK, MAX_ITER, BREAKDOWN = 0, 1, 2
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_PIVOT, ERR_STATE = -1, -2, -3, -4, -5
F64, F32 = 0, 1
DEVICE, HOST = 0, 1
PRECOND_NONE, PRECOND_JACOBI, PRECOND_CSR, PRECOND_LLT_MULTIPLY, PRECOND_LLT_SOLVE, PRECOND_CALLBACK = 0, 1, 2, 3, 4, 5
PRECOND_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)   # dpcg_precond_fn
INIT_CHECK_R, SPMV_F32, NO_GRAPH, NO_SMALL, VAL32_IF_LOSSLESS, NO_FUSE, NO_TEAM, TEAM = 1, 2, 4, 8, 16, 32, 64, 128
REORDER_NONE, REORDER_AUTO, REORDER_ALWAYS, REORDER_REGIONS = 0, 1, 2, 3
ORDER_CALLER, ORDER_MULTICOLOR = 0, 1

# name -> (restype, argtypes); every symbol include/dpcg.h declares
_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_dbl = C.c_double
SIGNATURES = {
    "dpcg_version": (_int, []),
    "dpcg_status_string": (C.c_char_p, [_int]),
    "dpcg_last_error": (C.c_char_p, []),
    "dpcg_device_info": (_int, [C.POINTER(_int), C.POINTER(_i64), C.c_char_p, _int]),
    "dpcg_create": (_int, [C.POINTER(_p), _i64, _i64, _p, _p, _p, _int, _int, _int, _p]),
    "dpcg_destroy": (_int, [_p]),
    "dpcg_get_info": (_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int),
                             C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int)]),
    "dpcg_reorder": (_int, [_p, _int, _p, C.POINTER(_int)]),
    "dpcg_get_permutation": (_int, [_p, C.POINTER(_int), _p, C.POINTER(_dbl)]),
    "dpcg_set_precond_none": (_int, [_p]),
    "dpcg_set_precond_callback": (_int, [_p, PRECOND_FN, _p]),
    "dpcg_set_precond_jacobi": (_int, [_p, _p, _int, _p]),
    "dpcg_set_precond_csr": (_int, [_p, _i64, _p, _p, _p, _int, _p]),
    "dpcg_set_precond_llt": (_int, [_p, _int, _i64, _p, _p, _p, _int, _p]),
    "dpcg_set_precond_ic0": (_int, [_p, _int, _p]),
    "dpcg_set_precond_ic0_ordered": (_int, [_p, _int, _int, _p]),
    "dpcg_get_precond_ordering": (_int, [_p, C.POINTER(_int), _p]),
    "dpcg_set_precond_ict": (_int, [_p, _int, _int, _dbl, _p]),
    "dpcg_set_precond_icholt": (_int, [_p, _int, _int, _dbl, _p]),
    "dpcg_get_reduction_geometry": (_int, [_p, _p]),
    "dpcg_get_chip_info": (_int, [_p, _p, _p]),
    "dpcg_debug_occupy": (_int, [_int, C.c_double, _p]),
    "dpcg_debug_l2_gather": (_int, [_int, _int, C.POINTER(C.c_int32), _int, _int, _p, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_int)]),
    "dpcg_get_factor": (_int, [_p, _p, _p, _p]),
    "dpcg_spmv": (_int, [_p, _p, _p, _p]),
    "dpcg_spmv_f32": (_int, [_p, _p, _p, _p]),
    "dpcg_precond_apply": (_int, [_p, _p, _p, _p]),
    "dpcg_sptrsv": (_int, [_p, _int, _p, _p, _p]),
    "dpcg_dot": (_int, [_i64, _p, _p, C.POINTER(_dbl), _p]),
    "dpcg_spmv_dot_bench": (_int, [_p, _p, _p, _int, C.POINTER(C.c_float), _p]),
    "dpcg_release_cached_memory": (_int, []),
    "dpcg_update_values": (_int, [_p, _p, _int, _int, _p]),
    "dpcg_stream_bench": (_int, [_int, _int, _int, _i64, _int, C.POINTER(C.c_float), C.POINTER(_i64), _p]),
    "dpcg_solve": (_int, [_p, _p, _p, _p, _dbl, _dbl, _int, _int, _p, C.POINTER(_int), C.POINTER(_dbl),
                          C.POINTER(_dbl), _p, _p, _p]),
    "dpcg_solve_batch": (_int, [_int, _p, _p, _p, _p, _dbl, _dbl, _int, _int, _int, _p, _p, _p, _p]),
    "dpcg_poisson_sizes": (_int, [_int, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "dpcg_gen_poisson": (_int, [_int, _i64, _p, _p, _p, _int, _p]),
    "dpcg_batched_coo_spmv": (_int, [_i64, _p, _p, _int, _i64, _p, _p, _int, _p]),
    "dpcg_batched_coo_edge": (_int, [_i64, _p, _int, _i64, _p, _p, _p, _int, _p]),
    "dpcg_batched_coo_spmm": (_int, [_i64, _p, _p, _int, _i64, _int, _p, _p, _int, _p]),
    "dpcg_batched_coo_sddmm": (_int, [_i64, _p, _int, _i64, _int, _p, _p, _p, _int, _p]),
    "dpcg_coo_to_csr": (_int, [_i64, _i64, _p, _p, _p, _p, _p, _p, C.POINTER(_i64), _p]),
    "dpcg_convnet_plan_create": (_int, [C.POINTER(_p), _int, _i64, _i64, _i64, _p, _int, _p, _p, _p]),
    "dpcg_convnet_plan_rebuild": (_int, [_p, _int, _i64, _i64, _i64, _p, _int, _p, _p, _p]),
    "dpcg_convnet_plan_destroy": (_int, [_p]),
    "dpcg_convnet_plan_info": (_int, [_p, _int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "dpcg_convnet_plan_output": (_int, [_p, _p, _p, _p, _p]),
    "dpcg_convnet_forward": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _int, _p]),
}


class DpcgError(RuntimeError):
    """A negative dpcg_status crossed the ABI."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libdpcg status {status}: {message}")
        self.status = status


def build(force: bool = False) -> pathlib.Path:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", str(_CSRC)] + (["-B"] if force else []) + ["libdpcg.so"]
    proc = subprocess.run(args, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"building libdpcg.so failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded library.  Raises (never falls back) when the HIP extension is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C deeppreconditioning_amd/csrc` (needs hipcc). There is no CPU fallback.")
        handle = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int) -> int:
    if status < 0:
        msg = lib().dpcg_last_error().decode(errors="replace") or lib().dpcg_status_string(status).decode()
        raise DpcgError(status, msg)
    return status
