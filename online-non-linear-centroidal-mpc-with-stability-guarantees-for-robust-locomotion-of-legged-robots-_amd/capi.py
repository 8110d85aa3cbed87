"""ctypes binding of the C ABI in include/cmpc.h (libcmpc_amd.so, built by build.py).

There is no CPU fallback: if the HIP library is missing or fails to load, ``load()``
raises, and so does every solver entry point.
"""
import ctypes
import os

from .problem import CSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMPC_LIB_PATH: developer knob for A/B measurements of two builds of the HIP library in one GPU session
# (tools/ab_bench.sh); there is still no fallback of any kind
LIB_PATH = os.environ.get("CMPC_LIB_PATH") or os.path.join(_HERE, "libcmpc_amd.so")

#: every symbol include/cmpc.h declares
SYMBOLS = ("cmpc_default_spec", "cmpc_create", "cmpc_destroy", "cmpc_workspace_bytes",
           "cmpc_solve_batch", "cmpc_solve_batch_state", "cmpc_last_kernel_ms", "cmpc_last_kernel_name", "cmpc_last_error",
           "cmpc_version",
           "cmpc_tables_create", "cmpc_tables_destroy", "cmpc_build_records",
           "cmpc_tables_set_plan_slots", "cmpc_build_records_planned")
#: every symbol include/cmpc_wbc.h declares (batched whole-body QP, same library)
WBC_SYMBOLS = ("cmpc_wbc_qp_solve_batch", "cmpc_wbc_last_error")

_lib = None


def load():
    """Load libcmpc_amd.so (raises OSError with a build hint when it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build the HIP extension first "
                      f"(python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(LIB_PATH)
    c_spec_p = ctypes.POINTER(CSpec)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.cmpc_default_spec.argtypes = [c_spec_p, i32, i32]
    lib.cmpc_default_spec.restype = None
    lib.cmpc_create.argtypes = [c_spec_p, ctypes.c_int, ctypes.POINTER(vp)]
    lib.cmpc_create.restype = ctypes.c_int
    lib.cmpc_destroy.argtypes = [vp]
    lib.cmpc_destroy.restype = ctypes.c_int
    lib.cmpc_workspace_bytes.argtypes = [c_spec_p, i32]
    lib.cmpc_workspace_bytes.restype = ctypes.c_size_t
    lib.cmpc_solve_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.cmpc_solve_batch.restype = ctypes.c_int
    lib.cmpc_solve_batch_state.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cmpc_solve_batch_state.restype = ctypes.c_int
    lib.cmpc_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.cmpc_last_kernel_ms.restype = ctypes.c_int
    lib.cmpc_last_kernel_name.argtypes = [vp]
    lib.cmpc_last_kernel_name.restype = ctypes.c_char_p
    lib.cmpc_last_error.argtypes = [vp]
    lib.cmpc_last_error.restype = ctypes.c_char_p
    lib.cmpc_tables_create.argtypes = [ctypes.c_int, i32] + [vp] * 7 + [ctypes.POINTER(vp)]
    lib.cmpc_tables_create.restype = ctypes.c_int
    lib.cmpc_tables_destroy.argtypes = [vp]
    lib.cmpc_tables_destroy.restype = ctypes.c_int
    lib.cmpc_build_records.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.cmpc_build_records.restype = ctypes.c_int
    lib.cmpc_tables_set_plan_slots.argtypes = [vp, i32, vp, vp]
    lib.cmpc_tables_set_plan_slots.restype = ctypes.c_int
    lib.cmpc_build_records_planned.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.cmpc_build_records_planned.restype = ctypes.c_int
    f64 = ctypes.c_double
    lib.cmpc_wbc_qp_solve_batch.argtypes = [ctypes.c_int, i32, vp, vp, vp, vp, vp, f64, f64, f64, i32, vp, vp, vp, vp, vp, vp]
    lib.cmpc_wbc_qp_solve_batch.restype = ctypes.c_int
    lib.cmpc_wbc_last_error.argtypes = []
    lib.cmpc_wbc_last_error.restype = ctypes.c_char_p
    lib.cmpc_version.argtypes = []
    lib.cmpc_version.restype = ctypes.c_char_p
    _lib = lib
    return lib
