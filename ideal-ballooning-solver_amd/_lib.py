"""ctypes binding of lib/libibs_hip.so (C ABI: include/ibs.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded this module raises,
and every compute call raises when no GPU/context is available.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IBS_LIB_PATH") or os.path.join(_HERE, "lib", "libibs_hip.so")   # (override: compiler-flag experiments)

MEM_DEVICE = 0
MEM_HOST = 1

# every symbol include/ibs.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_D = C.c_double
_I32 = C.c_int32
_I64 = C.c_int64
SYMBOLS = {
    "ibs_version": (C.c_int, []),
    "ibs_last_error": (C.c_char_p, []),
    "ibs_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "ibs_destroy": (C.c_int, [_P]),
    "ibs_set_stream": (C.c_int, [_P, _P]),
    "ibs_synchronize": (C.c_int, [_P]),
    "ibs_set_option": (C.c_int, [_P, C.c_char_p, _D]),
    "ibs_comm_load": (C.c_int, [C.c_char_p]),
    "ibs_comm_unique_id": (C.c_int, [_P]),
    "ibs_comm_init": (C.c_int, [_P, _P, _I32, _I32]),
    "ibs_comm_allgather_f64": (C.c_int, [_P, _P, _P, _I64]),
    "ibs_comm_allgather_start_f64": (C.c_int, [_P, _P, _P, _I64, _I32, _I32]),
    "ibs_comm_wait": (C.c_int, [_P, _I32]),
    "ibs_comm_destroy": (C.c_int, [_P]),
    "ibs_lbfgsb2_state_bytes": (C.c_int, []),
    "ibs_lbfgsb2_init": (C.c_int, [_P, _P, _P, _P, _D, _D, _I32, _I32]),
    "ibs_lbfgsb2_step": (C.c_int, [_P, _D, _P, _P]),
    "ibs_lbfgsb2_result": (C.c_int, [_P, _P, _P, _P]),
    "ibs_device_count": (C.c_int, []),
    "ibs_last_launch": (C.c_int, [C.c_char_p, _I32, C.POINTER(_I64), C.POINTER(_I32)]),
    "ibs_solve_gcf_f64": (C.c_int, [_P, _I64, _I32, _D, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _I32]),
    "ibs_solve_gcfh_f64": (C.c_int, [_P, _I64, _I32, _D, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _I32]),
    "ibs_solve_gcf_f32": (C.c_int, [_P, _I64, _I32, C.c_float, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _I32]),
    "ibs_gamma_scan_f64": (C.c_int, [_P, _I32, _I32, _I32, _D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _I32]),
    "ibs_gamma_scan_warm_f64": (C.c_int, [_P, _I32, _I32, _I32, _D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _D,
                                          _P, _P, _P, _P, _P, _P, _I32]),
    "ibs_gamma_scan_argmax_f64": (C.c_int, [_P, _I32, _I32, _I32, _D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I32,
                                            _P, _P, _P, _P]),
    "ibs_gamma_points_f64": (C.c_int, [_P, _I32, _I32, _D, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I32]),
    "ibs_scan_starts_f64": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "ibs_obj_w_grad_f64": (C.c_int, [_P, _I32, _I32, _D, _P, _I64, _P, _D, _P, _P, _P, _I32]),
    "ibs_hf_grad_f64": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I32]),
    "ibs_fieldline_geometry_f64": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _I32, _P,
                                             _I64, _P, _P, _I32, _P, _I32, _P, _D, _D, _I32]),
    "ibs_surface_tables_f64": (C.c_int, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32]),
    "ibs_refine_f64": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _I32, _P, _D, _D, _I32, _P, _P,
                                 _I32, _P, _D, _I32, _D, _D, _P, _P, _P, _I32]),
    "ibs_refine_stats": (C.c_int, [_P, _P]),
    "ibs_sturm_count_f64": (C.c_int, [_P, _I64, _I32, _D, _P, _P, _P, _I64, _P, _P, _I32]),
    "ibs_surface_argmax_f64": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _I32]),
    "ibs_surface_argmax_pack_f64": (C.c_int, [_P, _I32, _I32, _P, _P]),
}


class IbsError(RuntimeError):
    pass


def _preload_hip_runtime():
    """libibs_hip.so carries no DT_NEEDED on libamdhip64 (csrc/Makefile): the process must hold ONE
    HIP runtime.  With PyTorch present that is PyTorch's bundled copy (two runtimes in one process
    cannot both open the GPU); otherwise the ROCm installation's."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands += [os.path.join(root, "lib", "libamdhip64.so.7"), os.path.join(root, "lib", "libamdhip64.so")]
    for c in cands:
        if os.path.exists(c):
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
    raise IbsError("no HIP runtime (libamdhip64.so) found; tried %s" % cands)


def load():
    if not os.path.exists(LIB_PATH):
        raise IbsError(
            "HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C ideal-ballooning-solver_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def check(rc, what):
    if rc < 0:
        raise IbsError("%s failed (%d): %s" % (what, rc, lib().ibs_last_error().decode()))
    return rc
