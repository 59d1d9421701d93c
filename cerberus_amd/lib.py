"""Loader of the HIP extension libcerberus_mfp.so (built in-tree by
cerberus_amd/csrc/Makefile).  Fails loudly when the library is missing: there
is no Python or CPU fallback for any compute entry point."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcerberus_mfp.so")

dp = C.POINTER(C.c_double)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
DT_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)

# every symbol include/cerberus_mfp.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("cmfp_version", C.c_char_p, []),
    ("cmfp_device_count", C.c_int, []),
    ("cmfp_create", C.c_int, [C.POINTER(abi.Geometry), C.POINTER(abi.Params), C.c_int, C.POINTER(abi.StateDesc),
                              C.c_int, C.POINTER(abi.ActionDesc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("cmfp_destroy", C.c_int, [C.c_void_p]),
    ("cmfp_last_error", C.c_char_p, [C.c_void_p]),
    ("cmfp_state_ncomp", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_state_ngrow", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_state_upload", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("cmfp_state_download", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("cmfp_state_device_view", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                         C.POINTER(C.c_longlong * 3), C.POINTER(C.c_int)]),
    ("cmfp_state_set_face_values", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]),
    ("cmfp_state_set_dynamic", C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    ("cmfp_update_begin", C.c_int, [C.c_void_p]),
    ("cmfp_action_get_data", C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    ("cmfp_action_calc_time_derivative", C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double]),
    ("cmfp_action_calc_spatial_derivative", C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]),
    ("cmfp_action_get_allowed_time_step", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("cmfp_state_get_allowed_time_step", C.c_int, [C.c_void_p, C.c_int, dp]),
    ("cmfp_update_download", C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    ("cmfp_est_time_step", C.c_int, [C.c_void_p, dp]),
    ("cmfp_compute_new_dt", C.c_double, [C.c_double, C.c_double]),
    ("cmfp_advance", C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    ("cmfp_advance_steps", C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, dp, dp]),
    ("cmfp_advance_steps_static", C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, dp, dp]),
    ("cmfp_elliptic_project", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("cmfp_elliptic_project_bc", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("cmfp_last_error_cell", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("cmfp_set_fused", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_last_advance_fused", C.c_int, [C.c_void_p]),
    ("cmfp_last_source_fusion", C.c_int, [C.c_void_p]),
    ("cmfp_last_wrapped_dims", C.c_int, [C.c_void_p]),
    ("cmfp_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_profile_read", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), dp]),
    ("cmfp_set_exchange_callback", C.c_int, [C.c_void_p, EXCHANGE_FN, C.c_void_p]),
    ("cmfp_set_exchange_batched", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_set_dt_reduce_callback", C.c_int, [C.c_void_p, DT_REDUCE_FN, C.c_void_p]),
    ("cmfp_copy_async", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]),
    ("cmfp_set_overlap", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_fill_patch", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_halo_count", C.c_longlong, [C.c_void_p, C.c_int, C.c_int]),
    ("cmfp_halo_pack", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("cmfp_halo_unpack", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("cmfp_hbm_copy_probe", C.c_int, [C.c_int, C.c_void_p, C.c_longlong, C.c_int, dp]),
    ("cmfp_peer_init", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("cmfp_peer_open", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("cmfp_peer_set_neighbour", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("cmfp_peer_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("cmfp_peer_enabled", C.c_int, [C.c_void_p]),
    ("cmfp_leaf_reconstruct", C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    ("cmfp_leaf_riemann", C.c_int, [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                    C.c_double, C.c_double, C.c_void_p]),
    ("cmfp_leaf_cons2prim", C.c_int, [C.c_int, C.c_longlong, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    ("cmfp_leaf_prim2cons", C.c_int, [C.c_int, C.c_longlong, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
]

_lib = None


class ExtensionMissing(RuntimeError):
    pass


def load():
    """dlopen the extension and bind every declared symbol (no compute)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            f"{LIB_PATH} is missing: build it with `make -C cerberus_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class CmfpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cmfp error {code}: {msg}")
        self.code = code
