"""ctypes binding of libreco_hip.so (the C ABI in include/reco_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails this module raises.
PyTorch is used by the callers only for device memory and streams; the signatures here carry plain
pointers and sizes.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# RECO_HIP_LIB selects another build of the same library (diagnostic / ablation builds); never a different backend
LIB_PATH = os.environ.get("RECO_HIP_LIB") or os.path.join(_HERE, "lib", "libreco_hip.so")

OK, EINVAL, EHIP, ENOMEM, EINDEX = 0, -1, -2, -3, -4
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_CLAMP, ACT_RATING = 0, 1, 2, 3, 4
PREC_F32, PREC_SPLIT_BF16X3 = 0, 1
PROF_TOWER, PROF_GATHER, PROF_LINEAR, PROF_LINEAR_REDUCE, PROF_CSR, PROF_CSR_REDUCE, PROF_TOPK, PROF_TOPK_MERGE, PROF_ENCODER = range(1, 10)
ABI_VERSION = 5


class TowerDesc(Structure):
    """struct reco_tower_desc (include/reco_hip.h)."""
    _fields_ = [
        ("user_gmf", c_void_p), ("item_gmf", c_void_p), ("user_mlp", c_void_p), ("item_mlp", c_void_p),
        ("n_users", c_int64), ("n_items", c_int64),
        ("dg", c_int32), ("dm", c_int32),
        ("n_layers", c_int32), ("layer_out", c_int32 * 3),
        ("W", c_void_p * 3), ("b", c_void_p * 3),
        ("w_predict", c_void_p), ("b_predict", c_void_p),
        ("act", c_int32), ("lo", c_float), ("hi", c_float), ("precision", c_int32),
    ]


_SIGNATURES = {
    "reco_abi_version": (c_int, []),
    "reco_last_error": (c_char_p, []),
    "reco_gather_rows": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "reco_neumf_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                  c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "reco_linear_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                c_int64, c_int64, c_int, c_float, c_float, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "reco_linear_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "reco_csr_linear_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "reco_csr_linear_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_float, c_float,
                                    c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "reco_transpose_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "reco_mlp2_supported": (c_int, [c_int64, c_int64, c_int64]),
    "reco_mlp2_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int,
                              c_float, c_float, c_void_p, c_int64, c_void_p]),
    "reco_autorec_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "reco_autorec_forward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                     c_int, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_tower_create": (c_int, [POINTER(TowerDesc), c_void_p, POINTER(c_void_p)]),
    "reco_tower_destroy": (None, [c_void_p]),
    "reco_tower_set_reserved_cus": (c_int, [c_void_p, c_int]),
    "reco_tower_check_scale": (c_int, [c_void_p, c_void_p, c_void_p]),
    "reco_tower_score_pairs": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "reco_tower_rows_workspace_bytes": (c_size_t, [c_int64]),
    "reco_tower_rows_workspace_bytes_for": (c_size_t, [c_void_p, c_int, c_int64]),
    "reco_tower_score_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "reco_tower_matrix_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "reco_tower_score_matrix": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_tower_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "reco_tower_score_topk": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_tower_score_matrix_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                             c_void_p, c_size_t, c_void_p]),
    "reco_tower_score_topk_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                           c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_tower_score_topk_rows_seen": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                                c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_noalloc_scratch_bytes": (c_size_t, []),
    "reco_tower_topk_fused_workspace_bytes": (c_size_t, [c_void_p, c_int64, c_int64, c_int]),
    "reco_tower_score_topk_fused": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_tower_score_topk_noalloc": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int,
                                              c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "reco_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "reco_topk_rows": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "reco_topk_rows_seen": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "reco_profile_enable": (c_int, [c_int]),
    "reco_profile_enable_kinds": (c_int, [c_int, ctypes.c_uint]),
    "reco_profile_collect": (c_int, [POINTER(c_float), c_int, POINTER(c_int)]),
    "reco_profile_collect_kinds": (c_int, [POINTER(c_float), POINTER(c_int32), c_int, POINTER(c_int)]),
}

EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None


def lib():
    """The loaded library; raises ImportError with build instructions when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "recommender_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C recommender_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.reco_abi_version() != ABI_VERSION:
            raise ImportError("recommender_amd: %s has ABI version %d, this package needs %d; rebuild it (make -C recommender_amd/csrc)"
                              % (LIB_PATH, handle.reco_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


_EXC = {EINVAL: ValueError, EHIP: RuntimeError, ENOMEM: MemoryError, EINDEX: IndexError}


def check(rc):
    if rc != OK:
        msg = lib().reco_last_error()
        raise _EXC.get(rc, RuntimeError)("reco_hip: " + (msg.decode() if msg else "error %d" % rc))
