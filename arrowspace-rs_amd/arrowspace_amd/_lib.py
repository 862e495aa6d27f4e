"""ctypes binding of libarrowspace_hip.so (the C ABI in include/arrowspace_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950 device is present the
calls fail loudly (ArrowSpaceError / OSError).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG), "lib", "libarrowspace_hip.so")

(AS_OK, AS_ERR_SHAPE, AS_ERR_NONFINITE, AS_ERR_ZERO_LAMBDA, AS_ERR_NAN_SCORE, AS_ERR_DIM, AS_ERR_TOO_SPARSE,
 AS_ERR_ARG, AS_ERR_UNSUPPORTED, AS_ERR_HIP, AS_ERR_NO_DEVICE, AS_ERR_PEER) = range(12)

EXPORTS = [
    "as_abi_version", "as_ctx_create", "as_ctx_destroy", "as_last_error", "as_status_message", "as_ctx_set_stream",
    "as_ctx_synchronize", "as_ctx_set_option", "as_ctx_get_stat", "as_profile_enable", "as_profile_reset",
    "as_profile_get", "as_profile_names", "as_define_result_k", "as_build_laplacian", "as_knn_cosine",
    "as_knn_join_thresholds", "as_knn_join_capacity", "as_knn_join_pairs", "as_knn_join_finish",
    "as_laplacian_from_knn", "as_csr_n", "as_csr_nnz", "as_csr_export", "as_csr_import", "as_csr_free",
    "as_compute_taumode", "as_prepare_query_items", "as_select_tau", "as_index_create", "as_index_destroy",
    "as_index_nitems", "as_search_lambda_aware", "as_search_lambda_aware_hybrid", "as_range_search",
    "as_nearest_centroids", "as_knn_join_pack", "as_knn_join_unpack", "as_laplacian_from_knn_rows", "as_csr_cols",
    "as_csr_row_begin", "as_knn_join_applies", "as_two_nn_ratios", "as_kmeans_lloyd", "as_calinski_harabasz",
    "as_compute_optimal_k", "as_incremental_clustering", "as_jl_dimension", "as_jl_create", "as_jl_destroy",
    "as_jl_original_dim", "as_jl_reduced_dim", "as_jl_matrix", "as_jl_project",
    "as_comm_unique_id", "as_comm_create_rccl", "as_comm_create_custom", "as_comm_destroy", "as_comm_rank", "as_comm_world",
    "as_shard_range", "as_build_laplacian_sharded", "as_lambda_stats_sharded", "as_search_lambda_aware_sharded",
    "as_device_upload", "as_device_download", "as_device_free",
    "as_compute_taumode_signals", "as_csr_set_order", "as_compute_taumode_signals_sharded",
]

# transport callbacks of as_comm_create_custom (device buffers; byte counts / offsets are arrays of `world` entries)
AllGatherV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                         C.c_void_p)
AllToAllV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p,
                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)
AllReduceSumU64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class CommOpsC(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather_v", AllGatherV), ("all_to_all_v", AllToAllV),
                ("all_reduce_sum_u64", AllReduceSumU64)]


class GraphParamsC(C.Structure):
    _fields_ = [("eps", C.c_double), ("k", C.c_uint64), ("topk", C.c_uint64), ("p", C.c_double),
                ("sigma", C.c_double), ("has_sigma", C.c_int32), ("normalise", C.c_int32),
                ("sparsity_check", C.c_int32), ("rectify", C.c_int32), ("self_in_results", C.c_int32),
                ("reserved", C.c_int32)]


class ArrowSpaceError(RuntimeError):
    """Raised where the reference panics; .status is the as_status code."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libarrowspace_hip.so not built (%s): run `make -C arrowspace-rs_amd` or "
                      "__graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, dbl, i32 = C.c_void_p, C.c_uint64, C.c_double, C.c_int
    sig = {
        "as_abi_version": (i32, []),
        "as_ctx_create": (i32, [i32, C.POINTER(vp)]),
        "as_ctx_destroy": (None, [vp]),
        "as_last_error": (C.c_char_p, [vp]),
        "as_status_message": (C.c_char_p, [i32]),
        "as_ctx_set_stream": (i32, [vp, vp]),
        "as_ctx_synchronize": (i32, [vp]),
        "as_ctx_set_option": (i32, [vp, C.c_char_p, C.c_int64]),
        "as_ctx_get_stat": (C.c_int64, [vp, C.c_char_p]),
        "as_profile_enable": (i32, [vp, i32]),
        "as_profile_reset": (i32, [vp]),
        "as_profile_get": (i32, [vp, C.c_char_p, C.POINTER(dbl), C.POINTER(u64)]),
        "as_profile_names": (i32, [vp, C.c_char_p, u64]),
        "as_define_result_k": (u64, [u64, u64]),
        "as_build_laplacian": (i32, [vp, vp, u64, u64, C.POINTER(GraphParamsC), C.POINTER(vp)]),
        "as_knn_cosine": (i32, [vp, vp, u64, u64, u64, u64, C.POINTER(GraphParamsC), vp, vp]),
        "as_knn_join_thresholds": (i32, [vp, vp, u64, u64, u64, u64, C.POINTER(GraphParamsC), vp]),
        "as_knn_join_capacity": (C.c_uint32, [vp, C.POINTER(GraphParamsC)]),
        "as_knn_join_applies": (i32, [vp, u64, u64, C.POINTER(GraphParamsC)]),
        "as_knn_join_pairs": (i32, [vp, vp, u64, u64, u64, u64, C.POINTER(GraphParamsC), vp, C.c_uint32, vp, vp]),
        "as_knn_join_finish": (i32, [vp, vp, u64, u64, u64, u64, C.POINTER(GraphParamsC), vp, C.c_uint32, C.c_uint32, vp, vp,
                                     vp, vp]),
        "as_laplacian_from_knn": (i32, [vp, vp, vp, u64, C.POINTER(GraphParamsC), C.POINTER(vp)]),
        "as_laplacian_from_knn_rows": (i32, [vp, vp, vp, u64, C.POINTER(GraphParamsC), u64, u64, C.POINTER(vp)]),
        "as_csr_cols": (u64, [vp]),
        "as_csr_row_begin": (u64, [vp]),
        "as_csr_n": (u64, [vp]),
        "as_csr_nnz": (u64, [vp]),
        "as_csr_export": (i32, [vp, vp, vp, vp, vp]),
        "as_csr_import": (i32, [vp, u64, u64, vp, vp, vp, C.POINTER(vp)]),
        "as_csr_free": (None, [vp]),
        "as_compute_taumode": (i32, [vp, vp, u64, u64, vp, i32, dbl, vp]),
        "as_prepare_query_items": (i32, [vp, vp, u64, u64, vp, i32, dbl, vp]),
        "as_compute_taumode_signals": (i32, [vp, vp, u64, u64, vp, i32, dbl, vp]),
        "as_csr_set_order": (i32, [vp, vp, vp]),
        "as_compute_taumode_signals_sharded": (i32, [vp, vp, vp, u64, u64, vp, i32, dbl, vp]),
        "as_select_tau": (i32, [vp, vp, u64, u64, i32, dbl, vp]),
        "as_index_create": (i32, [vp, vp, vp, u64, u64, C.POINTER(vp)]),
        "as_index_destroy": (None, [vp]),
        "as_index_nitems": (u64, [vp]),
        "as_search_lambda_aware": (i32, [vp, vp, vp, vp, u64, u64, dbl, u64, vp, vp]),
        "as_search_lambda_aware_hybrid": (i32, [vp, vp, vp, vp, u64, u64, dbl, u64, vp, vp]),
        "as_range_search": (i32, [vp, vp, dbl, dbl, u64, vp, vp, u64, C.POINTER(u64)]),
        "as_nearest_centroids": (i32, [vp, vp, u64, u64, vp, u64, vp, vp]),
        "as_knn_join_pack": (i32, [vp, vp, vp, u64, C.c_uint32, vp, vp]),
        "as_two_nn_ratios": (i32, [vp, vp, u64, u64, vp, u64, vp]),
        "as_kmeans_lloyd": (i32, [vp, vp, u64, u64, u64, u64, u64, vp, vp]),
        "as_calinski_harabasz": (i32, [vp, vp, u64, u64, vp, u64, C.POINTER(dbl)]),
        "as_compute_optimal_k": (i32, [vp, vp, u64, u64, u64, C.POINTER(u64), C.POINTER(dbl), C.POINTER(u64)]),
        "as_incremental_clustering": (i32, [vp, vp, u64, u64, u64, dbl, u64, vp, C.POINTER(u64), vp, vp]),
        "as_jl_dimension": (u64, [u64, dbl]),
        "as_jl_create": (i32, [vp, u64, u64, u64, C.POINTER(vp)]),
        "as_jl_destroy": (None, [vp]),
        "as_jl_original_dim": (u64, [vp]),
        "as_jl_reduced_dim": (u64, [vp]),
        "as_jl_matrix": (i32, [vp, vp, vp]),
        "as_jl_project": (i32, [vp, vp, vp, u64, vp]),
        "as_knn_join_unpack": (i32, [vp, vp, vp, u64, C.c_uint32, vp]),
        "as_comm_unique_id": (i32, [vp]),
        "as_comm_create_rccl": (i32, [vp, vp, i32, i32, C.POINTER(vp)]),
        "as_comm_create_custom": (i32, [vp, C.POINTER(CommOpsC), i32, i32, C.POINTER(vp)]),
        "as_comm_destroy": (None, [vp]),
        "as_comm_rank": (i32, [vp]),
        "as_comm_world": (i32, [vp]),
        "as_shard_range": (None, [u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]),
        "as_build_laplacian_sharded": (i32, [vp, vp, vp, u64, u64, C.POINTER(GraphParamsC), C.POINTER(vp), C.POINTER(u64)]),
        "as_device_upload": (i32, [vp, vp, u64, C.POINTER(vp)]),
        "as_device_download": (i32, [vp, vp, u64, vp]),
        "as_device_free": (None, [vp, vp]),
        "as_lambda_stats_sharded": (i32, [vp, vp, vp, u64, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]),
        "as_search_lambda_aware_sharded": (i32, [vp, vp, vp, vp, vp, u64, u64, dbl, u64, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def ptr(x):
    """Raw address of a numpy array (host) or a torch tensor (host or HBM)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous(), "tensor must be contiguous"
        return x.data_ptr()
    raise TypeError("expected numpy array or torch tensor, got %r" % type(x))


def is_tensor(x):
    return hasattr(x, "data_ptr") and not isinstance(x, np.ndarray)
