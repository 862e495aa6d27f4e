"""arrowspace_amd: host-side mirror of the arrowspace lambda-tau API over libarrowspace_hip.so.

The compute lives in the HIP library (arrowspace-rs_amd/csrc, C ABI in include/arrowspace_hip.h);
this package only marshals pointers.  There is no CPU fallback.
"""
from ._lib import ArrowSpaceError, LIB_PATH, load  # noqa: F401
from .api import (ArrowItem, ArrowSpace, ArrowSpaceBuilder, Context, CsrMatrix, GraphFactory,  # noqa: F401
                  GraphLaplacian, GraphParams, TauMode, TAU_FLOOR, build_laplacian_matrix, build_spectral_laplacian, nearest_centroids,
                  compute_synthetic_lambda, default_context, knn_cosine, laplacian_from_knn, select_tau,
                  ImplicitProjection, calinski_harabasz, compute_jl_dimension, compute_optimal_k, incremental_clustering,
                  kmeans_lloyd, two_nn_ratios)
