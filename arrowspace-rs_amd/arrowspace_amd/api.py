"""Host-side mirror of the reference's public interface for the lambda-tau path, over the C ABI.

Same names and argument meaning as the Rust crate (paths relative to /root/reference):
  ArrowSpaceBuilder  src/builder.rs:20-455     (with_lambda_graph, with_synthesis, build, ...)
  GraphParams        src/graph.rs:94-102
  GraphLaplacian     src/graph.rs:127-135
  GraphFactory       src/graph.rs:143-204
  TauMode            src/taumode.rs:75-127
  ArrowItem          src/core.rs:84-239
  ArrowSpace         src/core.rs:366-1006      (prepare_query_item, search_lambda_aware, ...)
Errors the reference raises by panicking surface as ArrowSpaceError with the same message.

`build(rows)` runs the reference's stages: clustering pre-stage (compute_optimal_k + incremental clustering,
src/eigenmaps.rs:175-235; `build(rows, centroids)` skips it), optional JL projection of the centroids
(with_dims_reduction), the F x F graph, the per-item lambdas.  Inline sampling (src/sampling.rs) is not on the path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArrowSpaceError, GraphParamsC, ptr

U64_MAX = 0xFFFFFFFFFFFFFFFF


class TauMode:
    """src/taumode.rs:75-82.  TauMode.Median, TauMode.Mean, TauMode.Fixed(t), TauMode.Percentile(p)."""
    FIXED, MEDIAN, MEAN, PERCENTILE = 0, 1, 2, 3

    def __init__(self, kind, arg=0.0):
        self.kind, self.arg = kind, float(arg)

    @staticmethod
    def Fixed(t):
        return TauMode(TauMode.FIXED, t)

    @staticmethod
    def Percentile(p):
        return TauMode(TauMode.PERCENTILE, p)

    def __repr__(self):
        return {0: "Fixed(%r)" % self.arg, 1: "Median", 2: "Mean", 3: "Percentile(%r)" % self.arg}[self.kind]

    def __eq__(self, o):
        return isinstance(o, TauMode) and (self.kind, self.arg) == (o.kind, o.arg)


TauMode.Median = TauMode(TauMode.MEDIAN)
TauMode.Mean = TauMode(TauMode.MEAN)
TAU_FLOOR = 1e-10  # src/taumode.rs:84


class GraphParams:
    """src/graph.rs:94-102 (same field names)."""

    def __init__(self, eps, k, topk, p, sigma=None, normalise=False, sparsity_check=False, rectify=False,
                 self_in_results=False):
        self.eps, self.k, self.topk, self.p = float(eps), int(k), int(topk), float(p)
        self.sigma, self.normalise, self.sparsity_check = sigma, bool(normalise), bool(sparsity_check)
        # the two CosinePair variants of SURVEY.md Appendix A (not fields of the reference's GraphParams: they select
        # which reading of the absent smartcore source the library follows; defaults = the oracle's rule set)
        self.rectify, self.self_in_results = bool(rectify), bool(self_in_results)

    def to_c(self):
        return GraphParamsC(self.eps, self.k, self.topk, self.p, 0.0 if self.sigma is None else float(self.sigma),
                            int(self.sigma is not None), int(self.normalise), int(self.sparsity_check), int(self.rectify),
                            int(self.self_in_results), 0)

    def __eq__(self, o):  # src/graph.rs:105-121 (sparsity_check and topk are not compared there)
        return (self.k == o.k and np.isclose(self.eps, o.eps) and np.isclose(self.p, o.p) and
                ((self.sigma is None) == (o.sigma is None)) and
                (self.sigma is None or np.isclose(self.sigma, o.sigma)) and self.normalise == o.normalise)


class DeviceArray:
    """An HBM buffer owned through the C ABI (as_device_upload / as_device_free), with just enough of the tensor protocol
    (shape, data_ptr, is_contiguous) for this package's marshalling."""
    is_cuda = False   # (not a torch tensor: no torch stream to adopt)

    def __init__(self, ctx, handle, shape, dtype):
        self.ctx, self._h, self.shape, self.dtype = ctx, handle, tuple(int(x) for x in shape), np.dtype(dtype)

    def data_ptr(self):
        return self._h.value

    def is_contiguous(self):
        return True

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        """Row i of a matrix (or element i of a vector) as a host array: one small download."""
        i = int(i)
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        row_shape = self.shape[1:]
        out = np.empty(row_shape, dtype=self.dtype)
        nb = int(np.prod(row_shape, dtype=np.int64)) * self.dtype.itemsize
        self.ctx.check(self.ctx.lib.as_device_download(self.ctx.h, C.c_void_p(self._h.value + i * nb), nb, out.ctypes.data))
        return out

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx.check(self.ctx.lib.as_device_download(self.ctx.h, self._h, out.nbytes, out.ctypes.data))
        return out

    def free(self):
        if self._h is not None and self._h.value:
            self.ctx.lib.as_device_free(self.ctx.h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class Context:
    """One as_ctx: device, stream and workspace.  Not re-entrant (one per host thread)."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.as_ctx_create(int(device), C.byref(h))
        if rc:
            raise ArrowSpaceError(rc, self.lib.as_status_message(rc).decode())
        self.h = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def check(self, rc):
        if rc:
            msg = self.lib.as_last_error(self.h).decode() or self.lib.as_status_message(rc).decode()
            raise ArrowSpaceError(rc, msg)

    def set_stream(self, stream):
        self.check(self.lib.as_ctx_set_stream(self.h, C.c_void_p(stream)))

    def synchronize(self):
        self.check(self.lib.as_ctx_synchronize(self.h))

    def set_option(self, name, value):
        self.check(self.lib.as_ctx_set_option(self.h, name.encode(), int(value)))

    def stat(self, name):
        return int(self.lib.as_ctx_get_stat(self.h, name.encode()))

    def upload(self, array):
        """as_device_upload: a host matrix / vector -> HBM once; the DeviceArray that comes back is accepted wherever rows,
        items, queries or lambdas are (the library uses device pointers in place)."""
        a = np.asarray(array)
        # every data pointer of the C ABI is f64 (u32 only for a row order).  Floating-point input is widened HERE, so that a
        # float32 matrix is never reinterpreted as doubles on the device; integers other than uint32 are refused (an int64
        # permutation silently turned into doubles is a wrong order, not a conversion)
        if a.dtype == np.uint32:
            a = np.ascontiguousarray(a)
        elif a.dtype.kind == "f":
            a = np.ascontiguousarray(a, dtype=np.float64)
        else:
            raise TypeError("upload takes floating-point data (widened to float64) or a uint32 row order, got %s" % a.dtype)
        h = C.c_void_p()
        self.check(self.lib.as_device_upload(self.h, a.ctypes.data, a.nbytes, C.byref(h)))
        return DeviceArray(self, h, a.shape, a.dtype)

    def profile(self, on=True):
        self.lib.as_profile_enable(self.h, int(on))
        self.lib.as_profile_reset(self.h)

    def profile_get(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self.lib.as_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, int(n.value)

    def profile_all(self):
        buf = C.create_string_buffer(4096)
        self.lib.as_profile_names(self.h, buf, 4096)
        return {n: self.profile_get(n) for n in buf.value.decode().split("\n") if n}

    def close(self):
        if self.h:
            self.lib.as_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def _adopt_torch_stream(ctx):
    """Device tensors are produced on PyTorch's current stream: run the library on it too."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    if getattr(ctx, "_stream", "own") != st:
        ctx.set_stream(st)
        ctx._stream = st


def _is_torch(x):
    """A torch tensor (numpy >= 2 arrays have a .device too, and DeviceArray is tensor-like without being one)."""
    return _lib.is_tensor(x) and not isinstance(x, DeviceArray)


def _f64(a, ctx=None):
    if _lib.is_tensor(a):
        if isinstance(a, DeviceArray):
            if a.dtype != np.float64:   # (not an assert: python -O strips those)
                raise TypeError("a DeviceArray handed to the library as rows / items / lambdas must hold float64, got %s" % a.dtype)
        elif str(getattr(a, "dtype", "")) not in ("torch.float64", ""):
            raise TypeError("tensors handed to the library as rows / items / lambdas must be float64, got %s" % a.dtype)
        if a.is_cuda and ctx is not None:
            _adopt_torch_stream(ctx)
        return a
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _shape2(a):
    s = tuple(a.shape)
    if len(s) != 2:
        raise ArrowSpaceError(_lib.AS_ERR_SHAPE, "expected a 2-D matrix")
    return int(s[0]), int(s[1])


class CsrMatrix:
    """Device-resident CSR (as_csr handle) with a lazily exported host view; the counterpart of
    sprs::CsMat<f64> held in GraphLaplacian.matrix."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        self._host = None

    @classmethod
    def from_host(cls, ctx, indptr, indices, data):
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        h = C.c_void_p()
        ctx.check(ctx.lib.as_csr_import(ctx.h, len(indptr) - 1, len(indices), ptr(indptr), ptr(indices), ptr(data),
                                        C.byref(h)))
        return cls(ctx, h)

    def shape(self):
        """(rows, columns): square unless the handle is a row block of a sharded graph."""
        return (int(self.ctx.lib.as_csr_n(self.h)), int(self.ctx.lib.as_csr_cols(self.h)))

    def row_begin(self):
        return int(self.ctx.lib.as_csr_row_begin(self.h))

    def nnz(self):
        return int(self.ctx.lib.as_csr_nnz(self.h))

    def to_host(self):
        if self._host is None:
            n, nnz = self.shape()[0], self.nnz()
            indptr = np.empty(n + 1, dtype=np.uint64)
            indices = np.empty(nnz, dtype=np.uint64)
            data = np.empty(nnz, dtype=np.float64)
            self.ctx.check(self.ctx.lib.as_csr_export(self.ctx.h, self.h, ptr(indptr), ptr(indices), ptr(data)))
            self._host = (indptr, indices, data)
        return self._host

    def indptr(self):
        return self.to_host()[0]

    def indices(self):
        return self.to_host()[1]

    def data(self):
        return self.to_host()[2]

    def set_order(self, order):
        """as_csr_set_order: order[position] = row, an order of the rows in which neighbours sit together (None removes it);
        the item-graph lambda kernels visit the rows in it."""
        if order is None:
            self.ctx.check(self.ctx.lib.as_csr_set_order(self.ctx.h, self.h, None))
            return
        if _lib.is_tensor(order):
            dt = str(getattr(order, "dtype", ""))
            if isinstance(order, DeviceArray):
                if order.dtype != np.uint32:
                    raise TypeError("a row order on the device must hold uint32, got %s" % order.dtype)
            elif dt in ("torch.int64", "torch.int16", "torch.uint8", "torch.int8"):   # e.g. torch.argsort's int64
                import torch
                order = order.to(torch.int32).contiguous()
            elif dt not in ("torch.int32", "torch.uint32"):
                raise TypeError("a row order must be an integer tensor, got %s" % dt)
            self._order_keepalive = order
        else:
            order = np.ascontiguousarray(order, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.as_csr_set_order(self.ctx.h, self.h, ptr(order)))

    def to_dense(self):
        indptr, indices, data = self.to_host()
        n = len(indptr) - 1
        M = np.zeros((n, self.shape()[1]))   # (a row block of a sharded build has global column ids: rows x n_nodes)
        for i in range(n):
            s, e = int(indptr[i]), int(indptr[i + 1])
            M[i, indices[s:e].astype(np.int64)] = data[s:e]
        return M

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.as_csr_free(self.h)
                self.h = None
        except Exception:
            pass


class GraphLaplacian:
    """src/graph.rs:127-135: init_data (the matrix the graph was built from, rows = nodes), the
    CSR Laplacian, nnodes (= item count of the ORIGINAL data, not the matrix dimension) and the
    GraphParams."""

    def __init__(self, init_data, matrix, nnodes, graph_params):
        self.init_data, self.matrix, self.nnodes, self.graph_params = init_data, matrix, nnodes, graph_params

    @staticmethod
    def sparsity(matrix):  # src/graph.rs:572-578
        rows, cols = matrix.shape()            # (a row block: its own rows x all columns)
        return 1.0 - matrix.nnz() / float(rows * cols)


def build_laplacian_matrix(transposed, params, n_items=None, ctx=None):
    """src/laplacian.rs:122-178.  `transposed` is the matrix whose ROWS ARE NODES."""
    ctx = ctx or default_context()
    X = _f64(transposed, ctx)
    if len(X.shape) != 2:
        raise ArrowSpaceError(_lib.AS_ERR_SHAPE, "items should be at least of shape (2,2)")
    n, d = _shape2(X)
    gp = params.to_c()
    h = C.c_void_p()
    ctx.check(ctx.lib.as_build_laplacian(ctx.h, ptr(X), n, d, C.byref(gp), C.byref(h)))
    return GraphLaplacian(X, CsrMatrix(ctx, h), n if n_items is None else int(n_items), params)


def knn_cosine(rows, params, row_begin=0, row_count=None, ctx=None):
    """Stage-level neighbour table (CosinePair::query_row_top_k for a row range)."""
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, d = _shape2(X)
    row_count = n - row_begin if row_count is None else row_count
    K = params.topk + 1
    idx = np.empty((row_count, K), dtype=np.uint64)
    dist = np.empty((row_count, K), dtype=np.float64)
    gp = params.to_c()
    ctx.check(ctx.lib.as_knn_cosine(ctx.h, ptr(X), n, d, row_begin, row_count, C.byref(gp), ptr(idx), ptr(dist)))
    return idx, dist


def nearest_centroids(rows, centroids, ctx=None):
    """Batch form of `nearest_centroid` (src/clustering.rs:913-928): (index of the nearest centroid, squared
    distance) for every row, in the reference's arithmetic (first of equal minima)."""
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    Cn = _f64(centroids, ctx)
    n, d = _shape2(X)
    nc, dc = _shape2(Cn)
    assert d == dc, "rows and centroids must have the same width"
    idx = np.empty(n, dtype=np.uint64)
    d2 = np.empty(n, dtype=np.float64)
    ctx.check(ctx.lib.as_nearest_centroids(ctx.h, ptr(X), n, d, ptr(Cn), nc, ptr(idx), ptr(d2)))
    return idx, d2


def laplacian_from_knn(nbr_idx, nbr_dist, params, ctx=None, row_begin=None, row_count=None):
    """The CSR Laplacian from a complete neighbour table; with a row range, that row block of it (as a row-sharded
    build assembles it: as_laplacian_from_knn_rows)."""
    ctx = ctx or default_context()
    nbr_idx = nbr_idx if _lib.is_tensor(nbr_idx) else np.ascontiguousarray(nbr_idx, dtype=np.uint64)
    nbr_dist = _f64(nbr_dist, ctx)
    n = int(nbr_idx.shape[0])
    gp = params.to_c()
    h = C.c_void_p()
    if row_begin is None:
        ctx.check(ctx.lib.as_laplacian_from_knn(ctx.h, ptr(nbr_idx), ptr(nbr_dist), n, C.byref(gp), C.byref(h)))
    else:
        ctx.check(ctx.lib.as_laplacian_from_knn_rows(ctx.h, ptr(nbr_idx), ptr(nbr_dist), n, C.byref(gp), int(row_begin),
                                                     int(row_count), C.byref(h)))
    return CsrMatrix(ctx, h)


class GraphFactory:
    @staticmethod
    def build_laplacian_matrix_from_k_cluster(clustered, eps, k, topk, p, sigma_override, normalise, sparsity_check,
                                              n_items, ctx=None):
        """src/graph.rs:149-204: transposes the X x F centroid matrix (nodes = features) and
        applies the optional sparsity check."""
        Xc = np.ascontiguousarray(np.asarray(clustered, dtype=np.float64))
        assert Xc.shape[0] <= n_items
        params = GraphParams(eps, k, topk, p, sigma_override, normalise, sparsity_check)
        gl = build_laplacian_matrix(np.ascontiguousarray(Xc.T), params, n_items, ctx=ctx)
        if sparsity_check:
            s = GraphLaplacian.sparsity(gl.matrix)
            if s > 0.95:
                raise ArrowSpaceError(_lib.AS_ERR_TOO_SPARSE, "Resulting laplacian matrix is too sparse %r" % s)
        assert gl.nnodes == n_items
        return gl


def build_spectral_laplacian(aspace, gl):
    """GraphFactory::build_spectral_laplacian (src/graph.rs:212-270): the Laplacian of the Laplacian.  The F x F
    matrix of `gl` is made dense, transposed, and its rows become the nodes of a second build_laplacian_matrix
    with the same GraphParams; the result is stored in aspace.signals, which compute_taumode then prefers
    (queries keep using gl.matrix: src/core.rs:548)."""
    dense_t = np.ascontiguousarray(gl.matrix.to_dense().T)
    sig = build_laplacian_matrix(dense_t, gl.graph_params, aspace.nitems, ctx=aspace.ctx)
    aspace.signals = sig.matrix
    if gl.graph_params is not None and gl.graph_params.sparsity_check and GraphLaplacian.sparsity(sig.matrix) > 0.95:
        raise ArrowSpaceError(_lib.AS_ERR_TOO_SPARSE, "Resulting spectral matrix is too sparse")
    return aspace.signals


# ---- clustering pre-stage (src/clustering.rs) and JL projection (src/reduction.rs) --------------------------------------

CLUSTERING_SEED = 128  # src/clustering.rs:30


def compute_optimal_k(rows, seed=None, ctx=None):
    """ClusteringHeuristic::compute_optimal_k (src/clustering.rs:36-72): (k_optimal, radius, intrinsic dimension)."""
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, f = _shape2(X)
    k, rad, idd = C.c_uint64(), C.c_double(), C.c_uint64()
    ctx.check(ctx.lib.as_compute_optimal_k(ctx.h, ptr(X), n, f, CLUSTERING_SEED if seed is None else int(seed), C.byref(k),
                                           C.byref(rad), C.byref(idd)))
    return int(k.value), float(rad.value), int(idd.value)


def incremental_clustering(rows, max_clusters, radius, batch=4096, ctx=None):
    """run_incremental_clustering_with_sampling without a sampler (src/clustering.rs:547-910): (centroids X x F,
    assignments (-1 = dropped), cluster sizes).  batch = 1 is the reference's sequential mode exactly."""
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, f = _shape2(X)
    cents = np.zeros((max_clusters, f), dtype=np.float64)
    assign = np.empty(n, dtype=np.int64)
    counts = np.zeros(max_clusters, dtype=np.uint64)
    nc = C.c_uint64()
    ctx.check(ctx.lib.as_incremental_clustering(ctx.h, ptr(X), n, f, int(max_clusters), float(radius), int(batch), ptr(cents),
                                                C.byref(nc), ptr(assign), ptr(counts)))
    return cents[: nc.value].copy(), assign, counts[: nc.value].copy()


def two_nn_ratios(rows, sample_ids, ctx=None):
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, f = _shape2(X)
    ids = np.ascontiguousarray(sample_ids, dtype=np.uint64)
    out = np.empty(len(ids), dtype=np.float64)
    ctx.check(ctx.lib.as_two_nn_ratios(ctx.h, ptr(X), n, f, ptr(ids), len(ids), ptr(out)))
    return out


def kmeans_lloyd(rows, k, max_iter, seed, ctx=None):
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, f = _shape2(X)
    assign = np.empty(n, dtype=np.uint64)
    cents = np.zeros((min(k, n), f), dtype=np.float64)
    ctx.check(ctx.lib.as_kmeans_lloyd(ctx.h, ptr(X), n, f, int(k), int(max_iter), int(seed), ptr(assign), ptr(cents)))
    return assign, cents


def calinski_harabasz(rows, assign, k, ctx=None):
    ctx = ctx or default_context()
    X = _f64(rows, ctx)
    n, f = _shape2(X)
    a = np.ascontiguousarray(assign, dtype=np.uint64)
    out = C.c_double()
    ctx.check(ctx.lib.as_calinski_harabasz(ctx.h, ptr(X), n, f, ptr(a), int(k), C.byref(out)))
    return float(out.value)


def compute_jl_dimension(n_points, epsilon):
    """src/reduction.rs:126-140"""
    return int(_lib.load().as_jl_dimension(int(n_points), float(epsilon)))


class ImplicitProjection:
    """src/reduction.rs:166-203: a seeded Gaussian projection, here an as_jl handle (the matrix lives in HBM)."""

    def __init__(self, original_dim, reduced_dim, seed=0, ctx=None):
        self.ctx = ctx or default_context()
        self.original_dim, self.reduced_dim, self.seed = int(original_dim), int(reduced_dim), int(seed)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.as_jl_create(self.ctx.h, self.original_dim, self.reduced_dim, self.seed, C.byref(h)))
        self.h = h

    def matrix(self):
        G = np.empty((self.original_dim, self.reduced_dim), dtype=np.float64)
        self.ctx.check(self.ctx.lib.as_jl_matrix(self.ctx.h, self.h, ptr(G)))
        return G

    def project(self, rows):
        """project_matrix (:142-164) / project (:180-203) for one vector or a batch (host or device rows)."""
        X = _f64(rows, self.ctx)
        single = len(X.shape) == 1
        if single:
            X = X.reshape(1, -1)
        n, d = _shape2(X)
        assert d == self.original_dim
        if _is_torch(X):
            import torch
            out = torch.empty((n, self.reduced_dim), dtype=torch.float64, device=X.device)
        else:
            out = np.empty((n, self.reduced_dim), dtype=np.float64)
        self.ctx.check(self.ctx.lib.as_jl_project(self.ctx.h, self.h, ptr(X), n, ptr(out)))
        return out[0] if single else out

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.as_jl_destroy(self.h)
                self.h = None
        except Exception:
            pass


class ArrowItem:
    """src/core.rs:84-239."""

    def __init__(self, item, lambda_):
        self.item = np.asarray(item, dtype=np.float64)
        self.lambda_ = float(lambda_)

    def __len__(self):
        return len(self.item)


def select_tau(values, mode, ctx=None):
    """TauMode::select_tau (src/taumode.rs:87-127) for one slice or a batch of rows."""
    ctx = ctx or default_context()
    v = np.asarray(values, dtype=np.float64)
    single = v.ndim == 1
    v = np.ascontiguousarray(v.reshape(1, -1) if single else v)
    out = np.empty(v.shape[0], dtype=np.float64)
    ctx.check(ctx.lib.as_select_tau(ctx.h, ptr(v) if v.size else None, v.shape[0], v.shape[1], mode.kind, mode.arg,
                                    ptr(out)))
    return float(out[0]) if single else out


def compute_synthetic_lambda(x, matrix, tau, ctx=None):
    """TauMode::compute_synthetic_lambda_csr (src/taumode.rs:552-660) for one vector."""
    ctx = ctx or matrix.ctx
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(1, -1))
    out = np.empty(1, dtype=np.float64)
    ctx.check(ctx.lib.as_compute_taumode(ctx.h, ptr(x), 1, x.shape[1], matrix.h, TauMode.FIXED, float(tau), ptr(out)))
    return float(out[0])


def compute_taumode_signals(items, matrix, taumode=None, ctx=None):
    """Lambdas of the feature COLUMNS of `items` (N x F) as signals over the item graph `matrix` (n x n, n <= N): for every
    column f, compute_synthetic_lambda_csr(items[:, f], matrix, select_tau(items[:, f])) (src/taumode.rs:552-660, 87-127) --
    TauMode::compute_taumode_lambdas_parallel on the transposed matrix, without the transpose.  Returns F values (a torch
    tensor in HBM when the items are one)."""
    ctx = ctx or matrix.ctx
    taumode = taumode or TauMode.Median
    X = _f64(items, ctx)
    n, f = _shape2(X)
    if _is_torch(X):
        import torch
        out = torch.empty(f, dtype=torch.float64, device=X.device)
    else:
        out = np.empty(f, dtype=np.float64)
    ctx.check(ctx.lib.as_compute_taumode_signals(ctx.h, ptr(X), n, f, matrix.h, taumode.kind, taumode.arg, ptr(out)))
    return out


class ArrowSpace:
    """src/core.rs:366-1006 restricted to the lambda-tau path: dense N x F data, per-item
    lambdas, taumode."""

    def __init__(self, data, taumode=TauMode.Median, ctx=None):
        self.ctx = ctx or default_context()
        self.data = _f64(data, self.ctx)
        self.nitems, self.nfeatures = _shape2(self.data)
        self._resident = None  # HBM copy of host data (Context.upload), made at the first library call and kept: the
        #                        reference keeps its items in host memory and walks them three times (src/builder.rs:249-304);
        #                        here lambdas, queries' index and search all read the one device copy
        self.taumode = taumode
        self.lambdas_ = None
        self.signals = None  # F x F spectral Laplacian (src/core.rs:372), set by build_spectral_laplacian
        self._index = None
        self.id_offset = 0
        # clustering metadata and projection state (src/core.rs:373-384), set by the builder
        self.n_clusters, self.cluster_assignments, self.cluster_sizes, self.cluster_radius = 0, None, None, 0.0
        self.projection_matrix, self.reduced_dim = None, None

    def lambdas(self):
        return self.lambdas_

    def _items(self):
        """The items as the library should see them: in place when they are in HBM already, else uploaded once."""
        if _lib.is_tensor(self.data) or self.data.nbytes < (1 << 20):   # (torch tensor or DeviceArray: in HBM already)
            return self.data
        if self._resident is None:
            self._resident = self.ctx.upload(self.data)
        return self._resident

    def get_item(self, i):  # src/core.rs:586-594
        assert i < self.nitems, "item index out of bounds"
        row = self.data[i]
        row = row.cpu().numpy() if hasattr(row, "cpu") else row
        return ArrowItem(np.array(row), float(self.lambdas_[i]))

    def compute_taumode(self, gl):
        """EigenMaps::compute_taumode (src/eigenmaps.rs:358-383) ->
        TauMode::compute_taumode_lambdas_parallel (src/taumode.rs:174-312)."""
        lib, ctx = self.ctx.lib, self.ctx
        if _is_torch(self.data):   # a torch tensor in HBM: the lambdas stay there too
            import torch
            out = torch.empty(self.nitems, dtype=torch.float64, device=self.data.device)
        else:
            out = np.empty(self.nitems, dtype=np.float64)
        # graph = aspace.signals if non-empty else gl.matrix (src/taumode.rs:195-200)
        graph = self.signals if self.signals is not None else gl.matrix
        ctx.check(lib.as_compute_taumode(ctx.h, ptr(self._items()), self.nitems, self.nfeatures, graph.h,
                                         self.taumode.kind, self.taumode.arg, ptr(out)))
        self.update_lambdas(out)

    def update_lambdas(self, lambdas):  # src/core.rs:979-1005
        assert len(lambdas) == self.nitems
        self.lambdas_ = lambdas
        self._drop_index()

    def prepare_query_item(self, item, gl):
        """src/core.rs:533-549: returns the query's lambda."""
        return float(self.prepare_query_items(np.asarray(item, dtype=np.float64).reshape(1, -1), gl)[0])

    def prepare_query_items(self, queries, gl):
        lib, ctx = self.ctx.lib, self.ctx
        Q = _f64(queries, ctx)
        nq, f = _shape2(Q)
        if f != self.nfeatures:  # src/core.rs:510-516
            raise ArrowSpaceError(_lib.AS_ERR_DIM, "Query dimension %d doesn't match index original dimension %d"
                                  % (f, self.nfeatures))
        if _is_torch(Q):
            import torch
            out = torch.empty(nq, dtype=torch.float64, device=Q.device)
        else:
            out = np.empty(nq, dtype=np.float64)
        if self.projection_matrix is not None:  # src/core.rs:541-546: the query is projected first
            if not np.all(np.isfinite(Q.numpy() if isinstance(Q, DeviceArray) else (Q.cpu().numpy() if _lib.is_tensor(Q) else Q))):
                raise ArrowSpaceError(_lib.AS_ERR_NONFINITE, ctx.lib.as_status_message(_lib.AS_ERR_NONFINITE).decode())
            Q = self.project_query(Q)
            f = self.reduced_dim
        ctx.check(lib.as_prepare_query_items(ctx.h, ptr(Q), nq, f, gl.matrix.h, self.taumode.kind, self.taumode.arg,
                                             ptr(out)))
        return out

    def project_query(self, query):
        """src/core.rs:509-529"""
        if self.projection_matrix is None:
            return query
        return self.projection_matrix.project(query)

    def _ensure_index(self):
        if self._index is None:
            if self.lambdas_ is None:  # src/eigenmaps.rs:425-431
                raise ArrowSpaceError(_lib.AS_ERR_ARG, "call compute_taumode(...) before search")
            h = C.c_void_p()
            lam = self.lambdas_ if _lib.is_tensor(self.lambdas_) else np.ascontiguousarray(self.lambdas_, np.float64)
            self._lam_keep = lam
            self.ctx.check(self.ctx.lib.as_index_create(self.ctx.h, ptr(self._items()), ptr(lam), self.nitems,
                                                        self.nfeatures, C.byref(h)))
            self._index = h
        return self._index

    def _drop_index(self):
        if getattr(self, "_index", None):
            self.ctx.lib.as_index_destroy(self._index)
        self._index = None

    def search_lambda_aware(self, query, k, alpha):
        """src/core.rs:760-798: Vec<(usize, f64)> of the k best (index, score)."""
        idx, sc = self.search_lambda_aware_batch(query.item.reshape(1, -1), np.array([query.lambda_]), k, alpha)
        return [(int(i), float(s)) for i, s in zip(idx[0], sc[0]) if int(i) != U64_MAX]

    def search_lambda_aware_batch(self, queries, q_lambdas, k, alpha):
        lib, ctx = self.ctx.lib, self.ctx
        ix = self._ensure_index()
        Q = _f64(queries, ctx)
        nq, f = _shape2(Q)
        assert f == self.nfeatures, "items should be of the same length"  # src/core.rs:157-161
        QL = _f64(q_lambdas, ctx)
        kk = min(int(k), self.nitems)
        if _is_torch(Q):
            import torch
            idx = torch.empty((nq, kk), dtype=torch.int64, device=Q.device)
            sc = torch.empty((nq, kk), dtype=torch.float64, device=Q.device)
        else:
            idx = np.empty((nq, kk), dtype=np.uint64)
            sc = np.empty((nq, kk), dtype=np.float64)
        if kk > 0 and nq > 0:
            ctx.check(lib.as_search_lambda_aware(ctx.h, ix, ptr(Q), ptr(QL), nq, kk, float(alpha), self.id_offset,
                                                 ptr(idx), ptr(sc)))
        return idx, sc

    def search_lambda_aware_hybrid(self, query, k, alpha):
        """src/core.rs:802-928: lambda-aware top-k united with the near-exact cosine matches."""
        if k == 0:
            return []
        lib, ctx = self.ctx.lib, self.ctx
        ix = self._ensure_index()
        Q = np.ascontiguousarray(np.asarray(query.item, dtype=np.float64).reshape(1, -1))
        QL = np.array([query.lambda_], dtype=np.float64)
        idx = np.empty((1, k), dtype=np.uint64)
        sc = np.empty((1, k), dtype=np.float64)
        ctx.check(lib.as_search_lambda_aware_hybrid(ctx.h, ix, ptr(Q), ptr(QL), 1, int(k), float(alpha), self.id_offset,
                                                    ptr(idx), ptr(sc)))
        return [(int(i), float(s)) for i, s in zip(idx[0], sc[0]) if int(i) != U64_MAX]

    def range_search(self, query, gl, eps):
        """src/core.rs:944-976: items with query.lambda - lambda_i <= eps, in index order."""
        lq = query.lambda_
        if abs(lq) <= 1e-9:  # relative_eq!(query.lambda, 0.0, epsilon = 1e-9): prepare it first
            lq = self.prepare_query_item(query.item, gl)
        lib, ctx = self.ctx.lib, self.ctx
        ix = self._ensure_index()
        idx = np.empty(self.nitems, dtype=np.uint64)
        dist = np.empty(self.nitems, dtype=np.float64)
        found = C.c_uint64()
        ctx.check(lib.as_range_search(ctx.h, ix, float(lq), float(eps), self.id_offset, ptr(idx), ptr(dist), self.nitems,
                                      C.byref(found)))
        m = int(found.value)
        return [(int(i), float(d)) for i, d in zip(idx[:m], dist[:m])]

    def search(self, item, gl, k, alpha):
        """EigenMaps::search (src/eigenmaps.rs:410-455): prepare + search in one call."""
        lam = self.prepare_query_item(item, gl)
        return self.search_lambda_aware(ArrowItem(item, lam), k, alpha)

    def __del__(self):
        try:
            self._drop_index()
        except Exception:
            pass


class ArrowSpaceBuilder:
    """src/builder.rs:20-455 (the lambda-tau subset)."""

    def __init__(self, ctx=None):
        # defaults: src/builder.rs:59-91
        self.lambda_eps, self.lambda_k, self.lambda_topk, self.lambda_p = 1e-3, 6, 3, 2.0
        self.lambda_sigma = None
        self.normalise = False
        self.sparsity_check = False
        self.synthesis = TauMode.Median
        self.prebuilt_spectral = False
        self.ctx = ctx
        # clustering / projection (src/builder.rs:44-57,80-90)
        self.cluster_max_clusters, self.cluster_radius = None, 1.0
        self.clustering_seed, self.deterministic_clustering = None, False
        self.clustering_batch = 4096
        self.use_dims_reduction, self.rp_eps = False, 0.3

    @staticmethod
    def new(ctx=None):
        return ArrowSpaceBuilder(ctx)

    def with_lambda_graph(self, eps, k, topk, p, sigma_override=None):  # src/builder.rs:109-137
        self.lambda_eps, self.lambda_k, self.lambda_topk = eps, k, topk
        self.lambda_p, self.lambda_sigma = p, sigma_override
        return self

    def with_synthesis(self, tau_mode):  # :141-145
        self.synthesis = tau_mode
        return self

    def with_normalisation(self, normalise):
        self.normalise = bool(normalise)
        return self

    def with_spectral(self, compute_spectral):  # src/builder.rs:157-162
        self.prebuilt_spectral = bool(compute_spectral)
        return self

    def with_sparsity_check(self, flag):
        self.sparsity_check = bool(flag)
        return self

    def with_inline_sampling(self, sampler):
        if sampler is not None:
            raise ArrowSpaceError(_lib.AS_ERR_UNSUPPORTED, "inline sampling is outside the accelerated path")
        return self

    def with_seed(self, seed):  # src/builder.rs:191-195: fixes the seed and makes the clustering sequential
        self.clustering_seed = int(seed)
        self.deterministic_clustering = True
        return self

    def with_dims_reduction(self, enable, eps=None):  # src/builder.rs:164-172
        self.use_dims_reduction = bool(enable)
        if eps is not None:
            self.rp_eps = float(eps)
        return self

    def start_clustering(self, aspace):
        """EigenMaps::start_clustering (src/eigenmaps.rs:175-290): optimal K + radius, incremental clustering, optional
        JL projection of the centroids.  Returns the (possibly projected) centroids."""
        ctx = aspace.ctx
        k_opt, radius, _ = compute_optimal_k(aspace._items(), self.clustering_seed, ctx=ctx)
        self.cluster_max_clusters, self.cluster_radius = k_opt, radius
        batch = 1 if self.deterministic_clustering else self.clustering_batch
        cents, assign, sizes = incremental_clustering(aspace._items(), k_opt, radius, batch, ctx=ctx)
        aspace.n_clusters, aspace.cluster_assignments, aspace.cluster_sizes = len(cents), assign, sizes
        aspace.cluster_radius = radius
        return self._maybe_project(aspace, cents)

    def _maybe_project(self, aspace, cents):
        nf = aspace.nfeatures
        if self.use_dims_reduction and nf > 64:  # src/eigenmaps.rs:247-280
            target = min(compute_jl_dimension(len(cents), self.rp_eps), nf // 2)
            if target < nf:
                proj = ImplicitProjection(nf, target, seed=self.clustering_seed or CLUSTERING_SEED, ctx=aspace.ctx)
                aspace.projection_matrix, aspace.reduced_dim = proj, target
                return proj.project(cents)
        return cents

    def define_result_k(self):  # src/builder.rs:225-233
        self.lambda_topk = int(_lib.load().as_define_result_k(self.lambda_k, self.lambda_topk))

    def graph_params(self):
        return GraphParams(self.lambda_eps, self.lambda_k, self.lambda_topk, self.lambda_p, self.lambda_sigma,
                           self.normalise, self.sparsity_check)

    def build(self, rows, centroids=None):
        """src/builder.rs:249-455.  Stage 1: clustering (skipped when the centroids, X x F, are supplied) + optional JL
        projection; stage 2: eigenmaps (Laplacian over centroids^T); stage 3 (optional): spectral; stage 4: taumode."""
        ctx = self.ctx or default_context()
        self.define_result_k()
        aspace = ArrowSpace(rows, self.synthesis, ctx)
        if centroids is None:
            centroids = self.start_clustering(aspace)       # stage 1, src/builder.rs:296-300
        else:
            centroids = self._maybe_project(aspace, np.ascontiguousarray(np.asarray(centroids, dtype=np.float64)))
        gl = GraphFactory.build_laplacian_matrix_from_k_cluster(
            centroids, self.lambda_eps, self.lambda_k, self.lambda_topk, self.lambda_p, self.lambda_sigma,
            self.normalise, self.sparsity_check, aspace.nitems, ctx=ctx)
        if self.prebuilt_spectral:  # stage 3, src/eigenmaps.rs:325-345
            build_spectral_laplacian(aspace, gl)
        aspace.compute_taumode(gl)
        return aspace, gl

    def build_item_graph(self, rows):
        """build_laplacian_matrix on the untransposed rows (items are nodes), as
        src/energymaps.rs:270 and benches/base_laplacian_bench.rs:62-70 call it."""
        ctx = self.ctx or default_context()
        self.define_result_k()
        return build_laplacian_matrix(rows, self.graph_params(), None, ctx=ctx)
