"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Bars (BASELINE.json north_star): neighbour / adjacency / search indices bit-exact;
lambda-tau and scores within 1e-6 relative (asserted far tighter: 1e-12).

Every test here needs a real MI355X: run with `pytest -m gpu`.
"""
import numpy as np
import pytest

try:  # before the library: both bring a HIP runtime, and the tests that hand over device tensors need torch's to
    import torch  # noqa: F401   be the one that initialises the device (whatever subset of tests is selected)
except ImportError:
    torch = None

import arrowspace_amd as A
from arrowspace_amd import _lib
from oracle import oracle_ctypes as O
from tests import synth

pytestmark = pytest.mark.gpu

NAN, INF = float("nan"), float("inf")
LAMBDA_RTOL = 1e-12   # north_star allows 1e-6
WEIGHT_RTOL = 1e-13   # kernel weights: device pow/x*x vs glibc pow differ by <= 1 ulp


@pytest.fixture(scope="module")
def ctx():
    return A.default_context()


@pytest.fixture(params=[0, 32, 64], ids=["exact", "mfma_f32", "mfma_f64"])
def mode(request, ctx):
    """exact scan only / f32 MFMA candidates (then f64, then exact) / f64 MFMA candidates first"""
    ctx.set_option("knn_mfma", int(request.param != 0))
    ctx.set_option("search_mfma", int(request.param != 0))
    ctx.set_option("cand_precision", request.param or 32)
    yield request.param
    ctx.set_option("knn_mfma", 1)
    ctx.set_option("search_mfma", 1)
    ctx.set_option("cand_precision", 32)


def ogp(p):
    return O.graph_params(p.eps, p.k, p.topk, p.p, p.sigma)


def assert_csr_equal(got, want, rtol=WEIGHT_RTOL):
    gi, gj, gd = got
    wi, wj, wd = want
    assert np.array_equal(gi, wi), "indptr differs"
    assert np.array_equal(gj, wj), "column indices differ"
    np.testing.assert_allclose(gd, wd, rtol=rtol, atol=0)


# ---- select_tau: the reference's exact table (src/tests/test_taumode.rs:13-159) ---------------

def test_select_tau_table(ctx):
    T = A.TauMode
    e = [0.1, 0.5, 1.0]
    assert A.select_tau(e, T.Fixed(0.3)) == 0.3
    for bad in (-0.1, 0.0, NAN, INF):
        assert A.select_tau(e, T.Fixed(bad)) == A.TAU_FLOOR
    assert abs(A.select_tau([1.0, 2.0, 3.0], T.Mean) - 2.0) < 1e-12
    assert abs(A.select_tau([1.0, NAN, 3.0, INF, 2.0], T.Mean) - 2.0) < 1e-12
    assert A.select_tau([NAN, INF, -INF], T.Mean) == A.TAU_FLOOR
    assert A.select_tau([], T.Mean) == A.TAU_FLOOR
    assert A.select_tau([3.0, 1.0, 2.0], T.Median) == 2.0
    assert abs(A.select_tau([1.0, 2.0, 3.0, 4.0], T.Median) - 2.5) < 1e-12
    assert A.select_tau([5.0], T.Median) == 5.0
    assert A.select_tau([NAN, 1.0, 3.0, INF, 2.0], T.Median) == 2.0
    assert A.select_tau([NAN, INF], T.Median) == A.TAU_FLOOR
    assert A.select_tau([], T.Median) == A.TAU_FLOOR
    e = [1.0, 2.0, 3.0, 4.0, 5.0]
    for p, want in ((0.0, 1.0), (1.0, 5.0), (0.5, 3.0), (-0.1, 1.0), (1.5, 5.0)):
        assert A.select_tau(e, T.Percentile(p)) == want
    assert A.select_tau([], T.Percentile(0.5)) == A.TAU_FLOOR
    assert A.select_tau([A.TAU_FLOOR * 2.0], T.Mean) == A.TAU_FLOOR * 2.0
    assert A.select_tau([A.TAU_FLOOR / 2.0], T.Mean) == A.TAU_FLOOR
    assert A.select_tau([0.0], T.Mean) == A.TAU_FLOOR


@pytest.mark.parametrize("length", [1, 2, 7, 24, 64, 128, 129, 384, 768, 1000])
def test_select_tau_matches_oracle(ctx, length):
    rng = np.random.default_rng(length)
    X = rng.standard_normal((37, length))
    X[3, : length // 2] = X[3, 0]                     # duplicates
    if length > 4:
        X[5, 1], X[5, 2], X[6, 0] = NAN, -INF, INF    # non-finite entries are filtered
        X[7] = -np.abs(X[7])                          # all negative -> floor
    for m, om, arg in ((A.TauMode.Median, O.TAU_MEDIAN, 0.0), (A.TauMode.Mean, O.TAU_MEAN, 0.0),
                       (A.TauMode.Percentile(0.95), O.TAU_PERCENTILE, 0.95),
                       (A.TauMode.Percentile(0.25), O.TAU_PERCENTILE, 0.25), (A.TauMode.Fixed(0.2), O.TAU_FIXED, 0.2)):
        got = A.select_tau(X, m)
        want = np.array([O.select_tau(r, om, arg) for r in X])
        if om == O.TAU_MEAN:
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)  # sum order differs (lanes split the row)
        else:
            assert np.array_equal(got, want), (length, m)


# ---- lambda: scale invariance (src/tests/test_taumode.rs:498-528) -----------------------------

def test_rayleigh_scale_invariance(ctx):
    L = A.CsrMatrix.from_host(ctx, [0, 2, 4], [0, 1, 0, 1], [1.0, 0.5, 0.5, 1.0])
    l1 = A.compute_synthetic_lambda([1.0, 2.0], L, 0.5)
    l2 = A.compute_synthetic_lambda([2.0, 4.0], L, 0.5)
    assert l1 == pytest.approx(l2, rel=1e-10)
    assert l1 == pytest.approx(0.5 * (1.4 / 1.9), rel=1e-15)


# ---- graph: golden vectors -------------------------------------------------------------------

def test_golden_graphs(ctx, golden, mode):
    for name, c in golden.items():
        p = c["params"]
        gp = A.GraphParams(p["eps"], p["k"], p["topk"], p["p"], p["sigma"])
        idx, dist = A.knn_cosine(c["nodes"], gp)
        want_idx = c["knn_idx"].astype(np.int64)
        got_idx = idx.astype(np.int64)
        got_idx[idx == np.uint64(A.api.U64_MAX)] = -1
        assert np.array_equal(got_idx, want_idx), name
        assert np.array_equal(dist[want_idx >= 0], c["knn_dist"][want_idx >= 0]), name
        gl = A.build_laplacian_matrix(c["nodes"], gp)
        assert_csr_equal(gl.matrix.to_host(), c["csr"])
        assert gl.matrix.shape() == (c["nodes"].shape[0],) * 2


def test_golden_lambdas_and_search(ctx, golden, mode):
    for name, c in golden.items():
        if "lambdas" not in c:
            continue
        L = A.CsrMatrix.from_host(ctx, *c["csr"])
        gl = A.GraphLaplacian(c["nodes"], L, len(c["rows"]), None)
        tm = A.TauMode(c["tau_mode"], c["tau_arg"])
        asp = A.ArrowSpace(c["rows"], tm, ctx)
        asp.compute_taumode(gl)
        np.testing.assert_allclose(asp.lambdas(), c["lambdas"], rtol=LAMBDA_RTOL, atol=1e-300, err_msg=name)
        for q in c.get("queries", []):
            lq = asp.prepare_query_item(q["q"], gl)
            assert lq == pytest.approx(q["lambda_q"], rel=LAMBDA_RTOL), name
            if "idx" in q:
                # identical inputs (the oracle's lambdas) -> identical indices AND scores
                asp.update_lambdas(c["lambdas"])
                res = asp.search_lambda_aware(A.ArrowItem(q["q"], q["lambda_q"]), q["k"], q["alpha"])
                assert [r[0] for r in res] == q["idx"], name
                assert np.array_equal([r[1] for r in res], q["scores"]), name


def test_proteins_alpha1_top3(ctx, proteins, mode):
    """paper.md:124-133 / benches/index_compute_bench.rs:121-126 anchor."""
    asp = A.ArrowSpace(proteins, A.TauMode.Median, ctx)
    asp.update_lambdas(np.full(len(proteins), 0.25))
    res = asp.search_lambda_aware(A.ArrowItem(proteins[3] * 1.02, 0.25), 3, 1.0)
    assert [r[0] for r in res] == [3, 6, 0]
    assert ["%.6f" % r[1] for r in res] == ["1.000000", "0.999573", "0.999325"]


# ---- graph: reference error behaviour and literal-matrix invariants --------------------------

def test_too_small_input_panics(ctx):  # src/tests/test_laplacian.rs:302-311
    gp = A.GraphParams(0.5, 3, 2, 2.0, 0.1)
    with pytest.raises(A.ArrowSpaceError, match=r"items should be at least of shape \(2,2\)") as e:
        A.build_laplacian_matrix(np.ones((1, 5)), gp)
    assert e.value.status == _lib.AS_ERR_SHAPE
    with pytest.raises(A.ArrowSpaceError):
        A.build_laplacian_matrix(np.ones((5, 1)), gp)


def test_basic_laplacian_construction(ctx):  # src/tests/test_laplacian.rs:34-49
    items = np.array([[1, 0, 0], [.8, .6, 0], [0, 1, 0], [0, .8, .6], [0, 0, 1]], dtype=np.float64)
    params = A.GraphParams(0.5, 3, 2, 2.0, 0.1, False, True)
    gl = A.build_laplacian_matrix(np.ascontiguousarray(items.T), params, None)
    assert gl.matrix.shape() == (3, 3)
    gl5 = A.build_laplacian_matrix(np.ascontiguousarray(items.T), params, 5)
    assert gl5.nnodes == 5 and gl5.graph_params == params
    M = gl.matrix.to_dense()
    assert np.allclose(M.sum(axis=1), 0, atol=1e-12) and np.allclose(M, M.T, atol=1e-12)
    assert np.all(np.diag(M) >= -1e-12) and np.all(M[~np.eye(3, dtype=bool)] <= 1e-12)


def test_sparsity_check_panics(ctx):  # src/graph.rs:185-193
    X, cent = synth.mixture(400, 64, seed=3, n_clusters=8, rank=4)
    with pytest.raises(A.ArrowSpaceError, match="too sparse"):
        A.GraphFactory.build_laplacian_matrix_from_k_cluster(synth.gaussian_rows(40, 64, seed=2), 1e-3, 6, 4, 2.0,
                                                             None, False, True, 400)


def test_query_errors(ctx, proteins):
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(proteins)
    q = proteins[0].copy()
    q[3] = NAN
    with pytest.raises(A.ArrowSpaceError, match="invalid values") as e:  # src/tests/test_querying_proj.rs:277-291
        asp.prepare_query_item(q, gl)
    assert e.value.status == _lib.AS_ERR_NONFINITE
    with pytest.raises(A.ArrowSpaceError, match="doesn't match index original dimension"):
        asp.prepare_query_item(proteins[0][:10], gl)
    with pytest.raises(A.ArrowSpaceError, match="Lambda of the item is 0.0"):  # src/core.rs:773-776
        asp.search_lambda_aware(A.ArrowItem(proteins[0], 0.0), 3, 0.7)
    # k > nitems -> nitems results, sorted descending (src/tests/test_querying_proj.rs:369-399)
    res = asp.search_lambda_aware(A.ArrowItem(proteins[0], 0.1), 1000, 0.7)
    assert len(res) == len(proteins)
    assert all(res[i][1] >= res[i + 1][1] for i in range(len(res) - 1))


# ---- seeded synthetic sets vs the oracle, both mappings --------------------------------------

@pytest.mark.parametrize("n,d,topk,eps", [(1500, 64, 16, 0.5), (777, 33, 5, 0.3), (2048, 128, 16, 0.5), (300, 130, 3, 0.6),
                                          (1100, 200, 8, 0.5), (640, 768, 16, 0.5), (1300, 96, 30, 0.5), (900, 40, 45, 0.6)])
def test_item_graph_matches_oracle(ctx, mode, n, d, topk, eps):
    X, _ = synth.mixture(n, d, seed=11, n_clusters=16, rank=4)
    gp = A.GraphParams(eps, 16, topk, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    gl = A.build_laplacian_matrix(X, gp)
    assert_csr_equal(gl.matrix.to_host(), O.build_laplacian(X, ogp(gp)))


@pytest.fixture
def join_on(ctx):
    """The symmetric threshold join (csrc/mfma_join_f32.hip) at test sizes (it starts at 32,768 rows by default)."""
    ctx.set_option("join_min_rows", 512)
    yield
    ctx.set_option("join_min_rows", 32768)
    for name, v in (("join_sample", 0), ("join_rank", 0), ("join_cap", 0)):  # 0 = automatic
        ctx.set_option(name, v)


@pytest.mark.parametrize("n,d,topk,eps", [(1500, 64, 16, 0.5), (777, 33, 5, 0.3), (2048, 128, 16, 0.5), (1200, 128, 16, 0.5),
                                          (1300, 96, 30, 0.5), (900, 40, 45, 0.6), (513, 128, 3, 0.4), (6000, 100, 16, 0.5),
                                          (1100, 200, 8, 0.5), (640, 768, 16, 0.5), (2100, 384, 16, 0.5), (1500, 129, 16, 0.5)])
def test_item_graph_join_matches_oracle(ctx, join_on, n, d, topk, eps):
    """Odd and even counts of 256-row tiles (the cyclic half pattern has an antipodal tile only when even), ragged
    last tiles, list capacities from K + 1 up, rows wider than 128 (k panels: the A fragments are re-read per panel)."""
    X, _ = synth.mixture(n, d, seed=11, n_clusters=16, rank=4)
    gp = A.GraphParams(eps, 16, topk, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, ogp(gp)))


@pytest.mark.parametrize("n,d", [(1500, 64), (2048, 128), (3100, 17), (5000, 100), (1024, 128)])
@pytest.mark.parametrize("pipe", [16, 48, 32])
def test_item_graph_join_on_either_pipe(ctx, join_on, n, d, pipe):
    """The same join contracted from one f16 plane (512-row super tiles), as split-bf16 products (two-phase blocks) and
    on the f32 MFMA pipe: each certifies against its own rounding bound and gives the oracle's tables; rows that can
    form a 0 < |a||b| <= 1e-12 pair and zero rows included."""
    X, _ = synth.mixture(n, d, seed=13, n_clusters=10, rank=4)
    X[5] = 0.0
    X[40] *= 1e-8
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    ctx.set_option("join_pipe", pipe)
    try:
        idx, dist = A.knn_cosine(X, gp)
        blocks = ctx.stat("knn_join_blocks")
        assert ctx.stat("knn_join_pipe") == pipe
    finally:
        ctx.set_option("join_pipe", 16)
    assert ctx.stat("knn_join") == 1
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    assert blocks > 0
    if pipe == 48:  # the kernel reports what it contracted: every block on the hi planes, some of them in full
        assert 0 <= ctx.stat("knn_mfma_join_bf16_blocks_full") <= blocks
    if pipe == 16:  # ... and the f16 form how many blocks listed a pair, and the bound it certified against
        assert 0 <= ctx.stat("knn_mfma_join_f16_blocks_listed") <= blocks
        assert 0 < ctx.stat("knn_join_delta_e9") <= 1.1e6  # <= the worst-case bound of one f16 plane, 1.02e-3 at dim 128


@pytest.mark.parametrize("n,d", [(1100, 200), (640, 768), (2100, 384), (1500, 129), (3000, 300), (5200, 144)])
@pytest.mark.parametrize("pipe", [16, 32])
def test_item_graph_join_wide_rows_on_either_pipe(ctx, join_on, n, d, pipe):
    """Rows wider than 128: the panelled f16 form (mfma_join_f16_wide.hip: 256-row super tiles, accumulators over the
    whole k range, zero-padded to whole 64-wide panels) and the f32 MFMA form give the oracle's tables; zero rows and
    rows that can form a tiny-norm pair included."""
    X, _ = synth.mixture(n, d, seed=17, n_clusters=10, rank=4)
    X[5] = 0.0
    X[40] *= 1e-8
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    ctx.set_option("join_pipe", pipe)
    try:
        idx, dist = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_pipe") == pipe
        assert ctx.stat("knn_join_blocks") > 0
    finally:
        ctx.set_option("join_pipe", 16)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


@pytest.mark.parametrize("n,d", [(1300, 256), (70_000, 384), (900, 1024)])
def test_item_graph_image_of_wide_rows_in_one_pass(ctx, join_on, n, d):
    """Rows of a multiple of 128 columns: the f16 image and the plane error of every row come out of ONE pass over the rows, a
    workgroup per (128-row tile, 128 columns) ("join_pack_wide_tiles", round 5, mfma_join_f16.hip) -- the same stored values as the
    per-fragment-lane packer, the same error measure summed in another order.  The tables are those of the older pair of passes
    and the oracle's; zero rows, a tiny row and a row count that is no multiple of the tile included, and (70,000 rows) the image in
    the library's own row order."""
    X, _ = synth.mixture(n, d, seed=23, n_clusters=12, rank=4)
    X[5] = 0.0
    X[40] *= 1e-8
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_pack_wide_tiles") == 1
    delta = ctx.stat("knn_join_delta_e9")
    ctx.set_option("join_pack_wide_tiles", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_pack_wide_tiles") == 0
        assert abs(ctx.stat("knn_join_delta_e9") - delta) <= 2  # the same measure (units of 1e-9), another order of the sum
    finally:
        ctx.set_option("join_pack_wide_tiles", 1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.unique(np.concatenate([[5, 40, n - 1], np.random.default_rng(4).choice(n, size=min(n, 60), replace=False)]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d", [(70_000, 128), (131_075, 128), (70_000, 72), (66_000, 40)])
def test_item_graph_threshold_sweeps_with_two_row_groups_per_wave(ctx, join_on, n, d):
    """The threshold sweeps (grouped sample, own neighbourhood, checks, second pass) on four waves of 64 rows per 256-row unit --
    every fragment read feeds two MFMAs, the m-th largest bin maximum comes from a sorting network ("join_theta_rows64", round 6,
    mfma_theta_f16.hip: theta_f16x2_kernel) -- give the thresholds of the eight-waves-of-32 form bit for bit: the same candidate
    count, the same rows left open, the same tables, the oracle's rows.  Odd and even numbers of k steps, a row count that is no
    multiple of 256."""
    X, _ = synth.mixture(n, d, seed=31, n_clusters=48, rank=6)
    X[11] = 0.0
    X[9000:9012] = X[123]
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1
    cand, open_rows = ctx.stat("knn_join_candidates"), ctx.stat("knn_uncertified_rows")
    ctx.set_option("join_theta_rows64", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_candidates") == cand and ctx.stat("knn_uncertified_rows") == open_rows
    finally:
        ctx.set_option("join_theta_rows64", 1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.unique(np.concatenate([[11, 123, 9000, 9011, n - 1], np.random.default_rng(6).choice(n, size=80, replace=False)]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,clusters", [(70_001, 40), (90_000, 3)])
def test_item_graph_pruning_with_a_row_per_half_wave(ctx, join_on, n, clusters):
    """The pruning of the re-rank with a row per half wave and the list in registers ("rerank_prune_halves", round 6,
    rerank_split.h: rerank_prune_halves_kernel; lists longer than 128 entries go on to the LDS form) leaves the records of the
    one-row-per-wave form: the same rows open, the same tables, the oracle's rows.  An odd row count (a half without a row), few
    wide clusters (long lists), zero rows and duplicates."""
    X, _ = synth.mixture(n, 128, seed=37, n_clusters=clusters, rank=6)
    X[11] = 0.0
    X[5000:5040] = X[123]
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_rerank_pair_bound") == 1
    open_rows, long_lists = ctx.stat("knn_uncertified_rows"), ctx.stat("knn_rerank_long_lists")
    assert long_lists >= (1 if clusters == 3 else 0)
    ctx.set_option("rerank_prune_halves", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_uncertified_rows") == open_rows
    finally:
        ctx.set_option("rerank_prune_halves", 1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.unique(np.concatenate([[11, 123, 5000, 5039, n - 1], np.random.default_rng(8).choice(n, size=80, replace=False)]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n", [70_000, 131_075])
def test_item_graph_coordinates_and_image_from_one_pass_over_the_rows(ctx, join_on, n):
    """Rows of exactly 128 values: the rotated coordinates (on the matrix pipe) and the f16 form of the rows come out of ONE pass
    over the rows ("join_fused_rows_pass", round 6, rotate_wide.hip: zproj_pack_rows128_kernel); the image in the order of the join
    is re-arranged from the row-major f16 rows once the order is known.  The same stored values and the same plane errors as the
    packer's own pass: the tables are those of the two passes and the oracle's, zero rows, a tiny row, duplicates and a row count
    that is no multiple of 64 included."""
    X, _ = synth.mixture(n, 128, seed=29, n_clusters=40, rank=6)
    X[5] = 0.0
    X[40] *= 1e-8
    X[7000:7020] = X[77]
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_row_order") == 1 and ctx.stat("knn_join_fused_rows_pass") == 1
    delta = ctx.stat("knn_join_delta_e9")
    ctx.set_option("join_fused_rows_pass", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_fused_rows_pass") == 0
        assert ctx.stat("knn_join_delta_e9") == delta   # the same measure, added up in the same order
    finally:
        ctx.set_option("join_fused_rows_pass", 1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.unique(np.concatenate([[5, 40, 77, 7000, 7019, n - 1], np.random.default_rng(4).choice(n, size=80, replace=False)]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d", [(70_000, 256), (66_000, 384)])
def test_item_graph_rerank_of_wide_rows_in_the_order_of_the_image(ctx, join_on, n, d):
    """Rows of a multiple of 128 columns (the re-rank on records) on a reordered image: the lists hold row ids, but the re-rank lays
    its records out in the order of the image ("rerank_image_order", round 5) so that a row's partners -- its neighbours in that order -- are records and rows the
    same L2 has just seen (1 M x 768: link 7.4 -> 4.2 ms, scoring 41 -> 30 ms).  Same tables as with the records in row order, rows
    with duplicates (ties by row id), a zero row and a sample of oracle rows included."""
    X, _ = synth.mixture(n, d, seed=37, n_clusters=30, rank=5)
    X[6000:6030] = X[91]
    X[9] = 0.0
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_row_order") == 1 and ctx.stat("knn_rerank_image_order") == 1
    open_rows, open_ordered = ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_open_order")
    assert open_ordered == (1 if open_rows > 64 else 0)  # ("join_open_order": the rows left open go on in the order of the image too)
    for option, stat in (("rerank_image_order", "knn_rerank_image_order"), ("join_open_order", "knn_join_open_order")):
        ctx.set_option(option, 0)
        try:
            idx0, dist0 = A.knn_cosine(X, gp)
            assert ctx.stat(stat) == 0
        finally:
            ctx.set_option(option, 1)
        assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.concatenate([[9, 91, 6000, 6029], np.sort(np.random.default_rng(5).choice(n, size=40, replace=False))])
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d", [(70_000, 128), (66_000, 64), (131_072, 100)])
def test_item_graph_join_on_a_reordered_image(ctx, n, d):
    """From 65,536 rows on, the operand image of the join is packed with the rows grouped by direction (row_order.hip): at 128
    columns the lists stay in image positions and the pruning kernel translates them, at other widths the join kernel does
    where it appends.  Either way the tables are those of the run without the reordering, rows with exact duplicates (ties
    by row id) and a sample of oracle rows included."""
    X, _ = synth.mixture(n, d, seed=31, n_clusters=40, rank=5)
    X[5000:5040] = X[77]          # duplicates: equal distances, order by row id
    X[9] = 0.0
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_row_order") == 1
    ctx.set_option("join_row_order", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_row_order") == 0
    finally:
        ctx.set_option("join_row_order", 1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.concatenate([[9, 77, 5000, 5039], np.sort(np.random.default_rng(2).choice(n, size=40, replace=False))])
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d,mode,isotropic", [(3000, 128, 2, False), (3000, 128, 3, False), (2600, 128, 3, True),
                                                  (2600, 128, 1, True), (4100, 100, 2, False), (4100, 100, 3, False),
                                                  (1500, 48, 2, False), (1500, 48, 3, False), (2000, 24, 1, False),
                                                  (1027, 33, 3, False), (5000, 64, 1, False), (1100, 200, 3, False),
                                                  (2100, 384, 1, False), (640, 768, 3, False), (1500, 130, 3, True),
                                                  (1300, 1024, 1, False)])
def test_item_graph_join_behind_the_int8_first_pass(ctx, join_on, n, d, mode, isotropic):
    """The int8 first pass (csrc/mfma_prefilter_i8.hip) decides which blocks the f16 join contracts at all: full-width image
    (mode 2), the rotated 32-wide one (31 dominant directions + the tail norm; mode 3 forces it, also on isotropic rows for
    which its bound is useless but must still hold), the automatic choice (mode 1), rows of one, two and four int8 k steps,
    rows wider than 128 (directions by subspace iteration on the device), ragged tiles, duplicates and a zero row.  The tables are the oracle's and those of the run without the first pass."""
    if isotropic:
        X = np.random.default_rng(5).standard_normal((n, d))
    else:
        X, _ = synth.mixture(n, d, seed=23, n_clusters=9, rank=4)
    X[40:48] = X[3]
    X[11] = 0.0
    gp = A.GraphParams(0.6, 16, 12, 2.0, 0.25)
    ctx.set_option("join_prefilter_min_rows", 0)
    ctx.set_option("join_prefilter", mode)
    try:
        idx, dist = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_pipe") == 16
        used = ctx.stat("knn_join_prefilter")
        rotated = ctx.stat("knn_join_prefilter_rotated")
        if d > 128:  # the panelled kernel only takes the rotated form (rotate_wide.hip), reduced to pairs of 256-row super
            assert used == 1 or mode == 1  # tiles; left to itself it passes when the 31 directions hold too little
            assert rotated == 1 or used == 0
        else:
            assert used == 1
            if d <= 32 or mode == 2:
                assert rotated == 0
            elif mode == 3:
                assert rotated == 1
            elif isotropic:
                assert rotated == 1  # 31 of 128 directions of white noise hold a quarter of the energy: the pass rules
                # nothing out there, but it costs a quarter of the full-width one, so it is taken at any tail (round 3)
        if used:
            flagged, blocks = ctx.stat("knn_join_prefilter_blocks_flagged"), ctx.stat("knn_join_prefilter_blocks")
            assert 0 < flagged <= blocks // 2
        ctx.set_option("join_prefilter", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_prefilter") == 0
    finally:
        ctx.set_option("join_prefilter", 1)
        ctx.set_option("join_prefilter_min_rows", 65536)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


@pytest.mark.parametrize("n,d,topk", [(70000, 128, 16), (66001, 64, 40), (131072 + 300, 128, 5), (70144, 256, 16)])
def test_item_graph_first_pass_behind_the_cone_bound(ctx, n, d, topk):
    """"join_cone_tiles" (round 4, cone_tiles.hip): the int8 first pass sweeps only the pairs of 128-position tiles a one-sided cone
    bound keeps.  The bound is rigorous, so the first pass behind it must find what it found sweeping everything -- the same
    blocks up to the few int8 false positives that lay in tile pairs the bound rules out -- the lists must leave the same rows
    open, and the tables are the same and the oracle's.  Ragged tiles, an odd and an even count of super tiles, duplicates, a
    zero row, rows wider than 128 (their first pass runs on the same kernel)."""
    X, _ = synth.mixture(n, d, seed=31, n_clusters=48, rank=5)
    X[40:48] = X[3]
    X[11] = 0.0
    gp = A.GraphParams(0.6, 16, topk, 2.0, 0.25)
    ctx.set_option("join_cone_tiles_max_e3", 1000)   # (taken whatever share of the tile pairs the bound keeps)
    try:
        ctx.set_option("join_cone_tiles", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_prefilter") == 1 and ctx.stat("knn_join_cone_tiles") == 0
        flagged0, open0, cand0 = ctx.stat("knn_join_prefilter_blocks_flagged"), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_candidates")
        ctx.set_option("join_cone_tiles", 1)
        idx, dist = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_cone_tiles") == 1 and 0 < ctx.stat("knn_join_cone_tiles_kept_e6") <= 1_000_000
        flagged1, open1, cand1 = ctx.stat("knn_join_prefilter_blocks_flagged"), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_candidates")
    finally:
        ctx.set_option("join_cone_tiles", 1)
        ctx.set_option("join_cone_tiles_max_e3", 600)
    assert flagged1 <= flagged0 and flagged0 - flagged1 <= max(8, flagged0 // 500), (flagged0, flagged1)
    assert open1 <= open0 + max(2, open0 // 100) and abs(cand1 - cand0) <= max(16, cand0 // 1000), (open0, open1, cand0, cand1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(n, size=128, replace=False))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d,clusters,rank", [(262144 + 4500, 64, 60, 5), (300_000, 96, 900, 16)])
def test_item_graph_cone_bound_with_a_coarse_level(ctx, n, d, clusters, rank):
    """"join_cone_groups" (round 5, cone_tiles.hip): from 2 048 tiles on, the n^2 pass of the cone bound first meets every tile's rows
    with the cones around 32 tiles (direction: the normalised sum of the tiles' mean directions, half angle: the largest angle to
    a tile's mean plus that tile's own) and skips the groups such a cone excludes (4 M x 128: 9.7 -> 3.6 ms).  A bound on a superset
    of the rows: the pairs it keeps are at most those of the one-level pass, the lists leave no more rows open than a few, and the
    tables are the same and the oracle's.  A ragged last group of tiles, duplicates, a zero row."""
    X, _ = synth.mixture(n, d, seed=41, n_clusters=clusters, rank=rank)
    X[50:58] = X[7]
    X[13] = 0.0
    gp = A.GraphParams(0.6, 16, 16, 2.0, 0.25)
    ctx.set_option("join_cone_tiles_max_e3", 1000)   # (taken whatever share of the tile pairs the bound keeps)
    try:
        ctx.set_option("join_cone_groups", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join_cone_tiles") == 1 and ctx.stat("knn_join_cone_groups") == 0
        kept0, split0, open0 = ctx.stat("knn_join_cone_tiles_kept_e6"), ctx.stat("knn_join_cone_split"), ctx.stat("knn_uncertified_rows")
        ctx.set_option("join_cone_groups", 1)
        idx, dist = A.knn_cosine(X, gp)
        kept1, split1, open1 = ctx.stat("knn_join_cone_tiles_kept_e6"), ctx.stat("knn_join_cone_split"), ctx.stat("knn_uncertified_rows")
        assert ctx.stat("knn_join_cone_tiles") == 1
    finally:
        ctx.set_option("join_cone_groups", 1)
        ctx.set_option("join_cone_tiles_max_e3", 600)
    assert split0 == split1
    if not split1:  # (where the split form ran at once the whole-tile pass, and with it the coarse level, did not)
        assert ctx.stat("knn_join_cone_groups") == 1 and 0 < kept1 <= kept0, (kept0, kept1)
    assert open1 <= open0 + max(2, open0 // 100), (open0, open1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    sample = np.unique(np.concatenate([[7, 13, 50, 57, n - 1], np.random.default_rng(6).choice(n, size=60, replace=False)]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d,clusters,rank", [(131072 + 300, 128, 700, 24), (300_000, 64, 1500, 32), (70144, 256, 420, 24)])
def test_item_graph_row_order_draws_more_pivots_where_the_first_set_covers_too_little(ctx, n, d, clusters, rank):
    """"join_order_cover" (round 5, row_order.hip): where many rows lie far from their pivot -- data with more clusters than the
    first set has pivots -- further pivots are drawn from those rows (1 024 a round; beyond 1 024 pivots the assignment runs in
    passes, the second case).  Any set of pivots gives a correct image order: the tables with and without the growth are the same
    and the oracle's; the statistics say the growth ran, and that it left fewer rows uncovered.  Rows wider than 128 order their
    image by the same code (third case)."""
    X, _ = synth.mixture(n, d, seed=37, n_clusters=clusters, rank=rank)
    X[40:48] = X[3]
    X[11] = 0.0
    gp = A.GraphParams(0.6, 16, 16, 2.0, 0.25)
    try:
        ctx.set_option("join_order_cover", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1
        first = ctx.stat("knn_join_order_pivots")
        ctx.set_option("join_order_cover", 150)
        idx, dist = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1
        rounds, grown = ctx.stat("knn_join_order_cover_rounds"), ctx.stat("knn_join_order_pivots")
        low, left = ctx.stat("knn_join_order_cover_low_e3"), ctx.stat("knn_join_order_cover_left_e3")
        # pivot groups of fewer than 400 rows: the cone bound on tile pairs takes two cones per tile, split where a group ends
        # ("join_cone_split_above_e3"; cone_tiles.hip) -- the same table with and without, and fewer tile pairs kept with
        split, kept_split = ctx.stat("knn_join_cone_split"), ctx.stat("knn_join_cone_tiles_kept_e6")
        ctx.set_option("join_cone_split_above_e3", -1)
        idx1, dist1 = A.knn_cosine(X, gp)
        kept_whole = ctx.stat("knn_join_cone_tiles_kept_e6")
        assert ctx.stat("knn_join_cone_split") == 0
    finally:
        ctx.set_option("join_order_cover", 150)
        ctx.set_option("join_cone_split_above_e3", 80)
    if d <= 128 and ctx.stat("knn_join_cone_tiles") == 1 and n < 400 * grown:
        assert split == 1 and kept_split <= kept_whole, (split, kept_split, kept_whole)
    assert np.array_equal(idx, idx1) and np.array_equal(dist, dist1)
    assert rounds >= 1 and grown > first and grown % 32 == 0 and grown <= n // 240, (first, grown, rounds)
    assert low >= 150 and left < low, (low, left)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(n, size=96, replace=False))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


@pytest.mark.parametrize("n,d,topk,rectify", [(70000, 128, 16, 0), (66001, 128, 40, 1), (70144, 256, 16, 0), (40000, 128, 5, 0)])
def test_item_graph_rerank_scores_every_pair_once(ctx, n, d, topk, rectify):
    """The exact re-rank of the joined lists (round 5, rerank_pairs.h): a pair two rows both keep is fetched and chained ONCE and
    its key handed to both (x_i . x_j in the reference's order, src/core.rs:204-210, and the product of the two norms are the
    same bits from either side); batches are packed across rows; the ranking runs in a kernel of its own.  The tables must be
    what every row scoring all of its own survivors gives ("rerank_share" 0), what a slot stream that ran out of room and was
    laid out again gives ("rerank_stream_cap" 8), and the oracle's rows (src/laplacian.rs:243-252).  Duplicates (tie-break by index), a zero
    row, deep lists (topk 40: records of more than 32 slots take the full-wave ranking), rows of two 128-wide pieces, the
    rectified variant."""
    X, _ = synth.mixture(n, d, seed=41, n_clusters=40, rank=5)
    X[40:48] = X[3]
    X[11] = 0.0
    gp = A.GraphParams(0.6, 16, topk, 2.0, 0.25, rectify=bool(rectify))
    try:
        ctx.set_option("join_min_rows", 512)
        ctx.set_option("rerank_share", 1)
        idx1, dist1 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_rerank_batches") > 0
        scored, pairs, open1 = ctx.stat("knn_rerank_scored"), ctx.stat("knn_rerank_pairs_scored"), ctx.stat("knn_uncertified_rows")
        assert 0 < pairs < scored and ctx.stat("knn_rerank_stream_retries") == 0   # some pairs are mutual: fewer rows fetched than survivors
        ctx.set_option("rerank_share", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        # (a row is among its own candidates and is never scored: one survivor per row is no pair)
        assert ctx.stat("knn_rerank_pairs_scored") >= ctx.stat("knn_rerank_scored") - n and ctx.stat("knn_uncertified_rows") == open1
        ctx.set_option("rerank_share", 1)
        ctx.set_option("rerank_stream_cap", 8)
        idx2, dist2 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_rerank_stream_retries") == 1
    finally:
        ctx.set_option("rerank_stream_cap", 0)
        ctx.set_option("rerank_share", 1)
        ctx.set_option("join_min_rows", 32768)
    assert np.array_equal(idx1, idx0) and np.array_equal(dist1, dist0)
    assert np.array_equal(idx2, idx0) and np.array_equal(dist2, dist0)
    rng = np.random.default_rng(6)
    sample = np.sort(np.concatenate([rng.choice(n, size=120, replace=False), np.arange(40, 48)]))
    oi, od = O.knn_cosine_rows(X, O.graph_params(gp.eps, gp.k, gp.topk, gp.p, gp.sigma, rectify=gp.rectify), sample)
    assert np.array_equal(idx1[sample], oi) and np.array_equal(dist1[sample], od)


@pytest.mark.parametrize("n,d,topk", [(70000, 128, 16), (66001, 64, 30), (131072 + 300, 128, 5)])
def test_item_graph_thresholds_from_the_own_neighbourhood(ctx, n, d, topk):
    """"join_theta_band" (round 4): a row's threshold is raised to what its own neighbourhood in the ordered image certifies -- the
    (K+1)-th largest of the bin maxima of its products with ALL rows of its super tile and the ones beside it, lowered by twice the
    f16 bound: K other rows are then in its list with exact cosines above the cut + delta.  Fewer candidates, no more open rows than
    with the sampled thresholds alone, the same tables, the oracle's rows.  topk = 30: K + 1 = 32, the deepest the pass takes."""
    X, _ = synth.mixture(n, d, seed=37, n_clusters=40, rank=5)
    X[40:48] = X[3]
    X[11] = 0.0
    gp = A.GraphParams(0.6, 16, topk, 2.0, 0.25)
    try:
        # (the sampled thresholds from the plain sample in both runs: whether the grouped ones are kept is decided on the thresholds
        # the join will use, band included, and the two runs must differ in the band alone)
        ctx.set_option("join_theta_grouped", 0)
        ctx.set_option("join_theta_band", 0)
        idx0, dist0 = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_row_order") == 1
        open0, cand0 = ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_candidates")
        ctx.set_option("join_theta_band", 2)
        idx, dist = A.knn_cosine(X, gp)
        open1, cand1 = ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_candidates")
        ctx.set_option("join_theta_grouped", 1)
        idx2, dist2 = A.knn_cosine(X, gp)
        assert np.array_equal(idx2, idx0) and np.array_equal(dist2, dist0)
    finally:
        ctx.set_option("join_theta_band", 2)
        ctx.set_option("join_theta_grouped", 1)
    assert cand1 < cand0 and open1 <= open0 + max(2, open0 // 50), (cand0, cand1, open0, open1)
    assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    rng = np.random.default_rng(6)
    sample = np.sort(rng.choice(n, size=128, replace=False))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx[sample], oi) and np.array_equal(dist[sample], od)


def test_thresholds_of_few_rows_from_a_split_sample(ctx):
    """The threshold passes with few rows (the check of the grouped thresholds: 4 096 rows; the open rows of the second pass) split the
    sample over several units per 256 rows and take the order statistic of the bin-wise maxima ("join_theta_parts", mfma_theta_f16.hip):
    the very thresholds of one sweep -- the same candidate count, the same open rows at every stage, the same tables."""
    X, _ = synth.mixture(150_000, 128, seed=42)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    seen = []
    try:
        for parts in (1, 0):
            ctx.set_option("join_theta_parts", parts)
            idx, dist = A.knn_cosine(X, gp)
            seen.append((idx, dist, ctx.stat("knn_join_candidates"), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage1a"),
                         ctx.stat("knn_join_theta_grouped"), ctx.stat("knn_join_theta_grouped_bad")))
    finally:
        ctx.set_option("join_theta_parts", 1)
    assert seen[0][2:] == seen[1][2:], (seen[0][2:], seen[1][2:])
    assert seen[0][3] > 0  # (rows were left open: the second pass, and with it the fresh thresholds of a few rows, ran)
    assert np.array_equal(seen[0][0], seen[1][0]) and np.array_equal(seen[0][1], seen[1][1])


def _stress_case(seed, index):
    """Case `index` of scripts/stress_first_pass.py for `seed` (the same draws in the same order)."""
    rng = np.random.default_rng(seed)
    for c in range(index + 1):
        big = rng.random() < 0.25
        n = int(rng.integers(66000, 220000)) if big else int(rng.integers(600, 9000))
        d = int(rng.choice([8, 24, 31, 32, 33, 48, 64, 65, 96, 100, 127, 128, 129, 200, 256, 384, 500, 768, 1000, 1024]))
        if big and d > 256:
            d = int(rng.choice([64, 128, 192]))
        topk = int(rng.integers(3, 40))
        eps = float(rng.choice([0.3, 0.5, 0.9]))
        kind = rng.integers(0, 4)
        if kind == 0:
            X = rng.standard_normal((n, d))
        else:
            X, _ = synth.mixture(n, d, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.integers(2, 80)), rank=int(rng.integers(1, 9)))
        X[rng.integers(0, n, 5)] = X[rng.integers(0, n)]
        X[rng.integers(0, n)] = 0.0
        if rng.random() < 0.3:
            X *= rng.uniform(1e-3, 1e3, size=(n, 1))
        mode = int(rng.choice([1, 2, 3]))
    return X, topk, eps, mode


def test_item_graph_second_pass_does_not_list_the_row_itself(ctx, join_on):
    """The case of the randomised stress (seed 55, case 9: 211,570 x 128, K = 39) that came out with a wrong 39th neighbour in
    one run of five: the f16 second pass listed a row itself, so the pruning cut of the re-rank -- the K-th best LISTED score --
    sat one rank too high and the true K-th neighbour, 4e-6 from the next one, could be pruned.  (The row order, and with it
    which rows reach the second pass, varies from run to run: the subspace iteration accumulates with float atomics; hence the
    repeats.)  Tables with the first pass and the grouped thresholds must equal those of the plain join, and a sample of
    rows, the one that failed among them, the oracle's."""
    X, topk, eps, mode = _stress_case(55, 9)
    assert X.shape == (211570, 128) and topk == 38
    gp = A.GraphParams(eps, 16, topk, 2.0, 0.25)
    ctx.set_option("join_prefilter", 0)
    try:
        idx0, dist0 = A.knn_cosine(X, gp)
    finally:
        ctx.set_option("join_prefilter", 1)
    ctx.set_option("join_prefilter_min_rows", 0)
    try:
        for _ in range(4):
            idx, dist = A.knn_cosine(X, gp)
            assert ctx.stat("knn_join_prefilter") == 1
            assert np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
    finally:
        ctx.set_option("join_prefilter_min_rows", 65536)
    sample = np.concatenate([[119342], np.sort(np.random.default_rng(3).choice(len(X), size=23, replace=False))])
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    assert np.array_equal(idx0[sample], oi) and np.array_equal(dist0[sample], od)


@pytest.mark.parametrize("topk", [40, 63])
@pytest.mark.parametrize("pipe", [16, 48])
def test_item_graph_deep_tables_take_the_join(ctx, join_on, topk, pipe):
    """topk + 1 beyond what the top-k kernels keep in LDS (41 .. 64): the 16-bit join still produces the candidates (its
    lists do not depend on K) and what it leaves open goes straight to the exact scans; tables are the oracle's."""
    X, _ = synth.mixture(3000, 96, seed=21, n_clusters=6, rank=4)
    X[7] = 0.0
    gp = A.GraphParams(0.9, 16, topk, 2.0, 0.25)
    ctx.set_option("join_pipe", pipe)
    try:
        idx, dist = A.knn_cosine(X, gp)
        assert ctx.stat("knn_join") == 1 and ctx.stat("knn_join_pipe") == pipe
        assert ctx.stat("knn_uncertified_rows_stage1b") == 0 and ctx.stat("knn_uncertified_rows_stage2") == 0
    finally:
        ctx.set_option("join_pipe", 16)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)


@pytest.mark.parametrize("sample,rank,cap", [(16, 5, 192), (2, 1, 24), (64, 8, 18), (4, 32, 1024)])
@pytest.mark.parametrize("bf16", [1, 0])
def test_item_graph_join_any_threshold_is_exact(ctx, join_on, sample, rank, cap, bf16):
    """The thresholds only decide how much work the later stages get: far too tight (lists overflow, rows fall
    through to the top-k / f64 / exact stages) or far too loose, the tables are the oracle's."""
    ctx.set_option("join_sample", sample)
    ctx.set_option("join_rank", rank)
    ctx.set_option("join_cap", cap)
    ctx.set_option("join_bf16", bf16)  # thresholds from split-bf16 products / from the f32 top-k pass
    X, _ = synth.mixture(3000, 64, seed=3, n_clusters=12, rank=4)
    X = np.concatenate([X, X[:300]])   # exact duplicates: mass ties
    X[17] = 0.0
    X[900] *= 1e-9
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    try:
        idx, dist = A.knn_cosine(X, gp)
    finally:
        ctx.set_option("join_bf16", 1)
    assert ctx.stat("knn_join") == 1
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_item_graph_nonfinite_rows_same_on_every_path(ctx, join_on):
    """Rows with NaN / inf entries.  The reference has no defined behaviour here (its in-tree model sorts with
    partial_cmp().unwrap(), which panics on a NaN distance: src/tests/test_helpers.rs:115-119) and the oracle does
    not pin one either; the library's rule is that a pair with a NaN cosine never enters a neighbour list (and a
    pair whose norm product is NaN gets cos := 0 like any product that is not > 1e-12).  Checked: the symmetric
    join, the top-k MFMA path and the exact scan produce identical tables, and finite rows never list one."""
    X, _ = synth.mixture(1500, 64, seed=4, n_clusters=8, rank=3)
    X[5, 3] = NAN
    X[6, 0] = INF
    X[700, :] = NAN
    X[701, 10] = -INF
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    out = []
    try:
        for join, mfma in ((1, 1), (0, 1), (0, 0)):
            ctx.set_option("knn_join", join)
            ctx.set_option("knn_mfma", mfma)
            idx, dist = A.knn_cosine(X, gp)
            out.append((np.array(idx), np.array(dist)))
    finally:
        ctx.set_option("knn_join", 1)
        ctx.set_option("knn_mfma", 1)
    for i, d in out[1:]:
        assert np.array_equal(i, out[0][0]) and np.array_equal(d, out[0][1], equal_nan=True)
    bad = [5, 6, 700, 701]
    finite_rows = np.setdiff1d(np.arange(len(X)), bad)
    assert not np.isin(out[0][0][finite_rows], bad).any()


def test_item_graph_join_eps_above_one(ctx, join_on):
    """eps > 1 admits negative cosines: thresholds and bounds below zero."""
    X, _ = synth.mixture(2000, 32, seed=8, n_clusters=6, rank=3)
    gp = A.GraphParams(1.2, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_duplicate_rows_tie_break(ctx, mode):
    """Duplicated rows (src/tests/test_builder.rs:376-384 style): exact ties must resolve to
    the lower index, identically to the oracle."""
    X, _ = synth.mixture(200, 16, seed=5, n_clusters=4, rank=2)
    X = np.concatenate([X, X[:90], X[:40]])  # 330 rows with exact duplicates / triplicates
    gp = A.GraphParams(0.5, 8, 8, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, ogp(gp)))


def test_zero_and_tiny_rows(ctx, mode):
    """cos := 0 when |a||b| <= 1e-12 (oracle rule set; src/tests/test_laplacian.rs:316 tiny values)."""
    X, _ = synth.mixture(100, 8, seed=9, n_clusters=4, rank=2)
    X[7] = 0.0
    X[20] *= 1e-9
    X[21] *= 1e-7
    gp = A.GraphParams(1.5, 4, 4, 2.0, None)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, ogp(gp)))


@pytest.mark.parametrize("tm,om,arg", [(A.TauMode.Median, O.TAU_MEDIAN, 0.0), (A.TauMode.Mean, O.TAU_MEAN, 0.0),
                                       (A.TauMode.Percentile(0.95), O.TAU_PERCENTILE, 0.95),
                                       (A.TauMode.Fixed(0.4), O.TAU_FIXED, 0.4)])
@pytest.mark.parametrize("n,f", [(5000, 128), (1000, 24), (600, 768)])
def test_ref_mapping_chain_matches_oracle(ctx, mode, n, f, tm, om, arg):
    """build() chain with supplied centroids: F x F graph bit-exact -> lambdas -> search."""
    X, cent = synth.mixture(n, f, seed=21, n_clusters=32, rank=6)
    b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).with_synthesis(tm)
    asp, gl = b.build(X, cent)
    ocsr = O.build_laplacian(np.ascontiguousarray(cent.T), O.graph_params(0.5, 16, 16, 2.0, 0.25))
    assert_csr_equal(gl.matrix.to_host(), ocsr)
    assert gl.nnodes == n and gl.matrix.shape() == (f, f)
    olam = O.taumode_lambdas(X, gl.matrix.to_host(), om, arg)  # same CSR values on both sides
    np.testing.assert_allclose(asp.lambdas(), olam, rtol=LAMBDA_RTOL, atol=1e-300)
    assert np.all(np.isfinite(olam))
    qi = np.array([3, 77, n - 1])
    Q = X[qi] * 1.02
    lq = asp.prepare_query_items(Q, gl)
    olq = np.array([O.prepare_query_item(q, gl.matrix.to_host(), om, arg) for q in Q])
    np.testing.assert_allclose(lq, olq, rtol=LAMBDA_RTOL)
    asp.update_lambdas(olam)
    idx, sc = asp.search_lambda_aware_batch(Q, olq, 10, 0.7)
    for j in range(len(Q)):
        oi, osc = O.search_lambda_aware(X, olam, Q[j], olq[j], 10, 0.7)
        assert np.array_equal(idx[j], oi)
        assert np.array_equal(sc[j], osc)


@pytest.mark.parametrize("n,f", [(3000, 128), (800, 24)])
def test_spectral_laplacian_chain_matches_oracle(ctx, n, f):
    """with_spectral(true): signals = build_laplacian_matrix(dense(L)^T) (src/graph.rs:212-270); the item lambdas
    come from `signals` (src/taumode.rs:195-200) while queries keep gl.matrix (src/core.rs:548)."""
    X, cent = synth.mixture(n, f, seed=21, n_clusters=32, rank=6)
    b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).with_spectral(True)
    asp, gl = b.build(X, cent)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    ocsr = O.build_laplacian(np.ascontiguousarray(cent.T), gp)
    dense = np.zeros((f, f))
    for i in range(f):
        s, e = int(ocsr[0][i]), int(ocsr[0][i + 1])
        dense[i, ocsr[1][s:e].astype(np.int64)] = ocsr[2][s:e]
    osig = O.build_laplacian(np.ascontiguousarray(dense.T), gp)
    assert asp.signals is not None and asp.signals.shape() == (f, f)
    assert_csr_equal(asp.signals.to_host(), osig)
    olam = O.taumode_lambdas(X, asp.signals.to_host(), O.TAU_MEDIAN)
    np.testing.assert_allclose(asp.lambdas(), olam, rtol=LAMBDA_RTOL, atol=1e-300)
    lq = asp.prepare_query_item(X[5] * 1.02, gl)
    assert lq == pytest.approx(O.prepare_query_item(X[5] * 1.02, gl.matrix.to_host(), O.TAU_MEDIAN), rel=LAMBDA_RTOL)


def test_search_large_k_and_ties(ctx, mode):
    X, _ = synth.mixture(500, 32, seed=2, n_clusters=5, rank=3)
    X = np.concatenate([X, X[:100]])  # exact score ties -> lower index first
    lam = np.linspace(0.05, 0.9, len(X))
    lam[500:] = lam[:100]
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    Q = X[[1, 50, 420]] * 1.02
    ql = np.array([0.3, 0.5, 0.7])
    for k in (1, 10, 64, 100, 600):
        idx, sc = asp.search_lambda_aware_batch(Q, ql, k, 0.7)
        for j in range(3):
            oi, osc = O.search_lambda_aware(X, lam, Q[j], ql[j], k, 0.7)
            assert np.array_equal(idx[j], oi), k
            assert np.array_equal(sc[j], osc), k


def test_config1_plumbing(ctx):
    """BASELINE config 1: 1000 x 64 i.i.d. Gaussian, default builder: empty feature graph,
    lambda == 0 everywhere, search raises the zero-lambda error."""
    X = synth.gaussian_rows(1000, 64, seed=1)
    asp, gl = A.ArrowSpaceBuilder.new(ctx).build(X, X)
    assert gl.matrix.nnz() == 64 and np.all(gl.matrix.data() == 0.0)
    assert np.all(asp.lambdas() == 0.0)
    lq = asp.prepare_query_item(X[0] * 1.02, gl)
    assert lq == 0.0
    with pytest.raises(A.ArrowSpaceError, match="Lambda of the item is 0.0"):
        asp.search_lambda_aware(A.ArrowItem(X[0] * 1.02, lq), 3, 0.7)


def test_device_resident_inputs(ctx):
    """Device pointers are used in place (how bench.py keeps inputs in HBM)."""
    import torch
    X, cent = synth.mixture(4000, 128, seed=4, n_clusters=32, rank=6)
    Xd = torch.from_numpy(X).cuda()
    b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25)
    asp_h, gl_h = b.build(X, cent)
    asp_d, gl_d = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(Xd, cent)
    assert torch.is_tensor(asp_d.lambdas()) and asp_d.lambdas().is_cuda
    assert np.array_equal(asp_d.lambdas().cpu().numpy(), asp_h.lambdas())
    Q = Xd[:5] * 1.02
    lq = asp_d.prepare_query_items(Q, gl_d)
    idx, sc = asp_d.search_lambda_aware_batch(Q, lq, 5, 0.7)
    idx_h, sc_h = asp_h.search_lambda_aware_batch(Q.cpu().numpy(), lq.cpu().numpy(), 5, 0.7)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), idx_h) and np.array_equal(sc.cpu().numpy(), sc_h)


def test_repeated_small_calls_are_stable(ctx, proteins):
    """Regression: workspace blocks and host staging are recycled between calls; results of many
    back-to-back small calls (index re-creation included) must stay identical to the oracle."""
    lam = np.linspace(0.05, 0.9, len(proteins))
    want = {}
    for j in (3, 10, 40):
        want[j] = O.search_lambda_aware(proteins, lam, proteins[j] * 1.02, 0.3 + 0.01 * j, 3, 0.7)
    for rep in range(60):
        asp = A.ArrowSpace(proteins, A.TauMode.Median, ctx)
        asp.update_lambdas(lam)
        for j, (oi, osc) in want.items():
            res = asp.search_lambda_aware(A.ArrowItem(proteins[j] * 1.02, 0.3 + 0.01 * j), 3, 0.7)
            assert [r[0] for r in res] == oi.tolist(), (rep, j)
            assert np.array_equal([r[1] for r in res], osc), (rep, j)


# ---- full-size checks: sampled rows against the oracle + size-independent properties ---------

@pytest.mark.parametrize("n,d", [(200_000, 128), (60_000, 384)])
def test_large_item_graph_sampled_parity_and_invariants(ctx, n, d):
    """At sizes the oracle cannot finish, (a) a random sample of rows must match the oracle's
    neighbour rows bit for bit, (b) the CSR must satisfy the reference's invariants
    (src/tests/test_laplacian.rs:51-152): symmetric, zero row sums, sorted unique columns,
    explicit diagonal, non-positive off-diagonals, nnz bound."""
    import torch
    X, _ = synth.mixture(n, d, seed=42)
    Xd = torch.from_numpy(X).cuda()
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    K = gp.topk + 1
    import ctypes as C
    idx = torch.empty((n, K), dtype=torch.int64, device="cuda")
    dst = torch.empty((n, K), dtype=torch.float64, device="cuda")
    A.api._adopt_torch_stream(ctx)
    c = gp.to_c()
    ctx.check(ctx.lib.as_knn_cosine(ctx.h, Xd.data_ptr(), n, d, 0, n, C.byref(c), idx.data_ptr(), dst.data_ptr()))
    torch.cuda.synchronize()
    assert ctx.stat("knn_cand_capacity") > 0  # the MFMA path ran
    rng = np.random.default_rng(1)
    sample = np.sort(rng.choice(n, size=96, replace=False))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    gi = idx[torch.from_numpy(sample).cuda()].cpu().numpy().astype(np.uint64)
    gd = dst[torch.from_numpy(sample).cuda()].cpu().numpy()
    assert np.array_equal(gi, oi)
    assert np.array_equal(gd, od)
    # neighbour rows are ordered by (distance, index) and exclude self
    dd = dst.cpu().numpy()
    ii = idx.cpu().numpy()
    assert np.all(np.diff(dd, axis=1) >= 0)
    assert not np.any(ii == np.arange(n)[:, None])
    # CSR invariants on the full graph
    csr = A.laplacian_from_knn(idx, dst, gp, ctx=ctx)
    indptr, indices, data = csr.to_host()
    assert indptr[-1] == len(indices) == len(data) and len(data) <= n * (2 * K + 1)
    rows_of = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    cols = indices.astype(np.int64)
    assert np.all((np.diff(cols) > 0) | (np.diff(rows_of) > 0))          # sorted, unique per row
    diag = cols == rows_of
    assert diag.sum() == n                                                 # explicit diagonal everywhere
    assert np.all(data[~diag] <= 0) and np.all(data[diag] >= 0)
    row_sum = np.bincount(rows_of, weights=data, minlength=n)
    assert np.max(np.abs(row_sum)) < 1e-12
    # symmetry: the multiset of (min, max, value) must pair up
    off = ~diag
    key = np.minimum(rows_of[off], cols[off]) * n + np.maximum(rows_of[off], cols[off])
    order = np.argsort(key, kind="stable")
    k_sorted, v_sorted = key[order], data[off][order]
    assert len(k_sorted) % 2 == 0 and np.array_equal(k_sorted[0::2], k_sorted[1::2])
    assert np.array_equal(v_sorted[0::2], v_sorted[1::2])


def test_large_search_sampled_parity(ctx):
    """1e5 items x 128, 2000 queries through the MFMA path; 40 sampled queries against the oracle."""
    import torch
    n, d, nq = 100_000, 128, 2000
    X, cent = synth.mixture(n, d, seed=42)
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(torch.from_numpy(X).cuda(), cent)
    lam = asp.lambdas().cpu().numpy()
    qi = synth.query_indices(n, nq, seed=3)
    Q = X[qi] * 1.02
    lq = asp.prepare_query_items(Q, gl)
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    assert ctx.stat("search_cand_capacity") > 0
    for j in range(0, nq, 50):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, 0.7)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j
    assert np.all(np.diff(sc, axis=1) <= 0)


@pytest.mark.parametrize("scale,want_f32", [(1.0, 1), (40.0, 0), (1e6, 0)])
def test_search_lambda_term_precision_choice(ctx, scale, want_f32):
    """The f32 candidate pass evaluates the lambda term in f32 while max|lambda_q| + max|lambda_x| <= 64 and in
    f64 beyond (large tau makes lambdas large); results are bit-identical to the oracle either way."""
    n, d, nq = 20_000, 128, 300
    X, _ = synth.mixture(n, d, seed=11, n_clusters=64, rank=6)
    rng = np.random.default_rng(8)
    lam = rng.random(n) * scale
    lam[::7] = lam[3] + np.arange(len(lam[::7])) * 1e-9 * scale  # near-ties in the lambda term
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=5)
    Q = X[qi] * 1.02
    lq = lam[qi] + rng.normal(size=nq) * 0.05 * scale
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    assert ctx.stat("search_lambda_term_f32") == want_f32
    for j in range(0, nq, 10):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, 0.7)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j


def test_laplacian_hub_rows_match_oracle(ctx):
    """k-NN graphs have hubs.  The CSR writer takes rows of up to 64 union entries on one wave, up to 4096 on one
    workgroup (bitonic sort in LDS) and longer ones serially: a graph with one row that is everybody's nearest
    neighbour, twenty sub-hubs (in-degree in the hundreds) and ordinary rows."""
    rng = np.random.default_rng(12)
    n, d = 5000, 48
    v0 = rng.normal(size=d)
    v0 /= np.linalg.norm(v0)
    U = rng.normal(size=(n, d))
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    X = 0.9 * v0 + 0.45 * U
    X[0] = v0
    centres = rng.normal(size=(20, d))
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    for c in range(20):
        X[1 + c] = 0.9 * v0 + 0.3 * centres[c]
        lo = 100 + 200 * c
        X[lo:lo + 200] = 0.9 * v0 + 0.3 * centres[c] + 0.05 * U[lo:lo + 200]
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    got = A.build_laplacian_matrix(X, gp).matrix.to_host()
    want = O.build_laplacian(X, ogp(gp))
    assert_csr_equal(got, want)
    indptr = np.asarray(want[0])
    assert int(np.max(np.diff(indptr))) > 512          # a row that needs the ten-stage LDS network
    assert int(np.sum(np.diff(indptr) > 65)) >= 20     # and a population of rows in the workgroup form


def test_self_in_results_variant_matches_oracle(ctx, mode):
    """The second CosinePair reading of SURVEY.md Appendix A as a flag (as_graph_params.self_in_results): tables and
    Laplacian equal the oracle's with the same flag; duplicates (places go by index) and a zero row included."""
    X, _ = synth.mixture(1400, 24, seed=9, n_clusters=8, rank=3)
    X[:7] = X[0]
    X[50] = 0.0
    gp = A.GraphParams(0.5, 6, 4, 2.0, 0.25, self_in_results=True)
    og = O.graph_params(0.5, 6, 4, 2.0, 0.25, self_in_results=True)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, og)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert (idx[:, -1] == np.uint64(A.api.U64_MAX)).sum() > 1000      # generic rows keep topk neighbours, not topk + 1
    assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, og))


def test_deep_neighbour_lists_take_the_exact_scan(ctx):
    """topk + 1 beyond the 64 entries the MFMA stages keep per row: the exact scan continues in passes of 64
    (AS_ERR_UNSUPPORTED before), the assembly takes rows of up to 256 neighbours."""
    X, _ = synth.mixture(900, 16, seed=12, n_clusters=4, rank=3)
    for topk in (70, 130):
        gp = A.GraphParams(0.8, 20, topk, 2.0, 0.5)
        idx, dist = A.knn_cosine(X, gp)
        oi, od = O.knn_cosine(X, ogp(gp))
        assert idx.shape == (900, topk + 1)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
        assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, ogp(gp)))
    with pytest.raises(A.ArrowSpaceError, match="256"):
        A.knn_cosine(X, A.GraphParams(0.8, 20, 300, 2.0, 0.5))


def test_hybrid_search_beyond_64(ctx):
    X, _ = synth.mixture(1500, 48, seed=8, n_clusters=10, rank=4)
    X = np.concatenate([X, X[:20] * 1.000001])
    lam = np.linspace(0.02, 0.95, len(X))
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    for j, lq, k in ((3, 0.4, 65), (7, 0.11, 200)):
        q = X[j] * 1.02
        got = asp.search_lambda_aware_hybrid(A.ArrowItem(q, lq), k, 0.7)
        oi, osc = O.search_lambda_aware_hybrid(X, lam, q, lq, k, 0.7)
        assert [g[0] for g in got] == oi.tolist(), (j, k)
        assert np.array_equal([g[1] for g in got], osc), (j, k)


def test_handle_memory_cache_on_and_off(ctx):
    """Handle memory (CSR slabs, index images) goes through a cache of the blocks destroyed handles leave ("handle_cache_mb", api.hip):
    graphs and searches built on re-used blocks are the ones built on fresh memory."""
    import gc
    X, cent = synth.mixture(70_000, 128, seed=42)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    ref = None
    try:
        for mb in (4096, 4096, 0, 4096):
            ctx.set_option("handle_cache_mb", mb)
            gl = A.build_laplacian_matrix(X, gp, ctx=ctx)
            got = gl.matrix.to_host()
            asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
            asp.update_lambdas(np.linspace(0.05, 0.9, len(X)))
            hits = asp.search_lambda_aware(A.ArrowItem(X[11] * 1.01, 0.4), 10, 0.7)
            if ref is None:
                ref = (got, hits)
            assert all(np.array_equal(a, b) for a, b in zip(ref[0], got)) and hits == ref[1]
            del gl, asp
            gc.collect()
        # a handle takes the smallest idle block that holds it and is at most a quarter larger: graphs of other data (another
        # number of entries) re-use the blocks too, and what is built in a larger block is what is built in a fresh one
        X2, _ = synth.mixture(70_000, 128, seed=44, n_clusters=200)
        ctx.set_option("handle_cache_mb", 0)
        fresh2 = A.build_laplacian_matrix(X2, gp, ctx=ctx).matrix.to_host()
        ctx.set_option("handle_cache_mb", 4096)
        for Xa, want in ((X, ref[0]), (X2, fresh2), (X, ref[0]), (X2, fresh2)):
            gl = A.build_laplacian_matrix(Xa, gp, ctx=ctx)
            got = gl.matrix.to_host()
            assert all(np.array_equal(a, b) for a, b in zip(want, got))
            del gl
            gc.collect()
    finally:
        ctx.set_option("handle_cache_mb", 4096)


def test_graph_assembly_hub_rows_on_every_path(ctx):
    """A neighbour table with hubs of every size class of the assembly (graph_assembly.hip): two nodes that ~12 000 rows point to
    (the serial form past 4 096 entries), ten with ~1 200 in-edges (a workgroup per row), fifty with ~240 (a wave per row), the
    rest short rows side by side.  The lane-parallel kernels (K <= 32) and the thread-per-row form ("assembly_lanes" 0: its own sort and
    union code, what tables deeper than 32 take) must give the same CSR bit for bit, as whole matrix and as row blocks,
    and the CSR keeps the reference's invariants (src/tests/test_laplacian.rs:51-152)."""
    n, K = 12_000, 12
    rng = np.random.default_rng(5)
    idx = np.empty((n, K), dtype=np.uint64)
    rows = np.arange(n)
    idx[:, 0], idx[:, 1] = 0, 1
    idx[:, 2] = 2 + rows % 50
    idx[:, 3] = 100 + rows % 10
    for i in range(n):
        idx[i, 4:] = 200 + rng.choice(n - 200, size=K - 4, replace=False)
    # a distance per unordered pair, as a cosine distance is (the union keeps one weight per pair: src/laplacian.rs:303-320)
    pair = np.minimum(rows[:, None], idx.astype(np.int64)) * n + np.maximum(rows[:, None], idx.astype(np.int64))
    dist = 0.01 + 0.44 * ((pair * 2654435761) % (1 << 32)) / float(1 << 32)
    gp = A.GraphParams(0.5, 16, K - 1, 2.0, 0.25)
    got = {}
    for lanes in (2, 0):
        ctx.set_option("assembly_lanes", lanes)
        try:
            got[lanes] = A.laplacian_from_knn(idx, dist, gp, ctx=ctx).to_host()
            if lanes == 2:
                blocks = [A.laplacian_from_knn(idx, dist, gp, row_begin=a, row_count=b - a, ctx=ctx).to_host() for a, b in ((0, 5000), (5000, n))]
        finally:
            ctx.set_option("assembly_lanes", 2)
    for lanes in (0,):
        for a, b in zip(got[2], got[lanes]):
            assert np.array_equal(a, b), lanes
    indptr, indices, data = got[2]
    assert np.array_equal(np.concatenate([blocks[0][0], blocks[1][0][1:] + blocks[0][0][-1]]), indptr)
    assert np.array_equal(np.concatenate([blocks[0][1], blocks[1][1]]), indices) and np.array_equal(np.concatenate([blocks[0][2], blocks[1][2]]), data)
    counts = np.diff(indptr.astype(np.int64))
    assert counts[0] > 4096 and counts[1] > 4096 and 512 < counts[100] <= 4096 and 64 < counts[2] <= 512
    rows_of = np.repeat(np.arange(n), counts)
    cols = indices.astype(np.int64)
    assert np.all((np.diff(cols) > 0) | (np.diff(rows_of) > 0))          # sorted, unique per row
    diag = cols == rows_of
    assert diag.sum() == n and np.all(data[~diag] <= 0) and np.all(data[diag] >= 0)
    off = ~diag
    key = np.minimum(rows_of[off], cols[off]) * n + np.maximum(rows_of[off], cols[off])
    order = np.argsort(key, kind="stable")
    k_sorted, v_sorted = key[order], data[off][order]
    assert len(k_sorted) % 2 == 0 and np.array_equal(k_sorted[0::2], k_sorted[1::2]) and np.array_equal(v_sorted[0::2], v_sorted[1::2])
    # the degree of a row is the sequential sum of its weights in column order (src/laplacian.rs:369)
    for r in (0, 100, 2, 7000):
        seg = slice(int(indptr[r]), int(indptr[r + 1]))
        acc = 0.0
        for c, v in zip(cols[seg], data[seg]):
            if c != r:
                acc = acc + (-v)
        assert data[seg][cols[seg] == r][0] == acc


def test_laplacian_row_blocks_stack_to_the_whole_matrix(ctx):
    """as_laplacian_from_knn_rows (the sharded assembly: every rank its own row block from the gathered table): blocks
    of consecutive ranges, hub rows and an empty range included, stacked = the CSR of as_laplacian_from_knn."""
    X, _ = synth.mixture(6000, 32, seed=17, n_clusters=12, rank=4)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    whole = A.laplacian_from_knn(idx, dist, gp).to_host()
    assert_csr_equal(whole, O.build_laplacian(X, ogp(gp)))
    n = len(X)
    for cuts in ([0, 2048, 4096, 6000], [0, 512, 512, 5999, 6000], [0, 6000]):
        ip, ix, dv, base = [np.zeros(1, dtype=np.uint64)], [], [], 0
        for a, b in zip(cuts, cuts[1:]):
            blk = A.laplacian_from_knn(idx, dist, gp, row_begin=a, row_count=b - a)
            assert blk.shape() == (b - a, n) and blk.row_begin() == a
            bip, bix, bdv = blk.to_host()
            assert bip[0] == 0 and len(bip) == b - a + 1
            ip.append(bip[1:] + np.uint64(base))
            base += int(bip[-1])
            ix.append(bix)
            dv.append(bdv)
        got = (np.concatenate(ip), np.concatenate(ix), np.concatenate(dv))
        assert np.array_equal(got[0], whole[0]) and np.array_equal(got[1], whole[1]) and np.array_equal(got[2], whole[2])
    blk = A.laplacian_from_knn(idx, dist, gp, row_begin=0, row_count=100)
    with pytest.raises(A.ArrowSpaceError, match="row block"):   # a row block is not a lambda graph
        A.compute_synthetic_lambda(np.ones(n), blk, 0.5)


def test_very_wide_rows_take_the_exact_scan(ctx):
    """Rows wider than the exact re-rank can stage in LDS (a feature graph over thousands of centroids; build(rows)
    with every item its own centroid): the MFMA stages are skipped, results are the oracle's."""
    rng = np.random.default_rng(4)
    n, d = 600, 6000
    base = rng.normal(size=(8, d))
    X = base[rng.integers(0, 8, size=n)] + 0.4 * rng.normal(size=(n, d))
    gp = A.GraphParams(0.5, 6, 6, 2.0, 0.25)
    idx, dist = A.knn_cosine(X, gp)
    oi, od = O.knn_cosine(X, ogp(gp))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    lam = rng.random(n)
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    res = asp.search_lambda_aware(A.ArrowItem(X[3] * 1.02, 0.4), 5, 0.7)
    oi, osc = O.search_lambda_aware(X, lam, X[3] * 1.02, 0.4, 5, 0.7)
    assert [r[0] for r in res] == oi.tolist() and np.array_equal([r[1] for r in res], osc)


# ---- clustering pre-stage: batch nearest_centroid (src/clustering.rs:913-928) -----------------------

@pytest.mark.parametrize("n,d,nc", [(5000, 128, 256), (777, 33, 70), (64, 2, 1), (1, 5, 3), (3000, 200, 129), (4097, 16, 64)])
def test_nearest_centroids_match_oracle(ctx, n, d, nc):
    """Index and squared distance bit-identical to the oracle: ragged row / centroid / feature counts, duplicated
    centroids (the first of equal minima wins), a row equal to a centroid (distance 0), rows with NaN and inf."""
    rng = np.random.default_rng(n + d)
    X = rng.normal(size=(n, d))
    Cn = rng.normal(size=(nc, d))
    if nc >= 6:
        Cn[5] = Cn[2]           # duplicated centroid
        Cn[nc - 1] = Cn[0]
    if n >= 10:
        X[3] = Cn[min(2, nc - 1)]   # a row that IS a centroid
        X[7, 0] = NAN               # every distance NaN -> (0, +inf)
        X[8, d - 1] = INF
    idx, d2 = A.nearest_centroids(X, Cn)
    oi, od = O.nearest_centroids(X, Cn)
    assert np.array_equal(idx, oi)
    assert np.array_equal(d2, od, equal_nan=True)
    if n >= 10:
        assert idx[7] == 0 and d2[7] == INF


@pytest.mark.parametrize("case", ["blobs128", "narrow_many", "panels200", "near_ties", "offset", "odd_width"])
def test_nearest_centroids_mfma_candidates_equal_the_exact_sweep(ctx, case):
    """The candidate path of a clustering batch (csrc/centroids.hip: f32 matrix-pipe keys |p|^2 - 2 q.p, the entries within
    2 E of the best re-ranked in the reference's arithmetic, everything undecided swept exactly) against the oracle:
    index and squared distance bit for bit -- duplicated and nearly coincident centroids (first of equal minima; the window
    holds more than the lists: exact sweep), rows that ARE centroids, a large common offset (a window as wide as the data),
    widths across the packed shapes."""
    rng = np.random.default_rng({"blobs128": 1, "narrow_many": 2, "panels200": 3, "near_ties": 4, "offset": 5, "odd_width": 6}[case])
    if case == "blobs128":
        n, d, nc = 5000, 128, 317
        Cn = rng.normal(size=(nc, d)) * 2
        X = Cn[rng.integers(0, nc, size=n)] + 0.4 * rng.normal(size=(n, d))
    elif case == "narrow_many":
        n, d, nc = 4096, 8, 1500
        Cn = rng.normal(size=(nc, d))
        X = rng.normal(size=(n, d))
    elif case == "panels200":
        n, d, nc = 2000, 200, 100
        Cn = rng.normal(size=(nc, d))
        X = Cn[rng.integers(0, nc, size=n)] + 0.1 * rng.normal(size=(n, d))
    elif case == "odd_width":
        n, d, nc = 1500, 37, 70
        Cn = rng.normal(size=(nc, d))
        X = rng.normal(size=(n, d))
    elif case == "near_ties":
        n, d, nc = 3000, 64, 200
        base = rng.normal(size=(20, d))
        Cn = np.repeat(base, 10, axis=0) + 1e-9 * rng.normal(size=(nc, d))   # ten centroids within 1e-9 of each other
        Cn[7] = Cn[3]; Cn[150] = Cn[151]                                       # exact duplicates
        X = base[rng.integers(0, 20, size=n)] + 0.3 * rng.normal(size=(n, d))
    else:
        n, d, nc = 2048, 32, 128
        Cn = rng.normal(size=(nc, d)) + 1e4
        X = rng.normal(size=(n, d)) + 1e4
    X[5] = Cn[2]; X[6] = Cn[nc - 1]                                             # rows that are centroids: distance 0
    oi, od = O.nearest_centroids(X, Cn)
    ctx.set_option("nearest_mfma", 2)
    ctx.set_option("debug", 0x200000)
    try:
        c0, l0 = ctx.stat("nearest_mfma_calls"), ctx.stat("nearest_mfma_leftover_rows")
        idx, d2 = A.nearest_centroids(X, Cn, ctx=ctx)
        calls, left = ctx.stat("nearest_mfma_calls") - max(c0, 0), ctx.stat("nearest_mfma_leftover_rows") - max(l0, 0)
    finally:
        ctx.set_option("nearest_mfma", 1)
        ctx.set_option("debug", 0)
    assert calls == 1
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    if case in ("near_ties", "offset"):
        assert left > n // 2            # the window holds more than a list keeps: these rows took the exact sweep
    if case == "blobs128":
        assert left < n // 100          # separated centroids: one or two candidates per row


def test_nearest_centroids_device_inputs(ctx):
    import torch
    rng = np.random.default_rng(3)
    X = rng.normal(size=(2000, 128))
    Cn = rng.normal(size=(100, 128))
    idx, d2 = A.nearest_centroids(torch.from_numpy(X).cuda(), torch.from_numpy(Cn).cuda())
    oi, od = O.nearest_centroids(X, Cn)
    assert np.array_equal(np.asarray(idx), oi) and np.array_equal(np.asarray(d2), od)


def test_join_list_pack_unpack_round_trip(ctx):
    """as_knn_join_pack / as_knn_join_unpack (the compact form in which candidate lists cross the links): used slots
    side by side in row order, offsets in slots; rows over capacity contribute their first `slots` entries."""
    import torch
    from arrowspace_amd import distributed as D
    ops = D.HipOps(ctx)
    rng = np.random.default_rng(6)
    n, slots = 5000, 12
    cnt = rng.integers(0, slots + 4, size=n).astype(np.int32)
    cnt[::97] = 0
    cand = rng.integers(0, 1 << 30, size=(n, 2 * slots)).astype(np.int32)
    offsets, packed = ops.join_pack(torch.from_numpy(cnt).cuda(), torch.from_numpy(cand).cuda())
    used = np.minimum(cnt, slots)
    want_off = np.concatenate([[0], np.cumsum(used)])
    assert np.array_equal(offsets.cpu().numpy(), want_off)
    want = np.concatenate([cand[r, : 2 * used[r]] for r in range(n)])
    assert np.array_equal(packed.cpu().numpy()[: len(want)], want)
    back = ops.join_unpack(torch.from_numpy(cnt).cuda().reshape(1, n), packed[: len(want)], 2 * slots).cpu().numpy()[0]
    for r in range(0, n, 37):
        assert np.array_equal(back[r, : 2 * used[r]], cand[r, : 2 * used[r]])


# ---- threshold scan of the search (stage 0 of K9, csrc/mfma_join_bf16.hip MODE 2) ---------------

@pytest.fixture
def scan_on(ctx):
    """The threshold scan at test sizes (it starts at 32,768 items by default)."""
    ctx.set_option("join_min_rows", 512)
    yield
    ctx.set_option("join_min_rows", 32768)
    for name, v in (("search_scan", 1), ("search_scan_sample", 32), ("search_scan_rank", 3), ("search_scan_cap", 512)):
        ctx.set_option(name, v)


@pytest.mark.parametrize("n,d,nq,k,alpha", [(20_000, 128, 300, 10, 0.7), (5_000, 64, 77, 10, 0.7), (9_000, 100, 1, 5, 0.3),
                                             (3_000, 33, 40, 25, 1.0), (12_000, 128, 513, 1, 0.9), (2_000, 16, 10, 64, 0.5)])
def test_search_scan_matches_oracle(ctx, scan_on, n, d, nq, k, alpha):
    """Stage 0 on: results bit-identical to the oracle (and to the search without it), for one query and many,
    k below and above the usual, alpha = 1 (no lambda term) and small alpha (lambda term dominates)."""
    X, _ = synth.mixture(n, d, seed=11, n_clusters=48, rank=6)
    rng = np.random.default_rng(8)
    lam = rng.random(n)
    lam[::7] = lam[3] + np.arange(len(lam[::7])) * 1e-9  # near-ties in the lambda term
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=5)
    Q = X[qi] * 1.02
    lq = lam[qi] + rng.normal(size=nq) * 0.05
    ctx.set_option("search_scan", 1)
    idx, sc = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan") == 1
    assert ctx.stat("search_scan_candidates") >= nq * min(k, 1)
    ctx.set_option("search_scan", 0)
    idx0, sc0 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan") == 0
    assert np.array_equal(idx, idx0) and np.array_equal(sc, sc0)
    for j in range(0, nq, max(1, nq // 25)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], k, alpha)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j


@pytest.mark.parametrize("n,d,nq,k,alpha,spread", [(40_000, 128, 3000, 10, 0.7, 0.4), (36_000, 96, 1500, 7, 0.3, 5.0), (33_000, 128, 1100, 10, 0.9, 1e-6)])
def test_search_scan_in_the_order_of_the_lambdas(ctx, n, d, nq, k, alpha, spread):
    """A batch of 1024 queries or more on an index whose items are in their own order is scanned in the order of its
    lambdas, a lane of the scan ruling its column out by the lambda range of the row group (csrc/mfma_join_f16.hip): the
    oracle's results, the same with the order switched off; lambdas spread widely, narrowly, with duplicates, and queries
    whose lambda lies outside the items' range (src/core.rs:229-239,760-798)."""
    X, _ = synth.mixture(n, d, seed=23, n_clusters=20, rank=5)
    rng = np.random.default_rng(8)
    lam = rng.random(n) * spread + 0.01
    lam[::7] = lam[3]
    qi = synth.query_indices(n, nq, seed=9)
    Q = X[qi] * 0.97 + rng.normal(size=(nq, d)) * 0.01
    lq = lam[qi] + rng.normal(size=nq) * 0.05 * spread
    lq[:20] = lam.max() + 3.0 * spread + 2.0   # far outside: the lambda term is at its floor for every item
    lq[20:40] = lq[41]
    lq = np.abs(lq) + 1e-9
    res = {}
    try:
        for order in (1, 0):
            ctx.set_option("search_lambda_order", order)
            asp = A.ArrowSpace(X, A.TauMode.Median, ctx)  # a fresh index: items in their own order
            asp.update_lambdas(lam)
            res[order] = asp.search_lambda_aware_batch(Q, lq, k, alpha)
            assert ctx.stat("search_scan") == 1 and ctx.stat("search_scan_lambda_order") == order
    finally:
        ctx.set_option("search_lambda_order", 1)
    assert np.array_equal(np.asarray(res[0][0]), np.asarray(res[1][0])) and np.array_equal(np.asarray(res[0][1]), np.asarray(res[1][1]))
    for j in list(range(0, 45)) + list(range(45, nq, max(1, nq // 40))):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
        assert np.array_equal(np.asarray(res[1][0])[j], oi.astype(np.int64))
        assert np.array_equal(np.asarray(res[1][1])[j], osc)


@pytest.mark.parametrize("n,nq,k,alpha", [(131_072, 4133, 10, 0.7), (70_001, 2500, 5, 1.0), (90_000, 37, 10, 0.4)])
def test_index_adopts_the_images_of_the_item_graph_build(ctx, join_on, n, nq, k, alpha):
    """An item-graph build of 128-wide rows can keep its row order, its f16 image, its pivots, its 31 directions and its rotated int8
    image for an index of the SAME rows ("keep_join_images", round 6; as_index_create adopts them when pointer, shape and row norms
    agree): the FIRST search then runs on the ordered image behind the int8 first pass -- the batch projected on the build's
    directions, ordered by its pivots -- and gives what an index that found everything itself gives, and the oracle's results; so do
    later searches and small batches (whose filter finds its own directions).  Rows changed in between, other rows at another address
    and a build without the option leave the index on its own."""
    import torch
    X, _ = synth.mixture(n, 128, seed=41, n_clusters=40, rank=6)
    rng = np.random.default_rng(9)
    lam = rng.random(n) * 0.4
    qi = synth.query_indices(n, nq, seed=3)
    Q = X[qi] * 1.03 + rng.normal(size=(nq, 128)) * 0.01
    lq = lam[qi] + rng.normal(size=nq) * 0.02
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    Xd = torch.from_numpy(X).cuda()
    lamd = torch.from_numpy(lam).cuda()

    def index_of(rows_d):
        asp = A.ArrowSpace(rows_d, A.TauMode.Median, ctx)
        asp.update_lambdas(lamd)
        asp._ensure_index()
        return asp

    ref = index_of(Xd)  # no build before it: on its own
    assert ctx.stat("index_adopted_join_images") == 0
    idx0, sc0 = ref.search_lambda_aware_batch(Q, lq, k, alpha)
    idx0, sc0 = np.asarray(idx0.cpu() if torch.is_tensor(idx0) else idx0), np.asarray(sc0.cpu() if torch.is_tensor(sc0) else sc0)
    for j in [0, nq - 1] + list(rng.choice(nq, size=5, replace=False)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
        assert np.array_equal(idx0[j].astype(np.int64), oi.astype(np.int64)) and np.array_equal(sc0[j], osc)

    def same(res):
        i, s_ = res
        i = np.asarray(i.cpu() if torch.is_tensor(i) else i)
        s_ = np.asarray(s_.cpu() if torch.is_tensor(s_) else s_)
        return np.array_equal(i, idx0) and np.array_equal(s_, sc0)

    ctx.set_option("keep_join_images", 1)
    ctx.set_option("keep_join_images_min_rows", 65536)  # (the default keeps them from 2^18 rows)
    try:
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        assert ctx.stat("knn_join_images_kept") == 1
        asp = index_of(Xd)
        assert ctx.stat("index_adopted_join_images") == 1
        r1 = asp.search_lambda_aware_batch(Q, lq, k, alpha)  # the FIRST search of the index
        if nq >= 2048:
            assert ctx.stat("search_scan_prefilter") == 1 and ctx.stat("search_scan_order") == 1
        assert same(r1)
        assert same(asp.search_lambda_aware_batch(Q, lq, k, alpha))
        i5, s5 = asp.search_lambda_aware_batch(Q[:5], lq[:5], k, alpha)  # a small batch: the streaming scan, its filter's own directions
        assert np.array_equal(np.asarray(i5.cpu() if torch.is_tensor(i5) else i5), idx0[:5])
        assert same(asp.search_lambda_aware_batch(Q, lq, k, alpha))
        # the same rows at another address: nothing is adopted (and what was kept is gone with the first index)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        other = index_of(Xd.clone())
        assert ctx.stat("index_adopted_join_images") == 0
        assert same(other.search_lambda_aware_batch(Q, lq, k, alpha))
        # a row rewritten between the build and the index: its norm differs, nothing is adopted
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        Xd[12345] *= 1.5
        changed = index_of(Xd)
        assert ctx.stat("index_adopted_join_images") == 0
        Xd[12345] /= 1.5
        ctx.set_option("keep_join_images_min_rows", 262144)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        assert ctx.stat("knn_join_images_kept") == 0  # fewer rows than the default asks for
    finally:
        ctx.set_option("keep_join_images", 0)
        ctx.set_option("keep_join_images_min_rows", 262144)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    assert ctx.stat("knn_join_images_kept") == 0
    index_of(Xd)
    assert ctx.stat("index_adopted_join_images") == 0


@pytest.mark.parametrize("n,nq,k,alpha,run", [(131_072, 4133, 10, 0.7, 1), (70_001, 2049, 3, 1.0, 4), (200_000, 10_000, 10, 0.3, 1)])
def test_search_scan_with_a_wave_per_slice_of_queries(ctx, n, nq, k, alpha, run):
    """The scan of a re-searched index of 128-wide items behind the int8 first pass, a wave on its own per (64 queries, one column
    super tile) with the fragments of a flagged block straight from the image ("search_scan_slices", round 6, mfma_join_f16.hip:
    scan_slices_f16_kernel), gives what the form that shares a stage of columns among the eight waves of a workgroup gives, and the
    oracle's results: batches that end inside a slice of 64 and inside a super tile, runs of one and of four units a draw."""
    X, _ = synth.mixture(n, 128, seed=23, n_clusters=40, rank=6)
    rng = np.random.default_rng(5)
    lam = rng.random(n) * 0.4
    qi = synth.query_indices(n, nq, seed=8)
    Q = X[qi] * 1.03 + rng.normal(size=(nq, 128)) * 0.01
    lq = lam[qi] + rng.normal(size=nq) * 0.02
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    idx1, sc1 = asp.search_lambda_aware_batch(Q, lq, k, alpha)  # (first search: the image in the order of the items, no first pass)
    assert ctx.stat("search_scan_prefilter") == 0
    ctx.set_option("search_scan_run", run)
    try:
        idx2, sc2 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
        assert ctx.stat("search_scan_order") == 1 and ctx.stat("search_scan_prefilter") == 1
        ctx.set_option("search_scan_slices", 0)
        idx3, sc3 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
        assert ctx.stat("search_scan_prefilter") == 1
    finally:
        ctx.set_option("search_scan_slices", 2)
        ctx.set_option("search_scan_run", 1)
    assert np.array_equal(idx1, idx2) and np.array_equal(sc1, sc2)
    assert np.array_equal(idx1, idx3) and np.array_equal(sc1, sc3)
    for j in [0, 63, 64, nq - 1] + list(rng.choice(nq, size=6, replace=False)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
        assert np.array_equal(idx2[j], oi.astype(np.int64)) and np.array_equal(sc2[j], osc)


@pytest.mark.parametrize("n,d,nq,k,alpha", [(70_000, 128, 2500, 10, 0.7), (66_000, 64, 2100, 5, 1.0), (131_072, 100, 3000, 10, 0.4)])
def test_search_scan_on_an_ordered_index(ctx, n, d, nq, k, alpha):
    """An index of 65,536 items or more that is searched again keeps its f16 image grouped by direction, orders large query
    batches the same way and runs an int8 first pass on rotated images in front of the scan (csrc/mfma_prefilter_i8.hip):
    first scan in the order of the items, second and third with everything on (the switch happens between them), a small
    batch through the same stages, and a fresh index with the ordering at once but without the first pass.  All of them
    give the oracle's results."""
    X, _ = synth.mixture(n, d, seed=19, n_clusters=30, rank=5)
    rng = np.random.default_rng(4)
    lam = rng.random(n) * 0.4
    lam[::5] = lam[2] + np.arange(len(lam[::5])) * 1e-9
    qi = synth.query_indices(n, nq, seed=6)
    Q = X[qi] * 1.03 + rng.normal(size=(nq, d)) * 0.01
    lq = lam[qi] + rng.normal(size=nq) * 0.02
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    idx1, sc1 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan") == 1 and ctx.stat("search_scan_order") == 0 and ctx.stat("search_scan_prefilter") == 0
    assert ctx.stat("search_scan_lambda_order") == 1  # items in their own order: the batch goes in the order of its lambdas
    idx2, sc2 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan_order") == 1 and ctx.stat("search_scan_prefilter") == 1
    assert 0 < ctx.stat("search_scan_prefilter_blocks_flagged") < ctx.stat("search_scan_prefilter_blocks") // 2
    idx3, sc3 = asp.search_lambda_aware_batch(Q[::-1].copy(), lq[::-1].copy(), k, alpha)
    assert np.array_equal(idx1, idx2) and np.array_equal(sc1, sc2)
    assert np.array_equal(idx1, idx3[::-1]) and np.array_equal(sc1, sc3[::-1])
    ctx.set_option("search_staged_min_batch", 1)
    try:
        idx4, sc4 = asp.search_lambda_aware_batch(Q[:37], lq[:37], k, alpha)  # small batch: items ordered, queries not
        assert ctx.stat("search_stream_scan") == 0 and ctx.stat("search_scan_prefilter") == 1 and ctx.stat("search_scan_order") == 0
    finally:
        ctx.set_option("search_staged_min_batch", 1 << 30)
    assert np.array_equal(idx1[:37], idx4) and np.array_equal(sc1[:37], sc4)
    if d == 128:  # small batches on an index searched before: the int8 filter in front of the streaming scan (stream_scan.hip)
        for b in (1, 7, 32):
            idx6, sc6 = asp.search_lambda_aware_batch(Q[100:100 + b], lq[100:100 + b], k, alpha)
            assert ctx.stat("search_stream_scan") == 1 and ctx.stat("stream_scan_filter") == 1
            assert 0 < ctx.stat("stream_scan_filter_items") < n * b // 4 and ctx.stat("stream_scan_filter_overflow") == 0
            assert np.array_equal(idx1[100:100 + b], idx6) and np.array_equal(sc1[100:100 + b], sc6)
        ctx.set_option("search_stream_filter", 0)
        try:
            idx7, sc7 = asp.search_lambda_aware_batch(Q[:9], lq[:9], k, alpha)
            assert ctx.stat("search_stream_scan") == 1 and ctx.stat("stream_scan_filter") == 0
        finally:
            ctx.set_option("search_stream_filter", 1)
        assert np.array_equal(idx1[:9], idx7) and np.array_equal(sc1[:9], sc7)
    ctx.set_option("search_scan_order", 2)
    ctx.set_option("search_prefilter", 0)
    try:
        asp2 = A.ArrowSpace(X, A.TauMode.Median, ctx)
        asp2.update_lambdas(lam)
        idx5, sc5 = asp2.search_lambda_aware_batch(Q, lq, k, alpha)
        assert ctx.stat("search_scan_order") == 1 and ctx.stat("search_scan_prefilter") == 0
    finally:
        ctx.set_option("search_scan_order", 1)
        ctx.set_option("search_prefilter", 1)
    assert np.array_equal(idx1, idx5) and np.array_equal(sc1, sc5)
    for j in range(0, nq, max(1, nq // 20)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], k, alpha)
        assert np.array_equal(idx1[j], oi), j
        assert np.array_equal(sc1[j], osc), j


@pytest.mark.parametrize("n,d,nq,k,alpha", [(70_000, 256, 2600, 10, 0.7), (66_000, 384, 2100, 5, 1.0), (70_000, 192, 300, 10, 0.4)])
def test_search_scan_of_wide_items_on_an_ordered_index(ctx, n, d, nq, k, alpha):
    """Items wider than 128 (round 5; src/core.rs:760-798 over configs[4]'s shape): an index that is searched again keeps an f16
    image in an order found on the items' rotated coordinates (rotate_wide.hip), projects a batch of queries on the same 31
    directions, orders a large batch the same way and runs the rectangular int8 first pass in front of the panelled scan, which
    then contracts only the pairs of 256-row super tiles the first pass leaves (listed, mfma_join_f16_wide.hip).  First scan in the
    order of the items, second and third with everything on, a batch too small to be ordered, the first pass switched off: the
    same indices and scores bit for bit, and the oracle's."""
    X, _ = synth.mixture(n, d, seed=23, n_clusters=30, rank=5)
    rng = np.random.default_rng(8)
    lam = rng.random(n) * 0.4
    lam[::5] = lam[2] + np.arange(len(lam[::5])) * 1e-9
    qi = synth.query_indices(n, nq, seed=6)
    Q = X[qi] * 1.03 + rng.normal(size=(nq, d)) * 0.01
    lq = lam[qi] + rng.normal(size=nq) * 0.02
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    idx1, sc1 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan") == 1 and ctx.stat("search_scan_order") == 0 and ctx.stat("search_scan_prefilter") == 0
    pairs_all = ((nq + 255) // 256) * (((n + 511) // 512) * 2)   # pairs of 256-row super tiles (queries x items) the scan would contract
    idx2, sc2 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
    assert ctx.stat("search_scan_prefilter") == 1 and ctx.stat("search_scan_order") == (1 if nq >= 2048 else 0)
    assert 0 < ctx.stat("search_scan_prefilter_blocks_flagged") < ctx.stat("search_scan_prefilter_blocks")
    if nq >= 2048:
        assert 0 < ctx.stat("search_scan_f16_pairs_listed") < pairs_all // 2   # most pairs of super tiles are never contracted
    idx3, sc3 = asp.search_lambda_aware_batch(Q[::-1].copy(), lq[::-1].copy(), k, alpha)
    assert np.array_equal(idx1, idx2) and np.array_equal(sc1, sc2)
    assert np.array_equal(idx1, idx3[::-1]) and np.array_equal(sc1, sc3[::-1])
    idx4, sc4 = asp.search_lambda_aware_batch(Q[:300], lq[:300], k, alpha)   # items ordered, queries as given
    assert ctx.stat("search_scan_prefilter") == 1 and ctx.stat("search_scan_order") == 0
    assert np.array_equal(idx1[:300], idx4) and np.array_equal(sc1[:300], sc4)
    ctx.set_option("search_prefilter", 0)
    try:
        idx5, sc5 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
        assert ctx.stat("search_scan_prefilter") == 0
    finally:
        ctx.set_option("search_prefilter", 1)
    assert np.array_equal(idx1, idx5) and np.array_equal(sc1, sc5)
    # queries the scan leaves open (forced: 64 candidate slots instead of 512) take the quick streaming pass over the items, 32 at a
    # time, where the items have 256 .. 1024 values (the one-wave-per-query exact scan took 0.9 s a query at 1 M x 768: two open
    # queries of 100 000 made a 41 ms batch take 1.8 s); with the pass switched off they take that exact scan: the same results
    if d >= 256 and nq >= 2048:
        ctx.set_option("search_scan_cap", 64)
        try:
            idx6, sc6 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
            left = ctx.stat("search_uncertified_rows")
            assert 0 < left <= 4096 and ctx.stat("search_leftover_stream_scan") == left, (left, ctx.stat("search_leftover_stream_scan"))
            ctx.set_option("search_stream_quick", 0)
            idx7, sc7 = asp.search_lambda_aware_batch(Q, lq, k, alpha)
            assert ctx.stat("search_leftover_stream_scan") == 0
        finally:
            ctx.set_option("search_scan_cap", 512)
            ctx.set_option("search_stream_quick", 1)
        assert np.array_equal(idx1, idx6) and np.array_equal(sc1, sc6)
        assert np.array_equal(idx1, idx7) and np.array_equal(sc1, sc7)
    for j in range(0, nq, max(1, nq // 20)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], k, alpha)
        assert np.array_equal(idx1[j], oi), j
        assert np.array_equal(sc1[j], osc), j


@pytest.mark.parametrize("rate,rank,cap", [(2, 1, 16), (1000, 1, 512), (32, 32, 64), (8, 2, 1024)])
def test_search_scan_any_guess_is_exact(ctx, scan_on, rate, rank, cap):
    """The score guess only decides how many items are listed and how many queries fall through: guesses far too
    high (rank 1 of half the items), far too low (rank 32 into 64 slots: the lists overflow) give the same results."""
    n, d, nq = 6_000, 128, 200
    X, _ = synth.mixture(n, d, seed=3, n_clusters=24, rank=5)
    rng = np.random.default_rng(2)
    lam = rng.random(n) * 0.8
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=9)
    Q = X[qi] * 0.97
    lq = rng.random(nq)
    ctx.set_option("search_scan", 0)
    idx0, sc0 = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    ctx.set_option("search_scan", 1)
    for name, v in (("search_scan_sample", rate), ("search_scan_rank", rank), ("search_scan_cap", cap)):
        ctx.set_option(name, v)
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    assert ctx.stat("search_scan") == 1
    assert np.array_equal(idx, idx0) and np.array_equal(sc, sc0)


def test_search_scan_not_used_outside_its_range(ctx, scan_on):
    """alpha outside (0, 1], large lambdas, rows wider than 128 with the f16 pipe switched off: the top-k passes run as before
    (wide rows on the f16 pipe take the scan too: the last case, and test_search_scan_of_wide_items)."""
    n = 3_000
    rng = np.random.default_rng(4)
    for d, alpha, scale, pipe, want in ((64, 1.5, 1.0, 16, 0), (64, 0.0, 1.0, 16, 0), (200, 0.7, 1.0, 32, 0), (64, 0.7, 1e3, 16, 0),
                                        (200, 0.7, 1.0, 16, 1)):
        ctx.set_option("join_pipe", pipe)
        X, _ = synth.mixture(n, d, seed=6, n_clusters=16, rank=4)
        lam = rng.random(n) * scale
        asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
        asp.update_lambdas(lam)
        Q = X[:20] * 1.02
        lq = lam[:20] * 1.01
        try:
            idx, sc = asp.search_lambda_aware_batch(Q, lq, 5, alpha)
        finally:
            ctx.set_option("join_pipe", 16)
        assert ctx.stat("search_scan") == want, (d, alpha, scale, pipe)
        for j in (0, 7, 19):
            oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 5, alpha)
            assert np.array_equal(idx[j], oi) and np.array_equal(sc[j], osc)


# ---- clustering pre-stage and JL projection (csrc/prestage.hip) against the oracle's restatement -----------------------

def _blobs(n, f, k, seed, spread=0.25):
    rng = np.random.default_rng(seed)
    C0 = rng.normal(size=(k, f)) * 3
    return C0[rng.integers(0, k, size=n)] + spread * rng.normal(size=(n, f))


@pytest.mark.parametrize("n,f", [(3000, 24), (700, 128), (9, 5), (20000, 32)])
def test_prestage_pieces_match_oracle(ctx, n, f):
    """Two-NN ratios, Lloyd, Calinski-Harabasz, compute_optimal_k: bit for bit the oracle's (the library and the oracle
    share the seeded definitions that stand in for the reference's RNG; src/clustering.rs:36-531)."""
    X = _blobs(n, f, 6, seed=n)
    if n > 100:
        X[10] = X[3]                               # a duplicate: d1 = 0, the row is skipped (NaN)
    ids = O.pre_sample_indices(n, min(n, 500), 129)
    got = A.two_nn_ratios(X, ids)
    want = O.two_nn_ratios(X, ids)
    assert np.array_equal(got, want, equal_nan=True)
    if n >= 50:
        k = 7
        assign, cents = A.kmeans_lloyd(X, k, 20, 42)
        oa, oc = O.kmeans_lloyd(X, k, 20, 42)
        assert np.array_equal(assign, oa) and np.array_equal(cents, oc)
        assert A.calinski_harabasz(X, assign, k) == O.calinski_harabasz(X, oa, k)
    assert A.compute_optimal_k(X, ctx=ctx) == O.compute_optimal_k(X)
    import torch
    assert A.compute_optimal_k(torch.from_numpy(X).cuda(), ctx=ctx) == O.compute_optimal_k(X)   # device-resident rows


@pytest.mark.parametrize("n,f", [(70000, 128), (66000, 40)])
def test_two_nn_on_the_matrix_pipe_matches_oracle(ctx, n, f):
    """From 65 536 rows on the two nearest distances of a sampled row come from f32 matrix-pipe keys with an exact re-measure of
    the points inside the error window (csrc/centroids.hip: two_nn_mfma): the ratios are the oracle's bit for bit --
    a sampled row with an exact duplicate elsewhere (d1 = 0: NaN ratio), near duplicates, a sampled row far from everything."""
    X = _blobs(n, f, 40, seed=n + f, spread=0.6)
    ids = O.pre_sample_indices(n, 500, 129)
    X[int(ids[3])] = X[(int(ids[3]) + 17) % n]                      # the sampled row is a duplicate of another row
    X[(int(ids[5]) + 1) % n] = X[int(ids[5])] * (1.0 + 1e-12)       # a near duplicate of a sampled row
    X[int(ids[7])] += 500.0                                         # a sampled outlier
    got = A.two_nn_ratios(X, ids, ctx=ctx)
    assert ctx.stat("two_nn_mfma") == 1 and ctx.stat("two_nn_mfma_overflow") == 0
    want = O.two_nn_ratios(X, ids)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isnan(got[3])
    ctx.set_option("nearest_mfma", 0)
    try:
        exact = A.two_nn_ratios(X, ids, ctx=ctx)
        assert ctx.stat("two_nn_mfma") == 0
    finally:
        ctx.set_option("nearest_mfma", 1)
    assert np.array_equal(exact, want, equal_nan=True)


@pytest.mark.parametrize("batch", [1, 7, 256])
def test_incremental_clustering_matches_oracle(ctx, batch):
    """src/clustering.rs:547-910 without a sampler: batch = 1 is the reference's sequential mode; larger batches take
    the snapshot once per batch.  Centroids, assignments (dropped rows included) and sizes equal the oracle's."""
    X = _blobs(5000, 16, 9, seed=3, spread=0.35)
    X[100:110] += 40.0                              # far outliers: new clusters until the cap, then dropped
    k, radius, _ = O.compute_optimal_k(X)
    cents, assign, sizes = A.incremental_clustering(X, k, radius, batch, ctx=ctx)
    # no row visits the host: batches in which centroids can still appear run as one chain over their rows
    # (cluster_seq_kernel; batch = 1 is a single chain with no snapshot), once the cap is reached the decisions come from the
    # snapshot's nearest centroid and the running means are applied per centroid in row order: the same centroids, bit for
    # bit, as with every batch decided on the host
    assert ctx.stat("clustering_sequential_batches") > 0 and ctx.stat("clustering_host_batches") == 0
    assert batch == 1 or ctx.stat("clustering_device_batches") > 0
    oc, oa, osz = O.incremental_clustering(X, k, radius, batch)
    assert np.array_equal(assign, oa) and np.array_equal(sizes, osz) and np.array_equal(cents, oc)
    assert (assign < 0).sum() == (oa < 0).sum()
    ctx.set_option("cluster_on_device", 0)
    try:
        ch, ah, sh = A.incremental_clustering(X, k, radius, batch, ctx=ctx)
        assert ctx.stat("clustering_device_batches") == 0 and ctx.stat("clustering_sequential_batches") == 0
    finally:
        ctx.set_option("cluster_on_device", 1)
    assert np.array_equal(ah, oa) and np.array_equal(sh, osz) and np.array_equal(ch, oc)
    import torch
    c2, a2, s2 = A.incremental_clustering(torch.from_numpy(X).cuda(), k, radius, batch, ctx=ctx)
    assert np.array_equal(a2, oa) and np.array_equal(c2, oc)


@pytest.mark.parametrize("n,f,k,radius_scale,batch", [
    (6000, 8, 1500, 0.02, 4096),     # more than 1024 centroids appear inside one batch: a thread owns several
    (3000, 37, 200, 0.3, 1),         # ragged width, sequential mode
    (2500, 130, 40, 0.5, 512),       # a width past two whole chunks of the chain
    (900, 3, 5, 1.0, 64),            # cap reached at once; rows beyond 1.5 radius are dropped
    (300, 16, 400, 0.1, 128),        # the cap is never reached
    (30000, 128, 150, 0.05, 4096),   # batches of 4 096 rows against >= 64 centroids: candidates from the matrix pipe
    (20000, 24, 90, 0.05, 2048),     # narrow rows, rows beyond 1.5 radius dropped
])
def test_incremental_clustering_chain_shapes(ctx, n, f, k, radius_scale, batch):
    """cluster_seq_kernel against the oracle across widths, cluster counts above the workgroup size, caps that are reached
    at once or never, NaN / duplicate rows (src/clustering.rs:596-905)."""
    X = _blobs(n, f, max(2, min(k, 60)), seed=n * 7 + f, spread=0.5)
    X[n // 2] = X[n // 3]                                      # a duplicate row: distance exactly 0
    if n < 20000:
        X[n // 5, f // 2] = np.nan                              # never nearer than anything: dropped unless it creates
    X[-7:] += 1e3                                               # far rows late in the order
    radius = radius_scale * float(np.nanmedian(np.sum((X[:100] - X[100:200]) ** 2, axis=1)))
    cents, assign, sizes = A.incremental_clustering(X, k, radius, batch, ctx=ctx)
    assert ctx.stat("clustering_host_batches") == 0 and ctx.stat("clustering_sequential_batches") > 0
    oc, oa, osz = O.incremental_clustering(X, k, radius, batch)
    assert len(cents) == len(oc)
    assert np.array_equal(assign, oa) and np.array_equal(sizes, osz) and np.array_equal(cents, oc, equal_nan=True)
    if n >= 20000:
        assert ctx.stat("nearest_mfma_calls") > 0 and len(cents) >= 64


def test_build_without_centroids_runs_the_prestage(ctx):
    """ArrowSpaceBuilder::build(rows) (src/builder.rs:249-455): clustering pre-stage -> F x F graph -> lambdas; the
    whole chain against the oracle's (sequential clustering: with_seed)."""
    X = _blobs(4000, 24, 8, seed=5, spread=0.4)
    b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.9, 8, 8, 2.0, 0.5).with_seed(128)
    asp, gl = b.build(X)
    k, radius, _ = O.compute_optimal_k(X, 128)
    oc, oa, osz = O.incremental_clustering(X, k, radius, 1)
    assert (b.cluster_max_clusters, b.cluster_radius) == (k, radius)
    assert asp.n_clusters == len(oc) and np.array_equal(asp.cluster_assignments, oa) and np.array_equal(asp.cluster_sizes, osz)
    topk = O.define_result_k(8, 8)                 # build() rewrites topk first (src/builder.rs:225-233)
    ocsr = O.build_laplacian(np.ascontiguousarray(oc.T), O.graph_params(0.9, 8, topk, 2.0, 0.5))
    assert_csr_equal(gl.matrix.to_host(), ocsr)
    np.testing.assert_allclose(asp.lambdas(), O.taumode_lambdas(X, ocsr, O.TAU_MEDIAN), rtol=LAMBDA_RTOL, atol=1e-300)
    # the default (batched) clustering is another schedule of the same algorithm: same cluster count cap, usable graph
    asp2, gl2 = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.9, 8, 8, 2.0, 0.5).build(X)
    assert 2 <= asp2.n_clusters <= k and gl2.matrix.shape() == (24, 24)


def _moons_hd(n, noise_xy, noise_hd, dims, seed):
    """The shape of the reference's make_moons_hd fixture (src/tests/test_data.rs:6256-6303): two half circles in the
    first two coordinates, uniform [0, noise) jitter everywhere.  numpy's generator, not the reference's Pcg64: the
    acceptance below is statistical."""
    rng = np.random.default_rng(seed)
    n0 = n // 2
    out = np.zeros((n, dims))
    t = rng.random(n) * np.pi
    out[:n0, 0], out[:n0, 1] = np.cos(t[:n0]), np.sin(t[:n0])
    out[n0:, 0], out[n0:, 1] = 1.0 - np.cos(t[n0:]), -np.sin(t[n0:]) - 0.5
    out[:, :2] += noise_xy * rng.random((n, 2))
    out[:, 2:] = noise_hd * rng.random((n, dims - 2))
    return out


@pytest.mark.parametrize("n,dims,data_seed", [(80, 11, 789), (2000, 24, 17)])
def test_prestage_statistical_acceptance_of_the_reference(ctx, n, dims, data_seed):
    """src/tests/test_taumode.rs:328-428 (test_builder_lambdas_consistency_properties): the reference does not pin the
    clustering's draws, it accepts a build whose lambdas are all finite and in [0, 1], vary, and whose mean stays within
    2x between two builds.  Here the two builds are two clustering seeds plus the default batched schedule, on the
    reference's fixture shape and parameters (with_lambda_graph(0.3, 5, 2, 2.0, None), Median); the oracle's pre-stage
    is held to the same acceptance, and the cluster counts of library and oracle agree seed for seed."""
    X = _moons_hd(n, 0.15, 0.4, dims, data_seed)

    def stats(l):
        return l.mean(), l.var(), l.min(), l.max()

    builds = []
    for seed in (128, 977, None):
        b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.3, 5, 2, 2.0, None).with_normalisation(False) \
            .with_spectral(False).with_synthesis(A.TauMode.Median)
        if seed is not None:
            b = b.with_seed(seed)
        asp, gl = b.build(X)
        lam = np.asarray(asp.lambdas())
        assert 0 < asp.n_clusters <= n and len(lam) == n          # one lambda per item, clusters <= items
        assert gl.matrix.shape() == (dims, dims)
        assert np.all(np.isfinite(lam)) and np.all((lam >= 0.0) & (lam <= 1.0))
        mean, var, lo, hi = stats(lam)
        assert hi > lo and var > 0.0
        if seed is not None:                                      # the oracle's pre-stage at the same seed: same clusters
            k, radius, _ = O.compute_optimal_k(X, seed)
            oc, oa, osz = O.incremental_clustering(X, k, radius, 1)
            assert asp.n_clusters == len(oc) and np.array_equal(asp.cluster_sizes, osz)
            ocsr = O.build_laplacian(np.ascontiguousarray(oc.T), O.graph_params(0.3, 5, O.define_result_k(5, 2), 2.0, None))
            olam = O.taumode_lambdas(X, ocsr, O.TAU_MEDIAN)
            assert np.all(np.isfinite(olam)) and np.all((olam >= 0.0) & (olam <= 1.0)) and olam.var() > 0.0
            np.testing.assert_allclose(lam, olam, rtol=LAMBDA_RTOL, atol=1e-300)
        builds.append((asp.n_clusters, mean))
    means = [m for _, m in builds]
    assert max(means) / min(means) < 2.0, builds                  # the reference's tolerance between builds
    counts = [c for c, _ in builds]
    assert max(counts) <= 2 * min(counts) + 2, builds             # cluster counts in one range across seeds / schedules


def test_jl_projection_matches_oracle(ctx):
    """ImplicitProjection (src/reduction.rs:166-203): the Gaussian matrix is the build's own seeded draw (moments
    checked), the projection arithmetic the reference's: equal to the oracle's on the same matrix, bit for bit, for
    host and device rows; build() with dims reduction projects the centroids and, at query time, the queries
    (src/eigenmaps.rs:247-280, src/core.rs:509-549)."""
    import torch
    d, r = 200, 48
    proj = A.ImplicitProjection(d, r, seed=9, ctx=ctx)
    G = proj.matrix()
    assert G.shape == (d, r) and abs(G.mean()) < 0.03 and abs(G.std() - 1.0) < 0.03
    assert np.array_equal(A.ImplicitProjection(d, r, seed=9, ctx=ctx).matrix(), G)      # seeded
    assert not np.array_equal(A.ImplicitProjection(d, r, seed=10, ctx=ctx).matrix(), G)
    X = np.random.default_rng(2).normal(size=(333, d))
    want = O.jl_project(X, G)
    assert np.array_equal(proj.project(X), want)
    assert np.array_equal(proj.project(torch.from_numpy(X).cuda()).cpu().numpy(), want)
    assert np.array_equal(proj.project(X[5]), want[5])
    assert A.compute_jl_dimension(100, 0.3) == O.jl_dimension(100, 0.3) and A.compute_jl_dimension(2, 0.5) == 32
    # through the builder: centroids projected before the graph, queries projected before their lambda
    Xb, cent = synth.mixture(3000, 128, seed=6, n_clusters=40, rank=6)
    b = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).with_dims_reduction(True, 0.9)
    asp, gl = b.build(Xb, cent)
    target = min(O.jl_dimension(40, 0.9), 64)
    assert asp.reduced_dim == target and gl.matrix.shape() == (target, target)
    Gb = asp.projection_matrix.matrix()
    ocsr = O.build_laplacian(np.ascontiguousarray(O.jl_project(cent, Gb).T), O.graph_params(0.5, 16, 16, 2.0, 0.25))
    assert_csr_equal(gl.matrix.to_host(), ocsr)
    # item lambdas: raw items against the reduced graph (the reference's own behaviour: src/taumode.rs:570-575,596)
    np.testing.assert_allclose(asp.lambdas(), O.taumode_lambdas(Xb, ocsr, O.TAU_MEDIAN), rtol=LAMBDA_RTOL, atol=1e-300)
    q = Xb[17] * 1.02
    lq = asp.prepare_query_item(q, gl)
    assert lq == pytest.approx(O.prepare_query_item(O.jl_project(q.reshape(1, -1), Gb)[0], ocsr, O.TAU_MEDIAN), rel=LAMBDA_RTOL)


# ---- one query / small batches: the exact streaming scan (csrc/stream_scan.hip) ------------------------------------

@pytest.mark.parametrize("nq,k", [(1, 10), (7, 1), (32, 10), (5, 64), (9, 17)])
def test_small_batch_search_streams_the_items_once(ctx, nq, k):
    """nq <= 32 queries against >= 65,536 items of width 128: sample pass -> guaranteed guess -> list pass -> final
    sort, all in the reference's arithmetic; every query compared with the oracle."""
    n, d = 70_001, 128
    X, _ = synth.mixture(n, d, seed=21, n_clusters=64, rank=6)
    rng = np.random.default_rng(nq + k)
    lam = rng.random(n)
    lam[::5] = lam[2]                                  # ties in the lambda term
    X[100:140] = X[7]                                  # exact duplicates of an item: equal scores, order by index
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=3)
    qi[0] = 7
    Q = X[qi] * 1.02
    lq = rng.random(nq) + 0.01
    idx, sc = asp.search_lambda_aware_batch(Q, lq, k, 0.7)
    assert ctx.stat("search_stream_scan") == 1 and ctx.stat("stream_scan_overflow_rows") == 0
    for j in range(nq):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], k, 0.7)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j
    ctx.set_option("stream_scan", 0)                   # and the staged path gives the same
    try:
        idx0, sc0 = asp.search_lambda_aware_batch(Q, lq, k, 0.7)
    finally:
        ctx.set_option("stream_scan", 1)
    assert ctx.stat("search_stream_scan") == 0
    assert np.array_equal(idx, idx0) and np.array_equal(sc, sc0)


@pytest.mark.parametrize("nq,d", [(1, 768), (3, 300), (1, 1024)])
def test_small_batch_search_of_wide_items(ctx, nq, d):
    """One query (or a few) against items wider than 128: the top-k pass cuts the columns into many pieces to fill the
    chip, and the re-rank that follows stages a wave's splits x KC candidates next to the row in LDS -- the cut is
    bounded by that (it used to fail with `invalid argument` at d = 768)."""
    n = 40_000
    X, _ = synth.mixture(n, d, seed=23, n_clusters=32, rank=6)
    rng = np.random.default_rng(nq + d)
    lam = rng.random(n)
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=5)
    Q = X[qi] * 1.02
    lq = rng.random(nq) + 0.01
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    for j in range(nq):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, 0.7)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j


@pytest.mark.parametrize("nq,d,alpha", [(300, 768, 0.7), (257, 300, 0.4), (1000, 200, 1.0)])
def test_search_scan_of_wide_items(ctx, nq, d, alpha):
    """Batches against items wider than 128: score guesses and the one-sided scan on the panelled f16 kernel (modes 3 / 4
    of pair_join_f16_wide_kernel), exact re-rank and certification as for 128-wide items; every query compared with the
    oracle, ties in the lambda term and duplicated items included."""
    n = 40_000
    X, _ = synth.mixture(n, d, seed=29, n_clusters=32, rank=6)
    rng = np.random.default_rng(nq + d)
    lam = rng.random(n)
    lam[::5] = lam[2]
    X[100:140] = X[7]
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    qi = synth.query_indices(n, nq, seed=5)
    qi[0] = 7
    Q = X[qi] * 1.02
    lq = rng.random(nq) + 0.01
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, alpha)
    assert ctx.stat("search_scan") == 1 and ctx.stat("search_scan_candidates") > 0
    for j in range(0, nq, 7):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, alpha)
        assert np.array_equal(idx[j], oi), j
        assert np.array_equal(sc[j], osc), j


def test_small_batch_search_with_a_useless_sample(ctx):
    """The guess comes from 1 in 16 of the 8-row tiles.  Items ordered so that every sampled tile is far from the query:
    the guess is loose, the candidate list overflows, the query is re-done on the staged path -- same result."""
    n, d = 66_000, 128
    X, _ = synth.mixture(n, d, seed=22, n_clusters=16, rank=5)
    q = X[9] * 1.02
    tiles = np.arange(n) // 8
    sampled = (tiles % 16) == 0
    X[sampled] = -np.abs(X[sampled]) * np.sign(q)      # sampled rows point away from the query
    lam = np.full(n, 0.5)
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    res = asp.search_lambda_aware(A.ArrowItem(q, 0.5), 10, 1.0)
    oi, osc = O.search_lambda_aware(X, lam, q, 0.5, 10, 1.0)
    assert [r[0] for r in res] == oi.tolist() and np.array_equal([r[1] for r in res], osc)
    assert ctx.stat("search_stream_scan") == 1


def test_leftover_graph_rows_take_the_streaming_scan(ctx):
    """Rows the MFMA stages never certify (here: rows that can form a 0 < |a||b| <= 1e-12 pair are always re-done
    exactly) go through one exact pass over the matrix instead of a 300 ms one-wave-per-row scan."""
    n, d = 66_000, 128
    X, _ = synth.mixture(n, d, seed=23, n_clusters=32, rank=6)
    tiny = [40, 9000, 33333, 65999]
    for t in tiny:
        X[t] *= 1e-8
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    import torch
    Xd = torch.from_numpy(X).cuda()
    import ctypes as C
    K = gp.topk + 1
    idx = torch.empty((n, K), dtype=torch.int64, device="cuda")
    dst = torch.empty((n, K), dtype=torch.float64, device="cuda")
    A.api._adopt_torch_stream(ctx)
    c = gp.to_c()
    ctx.check(ctx.lib.as_knn_cosine(ctx.h, Xd.data_ptr(), n, d, 0, n, C.byref(c), idx.data_ptr(), dst.data_ptr()))
    torch.cuda.synchronize()
    assert ctx.stat("knn_join") == 1 and ctx.stat("knn_stream_scan_rows") >= len(tiny)
    sample = np.array(sorted(tiny + [0, 41, 12345, 50000]))
    oi, od = O.knn_cosine_rows(X, ogp(gp), sample)
    pos = torch.from_numpy(sample).cuda()
    assert np.array_equal(idx[pos].cpu().numpy().astype(np.uint64), oi)
    assert np.array_equal(dst[pos].cpu().numpy(), od)


# ---- callers of the same index: hybrid and range search (src/core.rs:802-976) ------------------

def test_hybrid_search_matches_oracle(ctx, mode):
    X, _ = synth.mixture(1500, 48, seed=8, n_clusters=10, rank=4)
    X = np.concatenate([X, X[:20] * 1.000001, X[:5]])  # near-duplicates -> cosine > 0.9999 matches
    lam = np.linspace(0.02, 0.95, len(X))
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    for j, lq, k in ((3, 0.4, 5), (7, 0.11, 10), (700, 0.8, 3), (1, 0.0, 4)):
        q = X[j] * 1.02
        got = asp.search_lambda_aware_hybrid(A.ArrowItem(q, lq), k, 0.7)
        oi, osc = O.search_lambda_aware_hybrid(X, lam, q, lq, k, 0.7)
        assert [g[0] for g in got] == oi.tolist(), (j, k)
        assert np.array_equal([g[1] for g in got], osc), (j, k)
    assert asp.search_lambda_aware_hybrid(A.ArrowItem(X[0], 0.3), 0, 0.7) == []


def test_range_search_matches_oracle(ctx, proteins):
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(proteins)
    lam = np.asarray(asp.lambdas())
    for lq, eps in ((0.3, 0.05), (0.9, 0.0), (0.01, -0.5), (0.5, 10.0)):
        got = asp.range_search(A.ArrowItem(proteins[2], lq), gl, eps)
        oi, od = O.range_search(lam, lq, eps)
        assert [g[0] for g in got] == oi.tolist()
        assert np.array_equal([g[1] for g in got], od)
    # lambda == 0 -> the query is prepared first (core.rs:953-957)
    q = proteins[5] * 1.02
    lq = asp.prepare_query_item(q, gl)
    got = asp.range_search(A.ArrowItem(q, 0.0), gl, 0.02)
    oi, od = O.range_search(lam, lq, 0.02)
    assert [g[0] for g in got] == oi.tolist()


def test_normalise_true_matches_oracle(ctx, mode):
    """normalise = true: StandardScaler z-score per column before the graph (src/laplacian.rs:146-151;
    oracle convention: population variance, zero std -> 1 -- smartcore's own is unpinned)."""
    X, _ = synth.mixture(900, 24, seed=6, n_clusters=6, rank=3)
    X[:, 5] = 3.25  # constant column: std 0 -> 1
    gp = A.GraphParams(0.6, 8, 8, 2.0, 0.5, normalise=True)
    og = O.graph_params(0.6, 8, 8, 2.0, 0.5, normalise=True)
    assert_csr_equal(A.build_laplacian_matrix(X, gp).matrix.to_host(), O.build_laplacian(X, og))


# ---- streaming lambda-tau kernel (n_features == 128, csrc/taumode_v3.hip) ---------------------------

def _lambda_case(ctx, X, csr, tm, om, arg, check=None, rtol=LAMBDA_RTOL):
    L = A.CsrMatrix.from_host(ctx, *csr)
    asp = A.ArrowSpace(X, tm, ctx)
    gl = A.GraphLaplacian(None, L, len(X), None)
    asp.compute_taumode(gl)
    want = O.taumode_lambdas(X, csr, om, arg)
    got = np.asarray(asp.lambdas())
    both_nan = np.isnan(got) & np.isnan(want)
    if check is not None:
        both_nan = both_nan | ~check
    np.testing.assert_allclose(np.where(both_nan, 0.0, got), np.where(both_nan, 0.0, want), rtol=rtol, atol=1e-300)
    return got


V3_MODES = [(A.TauMode.Median, O.TAU_MEDIAN, 0.0), (A.TauMode.Mean, O.TAU_MEAN, 0.0),
            (A.TauMode.Percentile(0.95), O.TAU_PERCENTILE, 0.95), (A.TauMode.Percentile(0.0), O.TAU_PERCENTILE, 0.0),
            (A.TauMode.Percentile(0.5), O.TAU_PERCENTILE, 0.5), (A.TauMode.Fixed(0.4), O.TAU_FIXED, 0.4)]


@pytest.mark.parametrize("tm,om,arg", V3_MODES)
@pytest.mark.parametrize("n", [1, 15, 16, 17, 127, 129, 4099, 70001])
def test_streaming_lambda_ragged_sizes(ctx, n, tm, om, arg):
    """Tiles of 16 items per wave: every ragged tail, one wave up to several tiles per wave."""
    X, cent = synth.mixture(max(n, 64), 128, seed=5, n_clusters=32, rank=6)
    csr = O.build_laplacian(np.ascontiguousarray(cent.T), O.graph_params(0.5, 16, 16, 2.0, 0.25))
    for v3 in (1, 0):  # streaming kernel and the generic kernels agree with the oracle alike
        ctx.set_option("lambda_v3", v3)
        try:
            _lambda_case(ctx, X[:n], csr, tm, om, arg)
        finally:
            ctx.set_option("lambda_v3", 1)


@pytest.mark.parametrize("tm,om,arg", V3_MODES)
def test_streaming_lambda_nonfinite_and_extreme_rows(ctx, tm, om, arg):
    """NaN / +-inf entries (not ranked by select_tau: src/taumode.rs:100-108), all-zero rows, constant
    rows, duplicated values around the middle, magnitudes that overflow the one-pass G term (the
    literal two-pass form takes over), an item that is non-finite everywhere."""
    X, cent = synth.mixture(600, 128, seed=9, n_clusters=32, rank=6)
    csr = O.build_laplacian(np.ascontiguousarray(cent.T), O.graph_params(0.5, 16, 16, 2.0, 0.25))
    X[5, 7] = INF
    X[9, 100] = -INF
    X[12, :] = 0.0
    X[20, 3:90] = NAN
    X[21] *= 1e-160
    X[22] *= 1e150
    X[23] *= 1e-100
    X[24, :] = 3.25
    X[25, :] = NAN
    X[26, 60:70] = X[26, 59]
    X[27, ::2] = INF
    X[28, 1::2] = -1.0
    X[28, ::2] = 2.0
    X[29] = np.abs(X[29])
    X[30] = -np.abs(X[30])
    X[31, :64] = NAN       # one lane quarter-pair entirely non-finite
    X[32, 32:] = INF
    keep = np.ones(len(X), dtype=bool)
    keep[24] = False
    got = _lambda_case(ctx, X, csr, tm, om, arg, keep)
    # A constant row has x'Lx = 0 exactly in real arithmetic; in floating point the Rayleigh sum is
    # pure cancellation noise whose value depends on the summation order (the reference's rayon
    # reduction is unordered there too: src/taumode.rs:565-588), so only its size is comparable.
    want24 = O.taumode_lambdas(X[24:25], csr, om, arg)[0]
    assert abs(got[24]) < 1e-14 and abs(want24) < 1e-14


def test_streaming_lambda_general_csr(ctx):
    """An imported CSR that is not a Laplacian: unsymmetric, positive off-diagonal entries (w = 0
    there: src/taumode.rs:577), missing diagonal entries, fewer nodes than features.  With random signs
    x'Lx is a cancelling sum, so the summation order shows up around 1e-11 relative: the bar here is
    1e-9 (north_star: 1e-6); the point of the case is the semantics, not the conditioning."""
    rng = np.random.default_rng(3)
    n = 100  # graph nodes < 128 features: the denominator still runs over all 128 (src/taumode.rs:596)
    dense = np.where(rng.random((n, n)) < 0.05, rng.normal(size=(n, n)), 0.0)
    dense[np.arange(0, n, 3), np.arange(0, n, 3)] = 0.0
    indptr = np.zeros(n + 1, dtype=np.uint64)
    indices, data = [], []
    for i in range(n):
        js = np.nonzero(dense[i])[0]
        indices.extend(js)
        data.extend(dense[i, js])
        indptr[i + 1] = len(indices)
    csr = (indptr, np.array(indices, dtype=np.uint64), np.array(data, dtype=np.float64))
    X, _ = synth.mixture(3000, 128, seed=4, n_clusters=16, rank=5)
    for tm, om, arg in V3_MODES[:3]:
        _lambda_case(ctx, X, csr, tm, om, arg, rtol=1e-9)
    sym = dense + dense.T  # symmetric but with positive entries
    indices, data = [], []
    for i in range(n):
        js = np.nonzero(sym[i])[0]
        indices.extend(js)
        data.extend(sym[i, js])
        indptr[i + 1] = len(indices)
    csr = (indptr, np.array(indices, dtype=np.uint64), np.array(data, dtype=np.float64))
    _lambda_case(ctx, X, csr, A.TauMode.Median, O.TAU_MEDIAN, 0.0, rtol=1e-9)


def test_streaming_lambda_query_nonfinite_is_an_error(ctx):
    X, cent = synth.mixture(500, 128, seed=5, n_clusters=32, rank=6)
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(X, cent)
    Q = X[:40] * 1.02
    Q[33, 5] = NAN
    with pytest.raises(A.ArrowSpaceError, match="invalid values"):
        asp.prepare_query_items(Q, gl)
    Q[33, 5] = 1e200  # finite, but its square is not: must not be taken for a non-finite value
    lq = asp.prepare_query_items(Q, gl)
    want = np.array([O.prepare_query_item(q, gl.matrix.to_host(), O.TAU_MEDIAN, 0.0) for q in Q])
    np.testing.assert_allclose(lq, want, rtol=LAMBDA_RTOL)


def test_concurrent_searches_on_one_index(ctx):
    """The reference's search takes &self (src/core.rs:760) and ArrowSpace is Sync: several threads, each with its own context,
    search ONE index at the same time -- from its very first search on, so that the lazily built state (images, orders,
    samples: as_index in csrc/common.h) is built while others are reading.  Every result equals the oracle's."""
    import ctypes as C
    import threading
    n, d, k, alpha = 70000, 128, 10, 0.7
    X, cent = synth.mixture(n, d, seed=61, n_clusters=32, rank=6)
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(X, cent)
    lam = np.asarray(asp.lambdas())
    index = asp._ensure_index()
    Xdev = asp._items()
    rng = np.random.default_rng(8)
    batches = {}
    for t, nq in enumerate((1, 7, 1500, 1, 33, 2048)):   # small batches (streamed), large ones (threshold scan), twice each kind
        qi = rng.choice(n, size=nq, replace=False)
        Q = np.ascontiguousarray(X[qi] * 1.02)
        batches[t] = (Q, np.asarray(asp.prepare_query_items(Q, gl)))
    results, errors = {}, []

    def worker(t):
        try:
            c = A.Context(0)          # a context per thread; the index is shared
            Q, lq = batches[t]
            nq = Q.shape[0]
            for rep in range(3):      # the first call builds state, the later ones find it (or find it being rebuilt in order)
                idx = np.empty((nq, k), dtype=np.uint64)
                sc = np.empty((nq, k), dtype=np.float64)
                c.check(c.lib.as_search_lambda_aware(c.h, index, _lib.ptr(Q), _lib.ptr(lq), nq, k, alpha, 0, _lib.ptr(idx), _lib.ptr(sc)))
                results[(t, rep)] = (idx, sc)
            c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in batches]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t, (Q, lq) in batches.items():
        for j in np.linspace(0, Q.shape[0] - 1, min(Q.shape[0], 6)).astype(int):
            oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
            for rep in range(3):
                idx, sc = results[(t, rep)]
                assert np.array_equal(idx[j], oi), (t, rep, j)
                assert np.array_equal(sc[j], osc), (t, rep, j)
        for rep in (1, 2):   # and the same batch gives the same table every time
            assert np.array_equal(results[(t, rep)][0], results[(t, 0)][0]) and np.array_equal(results[(t, rep)][1], results[(t, 0)][1])


def test_large_batch_after_a_small_one_on_the_same_index(ctx):
    """An index's second search finds the items' directions for whichever path runs; the rotated image by position belongs to the
    large-batch scan, the one in item order to the small-batch filter.  A small batch in between used to leave the scan's image
    unbuilt (null first-pass operand).  Sequence: large, small, large, small -- every table the oracle's."""
    n, d, k, alpha = 70000, 128, 10, 0.7
    X, cent = synth.mixture(n, d, seed=62, n_clusters=32, rank=6)
    asp, gl = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(X, cent)
    lam = np.asarray(asp.lambdas())
    rng = np.random.default_rng(9)
    for nq in (1200, 3, 2100, 1, 2100):
        qi = rng.choice(n, size=nq, replace=False)
        Q = np.ascontiguousarray(X[qi] * 1.02)
        lq = np.asarray(asp.prepare_query_items(Q, gl))
        idx, sc = asp.search_lambda_aware_batch(Q, lq, k, alpha)
        for j in np.linspace(0, nq - 1, min(nq, 5)).astype(int):
            oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
            assert np.array_equal(np.asarray(idx)[j], oi.astype(np.int64)) and np.array_equal(np.asarray(sc)[j], osc), (nq, j)
    assert ctx.stat("search_scan_prefilter") == 1   # the last large batch ran behind the first pass
