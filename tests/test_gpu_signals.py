"""Item-graph lambda (SURVEY.md section 8(a'), row a11 in its shape-generic form): the feature COLUMNS of the item matrix as
signals over the N x N Laplacian of the items, through as_compute_taumode_signals / as_compute_taumode, against the oracle's
aso_synthetic_lambda + aso_select_tau (src/taumode.rs:552-660, 87-127).  Bar: 1e-12 relative (north_star allows 1e-6).

Needs a real MI355X: run with `pytest -m gpu`.
"""
import numpy as np
import pytest

try:
    import torch  # noqa: F401
except ImportError:
    torch = None

import arrowspace_amd as A
from arrowspace_amd import _lib
from arrowspace_amd.api import compute_taumode_signals
from oracle import oracle_ctypes as O
from tests import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-12
T = A.TauMode
MODES = [(T.Median, O.TAU_MEDIAN, 0.0), (T.Mean, O.TAU_MEAN, 0.0), (T.Fixed(0.37), O.TAU_FIXED, 0.37),
         (T.Percentile(0.95), O.TAU_PERCENTILE, 0.95), (T.Percentile(0.0), O.TAU_PERCENTILE, 0.0)]


@pytest.fixture(scope="module")
def ctx():
    return A.default_context()


def oracle_signals(X, csr, omode, oarg):
    """lambda of every column of X as a signal: the oracle's per-item function on the transposed matrix"""
    return O.taumode_lambdas(np.ascontiguousarray(X.T), csr, omode, oarg)


def assert_lambdas(got, want, rtol=RTOL, atol=1e-300):
    got, want = np.asarray(got), np.asarray(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=rtol, atol=atol)


def random_csr(rng, n, deg, symmetric, laplacian=True, drop_diag=0.0, empty_rows=0.0):
    """a sparse matrix with sorted columns; laplacian: negative off-diagonals and the row sums on the diagonal"""
    rows = []
    pairs = {}
    for i in range(n):
        for j in rng.choice(n, size=deg, replace=False):
            j = int(j)
            if j == i:
                continue
            w = float(rng.random()) + 0.05
            if not laplacian and rng.random() < 0.3:
                w = -w   # a positive off-diagonal entry: it counts for x'Lx and not for the edge energy (taumode.rs:578)
            pairs[(i, j)] = w
            if symmetric:
                pairs[(j, i)] = w
    by_row = [[] for _ in range(n)]
    for (i, j), w in pairs.items():
        by_row[i].append((j, -w))
    indptr, indices, data = [0], [], []
    for i in range(n):
        ent = by_row[i]
        if rng.random() < empty_rows:
            ent = []
        elif rng.random() >= drop_diag:
            ent = ent + [(i, -sum(v for _, v in ent) if laplacian else float(rng.standard_normal()))]
        ent.sort()
        for j, v in ent:
            indices.append(j)
            data.append(v)
        indptr.append(len(indices))
    return (np.array(indptr, dtype=np.uint64), np.array(indices, dtype=np.uint64), np.array(data, dtype=np.float64))


@pytest.mark.parametrize("tm,omode,oarg", MODES, ids=["median", "mean", "fixed", "p95", "p0"])
def test_signals_over_a_built_item_graph(ctx, tm, omode, oarg):
    """the library's own graph (bitwise symmetric: streamed as its upper triangle), 128 signals, all tau modes"""
    X, _ = synth.mixture(6000, 128, seed=5, n_clusters=24, rank=8)
    gl = A.build_laplacian_matrix(X, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
    want = oracle_signals(X, gl.matrix.to_host(), omode, oarg)
    got = compute_taumode_signals(X, gl.matrix, tm, ctx=ctx)
    assert ctx.stat("signal_stream_sym") == 1
    assert_lambdas(got, want)
    assert np.all(want > 0) and len(np.unique(want)) > 100   # a real spread of values, not zeros


@pytest.mark.parametrize("n,n_len,S", [(500, 500, 1), (777, 1000, 7), (3000, 3001, 130), (4096, 5000, 256), (100, 100, 129)])
@pytest.mark.parametrize("symmetric", [True, False])
def test_signals_generic_shapes(ctx, n, n_len, S, symmetric):
    """imported graphs: unsymmetric, positive off-diagonals, missing diagonals, empty rows, graph smaller than the signal
    (the denominator runs over all entries, taumode.rs:596), odd widths (8-byte loads), several 128-column slabs"""
    rng = np.random.default_rng(n * 7 + S + symmetric)
    csr = random_csr(rng, n, 6, symmetric, laplacian=symmetric, drop_diag=0.1, empty_rows=0.0 if symmetric else 0.05)
    X = rng.standard_normal((n_len, S)) * rng.uniform(0.1, 10.0, size=S)
    m = A.CsrMatrix.from_host(ctx, *csr)
    for tm, omode, oarg in MODES[:2] + MODES[3:4]:
        want = oracle_signals(X, csr, omode, oarg)
        got = compute_taumode_signals(X, m, tm, ctx=ctx)
        assert_lambdas(got, want)
    assert ctx.stat("signal_stream_sym") == int(symmetric)


def test_order_statistics_of_long_columns(ctx):
    """tau of a column = median / percentile of N values: even and odd counts, duplicates, a constant column, non-finite
    entries (filtered, taumode.rs:99-107), a column without a finite value (the floor)"""
    rng = np.random.default_rng(11)
    for n_len in (9999, 10000):
        X = rng.standard_normal((n_len, 12))
        X[:, 1] = np.round(X[:, 1], 1)            # heavy duplicates around the median
        X[:, 2] = 3.25                             # constant
        X[:, 3] = np.abs(X[:, 3]) * 1e-300        # tiny, subnormal-range magnitudes
        X[::7, 4] = np.nan
        X[::5, 5] = np.inf
        X[::3, 5] = -np.inf
        X[:, 6] = np.nan                           # no finite value
        X[:, 7] = rng.integers(0, 3, size=n_len)   # three distinct values
        X[:, 8] = -np.abs(X[:, 8])                 # all negative: tau floors
        X[:, 9] = np.where(rng.random(n_len) < 0.5, 1.0, 2.0)   # the two middle values may differ
        csr = random_csr(rng, 2000, 5, True)
        m = A.CsrMatrix.from_host(ctx, *csr)
        for tm, omode, oarg in MODES:
            want = oracle_signals(X, csr, omode, oarg)
            got = compute_taumode_signals(X, m, tm, ctx=ctx)
            # (the constant column: x'Lx is 0 up to the rounding of a cancelling sum, ~1e-16 in any order of summation)
            assert_lambdas(got, want, atol=1e-14)
        # and tau itself (as_select_tau takes slices this long through the same kernels), bit for bit
        for tm, omode, oarg in MODES:
            got = A.select_tau(np.ascontiguousarray(X.T), tm, ctx=ctx)
            want = np.array([O.select_tau(X[:, c], omode, oarg) for c in range(X.shape[1])])
            if omode == O.TAU_MEAN:
                np.testing.assert_allclose(got, want, rtol=1e-13)   # the mean's sum is taken in another order
            else:
                assert np.array_equal(got, want), (tm, got, want)


@pytest.mark.parametrize("n_len", [200_000, 200_001, 131_072])
def test_order_statistics_of_very_long_columns_from_a_sampled_bracket(ctx, n_len):
    """"signal_tau_bracket" (round 5): columns of 131 072 values or more get their order statistic from 8 192 sampled values that
    bracket it and ONE pass that counts below the bracket and lists what lies inside.  Same value bit for bit as the oracle's
    select_tau and as the digit passes; smooth columns of any scale, heavy tails, non-finite entries, even and odd counts take it
    (no column falls back), columns of few distinct values run their lists over and hand ALL columns to the digit passes."""
    rng = np.random.default_rng(17)
    X = rng.standard_normal((n_len, 10))
    X[:, 1] = np.exp(4.0 * X[:, 1])                      # eight decades
    X[:, 2] = rng.standard_cauchy(n_len)                 # heavy tails
    X[:, 3] = np.abs(X[:, 3]) * 1e-300                   # subnormal range
    X[::7, 4] = np.nan
    X[::5, 5] = np.inf
    X[::3, 5] = -np.inf
    X[:, 6] = -np.abs(X[:, 6])                           # all negative: tau floors
    X[:, 7] = np.arange(n_len)[::-1] * 0.5               # sorted, every value once
    X[:, 8] = np.round(X[:, 8], 4)                       # a few hundred copies of each value near the median
    modes = [(T.Median, O.TAU_MEDIAN, 0.0), (T.Percentile(0.95), O.TAU_PERCENTILE, 0.95), (T.Percentile(0.0), O.TAU_PERCENTILE, 0.0),
             (T.Percentile(1.0), O.TAU_PERCENTILE, 1.0), (T.Percentile(0.5), O.TAU_PERCENTILE, 0.5)]
    Xt = np.ascontiguousarray(X.T)
    for tm, omode, oarg in modes:
        want = np.array([O.select_tau(X[:, c], omode, oarg) for c in range(X.shape[1])])
        got = A.select_tau(Xt, tm, ctx=ctx)
        assert ctx.stat("signal_tau_bracket_fallback_columns") == 0, tm
        assert np.array_equal(got, want), (tm, got, want)
        ctx.set_option("signal_tau_bracket", 0)
        try:
            assert np.array_equal(A.select_tau(Xt, tm, ctx=ctx), want)
        finally:
            ctx.set_option("signal_tau_bracket", 1)
    Y = X.copy()
    Y[:, 0] = 3.25                                        # constant
    Y[:, 1] = rng.integers(0, 3, size=n_len)              # three distinct values
    Y[:, 2] = np.nan                                      # no finite value (the floor; no fallback needed for it)
    Y[:, 9] = np.where(rng.random(n_len) < 0.5, 1.0, 2.0)
    Yt = np.ascontiguousarray(Y.T)
    for tm, omode, oarg in modes[:2]:
        want = np.array([O.select_tau(Y[:, c], omode, oarg) for c in range(Y.shape[1])])
        got = A.select_tau(Yt, tm, ctx=ctx)
        assert ctx.stat("signal_tau_bracket_fallback_columns") >= 3, tm
        assert np.array_equal(got, want), (tm, got, want)


def test_any_order_gives_the_same_lambdas(ctx):
    X, _ = synth.mixture(5000, 64, seed=9, n_clusters=16, rank=6)
    gl = A.build_laplacian_matrix(X, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
    want = oracle_signals(X, gl.matrix.to_host(), O.TAU_MEDIAN, 0.0)
    rng = np.random.default_rng(3)
    base = None
    for order in (None, rng.permutation(5000), np.arange(5000)[::-1].copy(), np.argsort(np.arange(5000) % 16, kind="stable")):
        gl.matrix.set_order(order)
        got = compute_taumode_signals(X, gl.matrix, T.Median, ctx=ctx)
        assert ctx.stat("signal_stream_ordered") == int(order is not None)
        assert_lambdas(got, want)
        again = compute_taumode_signals(X, gl.matrix, T.Median, ctx=ctx)
        assert np.array_equal(got, again)   # bit-identical from run to run
        base = got if base is None else base
    for bad in (np.zeros(5000, dtype=np.uint32), np.arange(5000, dtype=np.uint32) + 1):
        with pytest.raises(A.ArrowSpaceError) as e:
            gl.matrix.set_order(bad)
        assert e.value.status == _lib.AS_ERR_ARG


def test_extreme_magnitudes_take_the_literal_second_pass(ctx):
    """sum t^2 overflows although every share t / edge is fine: the reference's own second pass (taumode.rs:611-639)"""
    rng = np.random.default_rng(21)
    csr = random_csr(rng, 1500, 6, True)
    X = rng.standard_normal((1500, 10))
    X[:, :5] *= 1e90      # t ~ 1e180, t^2 overflows
    X[:, 5:8] *= 1e-90    # t ~ 1e-180, t^2 underflows to 0
    m = A.CsrMatrix.from_host(ctx, *csr)
    for sym in (1, 0):
        ctx.set_option("signal_sym", sym)
        m.set_order(None)   # drops the stream: rebuilt under the option
        want = oracle_signals(X, csr, O.TAU_FIXED, 0.5)
        got = compute_taumode_signals(X, m, T.Fixed(0.5), ctx=ctx)
        assert ctx.stat("signal_stream_sym") == sym
        assert np.all(np.isfinite(want)) and np.all(want > 0)
        assert_lambdas(got, want)
    ctx.set_option("signal_sym", 1)


def test_compute_taumode_takes_long_signals(ctx):
    """as_compute_taumode with items longer than any per-item kernel holds (the former AS_ERR_UNSUPPORTED): 20 signals of
    12,000 values against a 12,000-node graph, and against a smaller one"""
    rng = np.random.default_rng(5)
    n_len = 12000
    items = rng.standard_normal((20, n_len))
    for n in (n_len, 7000):
        csr = random_csr(rng, n, 5, True)
        m = A.CsrMatrix.from_host(ctx, *csr)
        asp = A.ArrowSpace(items, T.Median, ctx=ctx)
        asp.compute_taumode(A.GraphLaplacian(None, m, n_len, None))
        want = O.taumode_lambdas(items, csr, O.TAU_MEDIAN)
        assert_lambdas(asp.lambdas(), want)
        # prepare_query_item on the same route: a NaN is refused as the reference refuses it (core.rs:534-537)
        lq = asp.prepare_query_item(items[3], A.GraphLaplacian(None, m, n_len, None))
        assert abs(lq - want[3]) <= RTOL * abs(want[3])
        q = items[4].copy()
        q[77] = np.nan
        with pytest.raises(A.ArrowSpaceError) as e:
            asp.prepare_query_item(q, A.GraphLaplacian(None, m, n_len, None))
        assert e.value.status == _lib.AS_ERR_NONFINITE
    # a graph with more nodes than the signal has entries: the reference would index out of bounds (taumode.rs:570-575)
    big = A.CsrMatrix.from_host(ctx, *random_csr(rng, n_len + 1, 3, True))
    with pytest.raises(A.ArrowSpaceError) as e:
        compute_taumode_signals(items.T.copy(), big, T.Median, ctx=ctx)
    assert e.value.status == _lib.AS_ERR_DIM


@pytest.mark.skipif(torch is None, reason="torch")
def test_signals_device_resident(ctx):
    X, _ = synth.mixture(4000, 128, seed=2, n_clusters=16, rank=8)
    gl = A.build_laplacian_matrix(X, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
    want = oracle_signals(X, gl.matrix.to_host(), O.TAU_MEDIAN, 0.0)
    Xd = torch.from_numpy(X).cuda()
    got = compute_taumode_signals(Xd, gl.matrix, T.Median, ctx=ctx)
    assert got.is_cuda
    ctx.synchronize()
    assert_lambdas(got.cpu().numpy(), want)
