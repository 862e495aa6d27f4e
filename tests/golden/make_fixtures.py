#!/usr/bin/env python3
"""Generate tests/golden/*.json.  Runs ONLY in the build container (reads /root/reference).

Two kinds of fixture are written:

* DATA fixtures -- literal numeric datasets that the reference's own tests/benches hold
  (inputs only, no source text):
    proteins_64x24.json   <- benches/index_compute_bench.rs:21-86 (VECTORS_DATA rows)
    quora_15x384.json     <- src/tests/test_data.rs:6-5797        (QUORA_EMBEDDS)
    proteins_10.json      <- src/tests/test_data.rs:5801-6252     (PROTEINS_EMBEDS)
* ORACLE vectors -- oracle_vectors.json: outputs of oracle/liboracle.so (the CPU restatement)
  on those datasets, on the literal matrices of src/tests/test_laplacian.rs and on small seeded
  synthetic sets.  The reference is a Rust crate and cannot be built here (no rustc), so these
  are regression pins of the restatement, NOT outputs of the reference; the restatement itself
  is pinned against the reference's known-answer tests in tests/test_oracle_golden.py.

Usage:  python tests/golden/make_fixtures.py
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import oracle_ctypes as O  # noqa: E402
from tests import synth  # noqa: E402

NUM = r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?"


def parse_proteins_block():
    txt = open(os.path.join(REF, "benches/index_compute_bench.rs")).read()
    rows = []
    for line in txt.splitlines():
        m = re.match(r"\s*P\d{4,5};\s*(.*)$", line)
        if m:
            rows.append([float(x) for x in m.group(1).split(",")])
    assert len(rows) == 64 and all(len(r) == 24 for r in rows), (len(rows),)
    return rows


def parse_const_matrix(name):
    txt = open(os.path.join(REF, "src/tests/test_data.rs")).read()
    start = txt.index("pub const %s" % name)
    start = txt.index("= &[", start) + 4
    end = txt.index("\n];", start)
    body = txt[start:end]
    rows = []
    for blk in re.findall(r"&\[(.*?)\]", body, flags=re.S):
        vals = [float(x) for x in re.findall(NUM, blk)]
        if vals:
            rows.append(vals)
    return rows


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f)
    print("wrote", path, os.path.getsize(path), "bytes")


def csr_to_json(csr):
    indptr, indices, data = csr
    return {"indptr": [int(x) for x in indptr], "indices": [int(x) for x in indices],
            "data": [float(x).hex() for x in data]}


DATASETS = {}


def case(name, rows, eps, k, topk, p, sigma, transpose, queries=None, tau=(O.TAU_MEDIAN, 0.0),
         alpha=0.7, kq=3, dataset=None):
    """One oracle vector: graph over `nodes` (= rows^T when transpose, as build() does via
    graph.rs:170-172), lambdas for the item rows when legal (L.n <= n_features)."""
    rows = np.asarray(rows, dtype=np.float64)
    nodes = rows.T.copy() if transpose else rows
    gp = O.graph_params(eps, k, topk, p, sigma)
    if dataset is None:
        dataset = name
    if dataset not in DATASETS:
        DATASETS[dataset] = [[float(x).hex() for x in r] for r in rows]
    out = {"name": name, "dataset": dataset, "transpose": bool(transpose),
           "params": {"eps": eps, "k": k, "topk": topk, "p": p, "sigma": sigma},
           "tau_mode": tau[0], "tau_arg": tau[1]}
    nbr_idx, nbr_dist = O.knn_cosine(nodes, gp)
    out["knn_idx"] = [[int(x) if x != O.U64_MAX else -1 for x in r] for r in nbr_idx]
    out["knn_dist"] = [[float(x).hex() for x in r] for r in nbr_dist]
    csr = O.build_laplacian(nodes, gp)
    out["csr"] = csr_to_json(csr)
    n_graph = len(csr[0]) - 1
    if n_graph <= rows.shape[1]:
        lam = O.taumode_lambdas(rows, csr, tau[0], tau[1])
        out["lambdas"] = [float(x).hex() for x in lam]
        if queries is not None:
            res = []
            for q in queries:
                lq = O.prepare_query_item(q, csr, tau[0], tau[1])
                entry = {"q": [float(x) for x in q], "lambda_q": float(lq).hex()}
                if lq != 0.0:
                    idx, sc = O.search_lambda_aware(rows, lam, q, lq, kq, alpha)
                    entry.update({"k": kq, "alpha": alpha, "idx": [int(x) for x in idx],
                                  "scores": [float(x).hex() for x in sc]})
                res.append(entry)
            out["queries"] = res
    return out


def main():
    proteins = parse_proteins_block()
    dump("proteins_64x24.json", {"source": "benches/index_compute_bench.rs:21-86", "rows": proteins})
    quora = parse_const_matrix("QUORA_EMBEDDS")
    assert len(quora) == 15 and all(len(r) == 384 for r in quora), (len(quora), len(quora[0]))
    dump("quora_15x384.json", {"source": "src/tests/test_data.rs:6-5797", "rows": quora})
    prot10 = parse_const_matrix("PROTEINS_EMBEDS")
    assert len(set(len(r) for r in prot10)) == 1, [len(r) for r in prot10]
    dump("proteins_10.json", {"source": "src/tests/test_data.rs:5801-6252", "rows": prot10})

    P = np.asarray(proteins)
    cases = []
    # literal matrices of src/tests/test_laplacian.rs (line numbers cited per case)
    cases.append(case("lap_5x3_l12", [[1, 0, 0], [.8, .6, 0], [0, 1, 0], [0, .8, .6], [0, 0, 1]],
                      0.5, 3, 2, 2.0, 0.1, True))
    cases.append(case("lap_4x2_l157", [[1, 0], [.707, .707], [0, 1], [-1, 0]], 2.0, 3, 2, 1.0, 0.5, False))
    cases.append(case("lap_3x2_l215", [[1, 0], [.5, 0], [0, 1]], 0.3, 10, 4, 2.0, 0.1, False))
    cases.append(case("lap_4x5_l475", [[1, 0, 0, 0, 0], [.9, .1, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 0, 1, 0]],
                      0.5, 4, 3, 2.0, 0.1, True))
    cases.append(case("lap_4x3_l558", [[1, 0, 0], [.8, .2, 0], [0, 1, 0], [0, .8, .2]], 0.8, 2, 1, 2.0, 0.1, True))
    cases.append(case("lap_3x2_l686", [[1, 0], [.9, .1], [0, 1]], 0.5, 2, 1, 1.0, 0.2, False))
    # 8x3 of src/tests/test_builder.rs:14-23 shape class: items as nodes AND feature graph
    # 64x24 dataset: feature graph over the 64 rows used as "centroids" (ref-mapping with X = N)
    q3 = (P[3] * 1.02).tolist()
    q10 = (P[10] * 1.02).tolist()
    cases.append(case("proteins_ref_map", P, 0.5, 16, 16, 2.0, 0.25, True, queries=[q3, q10], dataset="proteins"))
    cases.append(case("proteins_ref_map_default", P, 1e-3, 3, 3, 2.0, None, True, queries=[q3], dataset="proteins"))
    cases.append(case("proteins_item_graph", P, 0.05, 6, 4, 2.0, None, False, dataset="proteins"))
    cases.append(case("quora_item_graph", quora, 0.9, 6, 4, 2.0, 0.5, False, dataset="quora"))
    # seeded synthetic mixture (generator in tests/synth.py), both mappings, all tau modes
    X, cent = synth.mixture(300, 32, seed=7, n_clusters=12, rank=4)
    qs = (X[[5, 17]] * 1.02).tolist()
    cases.append(case("mix300x32_ref_median", X, 0.5, 16, 16, 2.0, 0.25, True, queries=qs, dataset="mix300x32"))
    cases.append(case("mix300x32_item_graph", X, 0.5, 16, 16, 2.0, 0.25, False, dataset="mix300x32"))
    cases.append(case("mix300x32_ref_mean", X, 0.5, 8, 8, 2.0, 0.25, True, queries=qs, tau=(O.TAU_MEAN, 0.0), dataset="mix300x32"))
    cases.append(case("mix300x32_ref_pct", X, 0.5, 8, 8, 1.5, 0.25, True, queries=qs, tau=(O.TAU_PERCENTILE, 0.95), dataset="mix300x32"))
    cases.append(case("mix300x32_ref_fixed", X, 0.5, 8, 8, 2.0, None, True, queries=qs, tau=(O.TAU_FIXED, 0.3), dataset="mix300x32"))
    dump("oracle_vectors.json", {"generator": "tests/golden/make_fixtures.py (oracle/liboracle.so)",
                                 "datasets": DATASETS, "cases": cases})


if __name__ == "__main__":
    main()
