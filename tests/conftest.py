import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unhex(x):
    return float.fromhex(x) if isinstance(x, str) else float(x)


def unhex_arr(a):
    return np.array([[unhex(v) for v in r] for r in a], dtype=np.float64) if a and isinstance(a[0], list) \
        else np.array([unhex(v) for v in a], dtype=np.float64)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "oracle_vectors.json")) as f:
        g = json.load(f)
    ds = {k: unhex_arr(v) for k, v in g["datasets"].items()}
    cases = {}
    for c in g["cases"]:
        c = dict(c)
        c["rows"] = ds[c["dataset"]]
        c["nodes"] = np.ascontiguousarray(c["rows"].T) if c["transpose"] else c["rows"]
        c["csr"] = (np.array(c["csr"]["indptr"], dtype=np.uint64), np.array(c["csr"]["indices"], dtype=np.uint64),
                    unhex_arr(c["csr"]["data"]))
        c["knn_idx"] = np.array(c["knn_idx"], dtype=np.int64)
        c["knn_dist"] = unhex_arr(c["knn_dist"])
        if "lambdas" in c:
            c["lambdas"] = unhex_arr(c["lambdas"])
        for q in c.get("queries", []):
            q["lambda_q"] = unhex(q["lambda_q"])
            if "scores" in q:
                q["scores"] = unhex_arr(q["scores"])
        cases[c["name"]] = c
    return cases


@pytest.fixture(scope="session")
def proteins():
    with open(os.path.join(GOLDEN, "proteins_64x24.json")) as f:
        return np.array(json.load(f)["rows"], dtype=np.float64)


@pytest.fixture(scope="session")
def quora():
    with open(os.path.join(GOLDEN, "quora_15x384.json")) as f:
        return np.array(json.load(f)["rows"], dtype=np.float64)
