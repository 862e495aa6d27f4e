"""On-disk formats of the reference for what the device produces (src/storage/parquet.rs), written with pyarrow so
that a Rust host reads them with the crate's own loaders and vice versa:

  dense matrix   `<name>.parquet`: name_id (utf8), n_rows, n_cols (u64), col_0 .. col_{F-1} (f64), one record per
                 row                                                            src/storage/parquet.rs:192-289
  sparse matrix  `<name>.parquet`: name_id, n_rows, n_cols, nnz (u64), row, col (u64), value (f64): COO triplets in
                 CSR order                                                      src/storage/parquet.rs:354-460
  lambda vector  `<name>.parquet`: name_id, n_values, row_index (u64), lambda (f64)  src/storage/parquet.rs:664-760
  metadata       `<name>_metadata.json` sidecar = serde_json of ArrowSpaceMetadata (src/storage/parquet.rs:29-54): name_id,
                 timestamp (RFC 3339), n_rows, n_cols, builder_config (HashMap<String, ConfigValue>: an externally tagged
                 enum, {"Usize": 3}, src/builder.rs:528-538), files (HashMap<String, FileInfo>; key "matrix" for the
                 matrices, "lambda_vector" with file_type "lambda_vector" for the lambdas, :715-742)
Snappy compression, one record batch, as the reference writes them.  Host-side I/O: nothing here touches the GPU
except reading the CSR back from HBM (CsrMatrix.to_host).
"""
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq


def _write(table, path, name_id):
    os.makedirs(path, exist_ok=True)
    fp = os.path.join(path, "%s.parquet" % name_id)
    pq.write_table(table, fp, compression="snappy")
    return fp


# ConfigValue variants of the keys ArrowSpaceBuilder::builder_config_typed writes (src/builder.rs:580-634)
_OPTION_KEYS = {"lambda_sigma": "OptionF64", "cluster_max_clusters": "OptionUsize", "clustering_seed": "OptionU64",
                "sampling": "OptionSamplerType"}
_TAGS = ("Bool", "Usize", "F64", "String", "OptionF64", "OptionUsize", "OptionU64", "TauMode", "OptionSamplerType")


def _tau_json(t):
    """serde's form of TauMode (src/taumode.rs:75-82): "Median" / "Mean" / {"Fixed": x} / {"Percentile": p}."""
    kind, arg = getattr(t, "kind", None), getattr(t, "arg", 0.0)
    return {0: {"Fixed": float(arg)}, 1: "Median", 2: "Mean", 3: {"Percentile": float(arg)}}[int(kind)]


def config_value(key, v):
    """One builder_config entry as serde_json writes ConfigValue (externally tagged: {"Variant": payload})."""
    if isinstance(v, dict) and len(v) == 1 and next(iter(v)) in _TAGS:
        return v                                   # already tagged
    if key in _OPTION_KEYS:
        return {_OPTION_KEYS[key]: v}
    if hasattr(v, "kind") and hasattr(v, "arg"):
        return {"TauMode": _tau_json(v)}
    if isinstance(v, (bool, np.bool_)):
        return {"Bool": bool(v)}
    if isinstance(v, (int, np.integer)):
        return {"Usize": int(v)}
    if isinstance(v, (float, np.floating)):
        return {"F64": float(v)}
    if isinstance(v, str):
        return {"String": v}
    raise TypeError("builder_config[%r]: no ConfigValue variant for %r" % (key, type(v)))


def builder_config_typed(builder):
    """ArrowSpaceBuilder::builder_config_typed (src/builder.rs:580-634) for this package's builder mirror."""
    g = lambda name, default=None: getattr(builder, name, default)  # noqa: E731
    cfg = {"prebuilt_spectral": bool(g("prebuilt_spectral", False)), "lambda_eps": float(g("lambda_eps")),
           "lambda_k": int(g("lambda_k")), "lambda_topk": int(g("lambda_topk")), "lambda_p": float(g("lambda_p")),
           "lambda_sigma": g("lambda_sigma"), "normalise": bool(g("normalise", False)),
           "sparsity_check": bool(g("sparsity_check", False)), "synthesis": g("synthesis"), "sampling": None,
           "cluster_max_clusters": g("cluster_max_clusters"), "cluster_radius": float(g("cluster_radius", 1.0) or 0.0),
           "clustering_seed": g("clustering_seed"), "deterministic_clustering": bool(g("deterministic_clustering", False)),
           "use_dims_reduction": bool(g("use_dims_reduction", False)), "rp_eps": float(g("rp_eps", 0.3) or 0.0)}
    return {k: config_value(k, v) for k, v in cfg.items()}


def _metadata(path, name_id, file_type, rows, cols, nnz, fp, builder_config, key="matrix"):
    if builder_config is None:
        return
    import datetime
    meta = {"name_id": name_id, "timestamp": datetime.datetime.now(datetime.timezone.utc).isoformat(),  # chrono to_rfc3339
            "n_rows": int(rows), "n_cols": int(cols),
            "builder_config": {k: config_value(k, v) for k, v in builder_config.items()},
            "files": {key: {"filename": os.path.basename(fp), "file_type": file_type, "rows": int(rows), "cols": int(cols),
                            "nnz": None if nnz is None else int(nnz), "size_bytes": os.path.getsize(fp)}}}
    with open(os.path.join(path, "%s_metadata.json" % name_id), "w") as f:
        json.dump(meta, f, indent=1)


def save_dense_matrix(matrix, path, name_id, builder_config=None):
    """src/storage/parquet.rs:192-289"""
    M = np.asarray(matrix, dtype=np.float64)
    n_rows, n_cols = M.shape
    cols = {"name_id": pa.array([name_id] * n_rows, type=pa.string()),
            "n_rows": pa.array(np.full(n_rows, n_rows, dtype=np.uint64)),
            "n_cols": pa.array(np.full(n_rows, n_cols, dtype=np.uint64))}
    for c in range(n_cols):
        cols["col_%d" % c] = pa.array(np.ascontiguousarray(M[:, c]))
    fp = _write(pa.table(cols), path, name_id)
    _metadata(path, name_id, "dense", n_rows, n_cols, None, fp, builder_config)
    return fp


def load_dense_matrix(file_path):
    """src/storage/parquet.rs:291-352"""
    t = pq.read_table(file_path)
    cols = [c for c in t.column_names if c.startswith("col_")]
    if t.num_rows == 0:   # (the shape columns repeat per record: an empty matrix carries its width only in the schema)
        return np.zeros((0, len(cols)), dtype=np.float64)
    n_cols = int(t.column("n_cols")[0].as_py())
    return np.stack([t.column("col_%d" % c).to_numpy() for c in range(n_cols)], axis=1)


def save_sparse_matrix(csr, path, name_id, builder_config=None, shape=None):
    """src/storage/parquet.rs:354-460.  `csr`: a CsrMatrix (device handle) or an (indptr, indices, data) triple."""
    indptr, indices, data = csr.to_host() if hasattr(csr, "to_host") else csr
    n_rows = len(indptr) - 1
    n_cols = shape[1] if shape else (csr.shape()[1] if hasattr(csr, "shape") else n_rows)
    nnz = len(indices)
    rows = np.repeat(np.arange(n_rows, dtype=np.uint64), np.diff(np.asarray(indptr).astype(np.int64)))
    table = pa.table({"name_id": pa.array([name_id] * nnz, type=pa.string()),
                      "n_rows": pa.array(np.full(nnz, n_rows, dtype=np.uint64)),
                      "n_cols": pa.array(np.full(nnz, n_cols, dtype=np.uint64)),
                      "nnz": pa.array(np.full(nnz, nnz, dtype=np.uint64)),
                      "row": pa.array(rows), "col": pa.array(np.asarray(indices, dtype=np.uint64)),
                      "value": pa.array(np.asarray(data, dtype=np.float64))})
    fp = _write(table, path, name_id)
    _metadata(path, name_id, "sparse", n_rows, n_cols, nnz, fp, builder_config)
    return fp


def load_sparse_matrix(file_path):
    """src/storage/parquet.rs:462-560: (indptr, indices, data), shape."""
    t = pq.read_table(file_path)
    if t.num_rows == 0:   # nnz = 0: no record carries the shape (the reference's loader fails on such a file too)
        raise ValueError("%s holds no entries: the COO layout keeps the shape in its records" % file_path)
    n_rows, n_cols = int(t.column("n_rows")[0].as_py()), int(t.column("n_cols")[0].as_py())
    row, col, val = (t.column(c).to_numpy() for c in ("row", "col", "value"))
    order = np.lexsort((col, row))
    row, col, val = row[order], col[order], val[order]
    indptr = np.zeros(n_rows + 1, dtype=np.uint64)
    np.add.at(indptr, row.astype(np.int64) + 1, 1)
    return (np.cumsum(indptr).astype(np.uint64), col.astype(np.uint64), val.astype(np.float64)), (n_rows, n_cols)


def save_lambda(lambdas, path, name_id, builder_config=None):
    """src/storage/parquet.rs:664-760"""
    lam = np.asarray(lambdas.cpu().numpy() if hasattr(lambdas, "cpu") else lambdas, dtype=np.float64)
    n = len(lam)
    table = pa.table({"name_id": pa.array([name_id] * n, type=pa.string()),
                      "n_values": pa.array(np.full(n, n, dtype=np.uint64)),
                      "row_index": pa.array(np.arange(n, dtype=np.uint64)), "lambda": pa.array(lam)})
    fp = _write(table, path, name_id)
    _metadata(path, name_id, "lambda_vector", n, 1, None, fp, builder_config, key="lambda_vector")
    return fp


def load_lambda(file_path):
    t = pq.read_table(file_path)
    idx, lam = t.column("row_index").to_numpy(), t.column("lambda").to_numpy()
    out = np.empty(len(lam), dtype=np.float64)
    out[idx.astype(np.int64)] = lam
    return out


def save_arrowspace(aspace, gl, path, name_id, builder_config=None):
    """What the builder persists when a path is configured (src/builder.rs:327-360): the raw items, the Laplacian
    and the lambdas, each under its own name."""
    data = aspace.data.cpu().numpy() if hasattr(aspace.data, "cpu") else aspace.data
    return {"raw": save_dense_matrix(data, path, "%s-raw_input" % name_id, builder_config),
            "laplacian": save_sparse_matrix(gl.matrix, path, "%s-gl-matrix" % name_id, builder_config),
            "lambdas": save_lambda(aspace.lambdas(), path, "%s-lambdas" % name_id, builder_config)}
