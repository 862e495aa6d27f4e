"""The ONE stdout line of bench.py, kept small enough for the driver to parse.

bench.py measures far more than the driver's contract asks for (every long kernel against its pipe, the other BASELINE
configs, a geometry sweep, the published-baseline legs ...).  Round 5 printed all of it on the one line: 23 KB, which the
driver could not read (BENCH_r05.json: "parsed": null).  Now the full record goes to `bench_detail.json` (next to bench.py,
and under gpurun_out/ when that directory exists so that it travels back from the GPU box) and stdout carries only
`compact_line(full)`: the contract's fields, `roofline`, `cpu_baseline` and a handful of scalars -- below LINE_LIMIT bytes
whatever the legs hold, strictly valid JSON (no NaN / Infinity).

No torch, no numpy: tests/test_bench_line.py exercises this on the CPU.
"""
import json
import math
import os

LINE_LIMIT = 6000      # the driver keeps an 8 081-character tail; the line has to fit with room to spare
KERNEL_CHARS = 80


def _num(v, digits=6):
    """A JSON-safe number rounded to `digits` significant digits (None for NaN / inf / non-numbers)."""
    if v is None or isinstance(v, bool):
        return v
    if isinstance(v, int):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(f):
        return None
    if f == 0.0:
        return 0.0
    return float("%.*g" % (digits, f))


def sanitize(o):
    """Deep copy with every non-finite float replaced by None and every key a string (json.dumps(allow_nan=False) then holds)."""
    if isinstance(o, dict):
        return {str(k): sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [sanitize(v) for v in o]
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, (str, int, bool)) or o is None:
        return o
    try:                       # numpy / torch scalars
        f = float(o)
        return f if math.isfinite(f) else None
    except (TypeError, ValueError):
        return repr(o)


def _short(s, limit):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= limit else s[: limit - 3].rstrip() + "..."


def _kernel_name(s):
    """'pair_join_f16_kernel (v_mfma ... long description)' -> the name alone, at most KERNEL_CHARS characters."""
    if s is None:
        return None
    s = str(s)
    cut = s.find(" (")
    if cut > 0:
        s = s[:cut]
    return _short(s, KERNEL_CHARS)


def _roofline(r):
    if not isinstance(r, dict) or "error" in r:
        return None
    out = {"kernel": _kernel_name(r.get("kernel")), "bound": r.get("bound"), "achieved": _num(r.get("achieved")),
           "peak": _num(r.get("peak")), "unit": r.get("unit"), "frac": _num(r.get("frac"), 4), "traffic": _num(r.get("traffic")),
           "avg_launch_ms": _num(r.get("avg_launch_ms", r.get("avg_step_ms")), 5)}
    if r.get("traffic_stale") is not None:
        out["traffic_stale"] = bool(r.get("traffic_stale"))
    return out


def _frac(full, key):
    r = full.get(key)
    return _num(r.get("frac"), 4) if isinstance(r, dict) and "frac" in r else None


def _leg(full, key, fields):
    r = full.get(key)
    if not isinstance(r, dict) or "error" in r:
        return None
    return {f: _num(r.get(f), 5) for f in fields if r.get(f) is not None}


def compact_line(full):
    """The driver's line: contract fields + roofline + cpu_baseline + scalar fractions; everything else stays in the detail file."""
    cfg = full.get("config") or {}
    line = {
        "metric": full.get("metric"), "value": _num(full.get("value"), 7), "unit": full.get("unit"),
        "n_gpus": full.get("n_gpus"), "steps": full.get("steps"), "warmup": full.get("warmup"),
        "ms_per_step": _num(full.get("ms_per_step"), 6), "higher_is_better": full.get("higher_is_better"),
        "scaling": full.get("scaling"), "vs_baseline": _num(full.get("vs_baseline")), "dtype": full.get("dtype"),
        "data": full.get("data"),
        "config": {"workload": _short(cfg.get("workload"), 300), "n_items": cfg.get("n_items"), "dim": cfg.get("dim"),
                   "n_queries": cfg.get("n_queries"), "sharding": cfg.get("sharding"), "transport": _short(cfg.get("transport"), 160)},
        "search_qps": _num(full.get("search_qps"), 6), "build_ms": _num(full.get("build_ms"), 5),
        "search_ms": _num(full.get("search_ms"), 5),
        "roofline": _roofline(full.get("roofline")),
    }
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample"), 320), "search_qps": _num(cb.get("search_qps"))}
    line["roofline_scan_frac"] = _frac(full, "roofline_scan")              # lambda-tau scan against HBM
    line["roofline_search_frac"] = _frac(full, "roofline_search")          # search threshold scan against the f16 MFMA peak
    line["roofline_search_hbm_frac"] = _frac(full, "roofline_search_hbm")  # small-batch search against HBM
    line["roofline_spmm_frac"] = _frac(full, "roofline_spmm")              # item-graph lambda against HBM
    dense = (full.get("value_all_bounds_off") or {}).get("roofline") if isinstance(full.get("value_all_bounds_off"), dict) else None
    line["roofline_dense_join_frac"] = _num(dense.get("frac"), 4) if isinstance(dense, dict) else None   # all-pairs f16 join, no bound on
    others = {}
    for key, fields in (("value_cfg1", ("build_ms", "search_ms")), ("value_cfg3_one_gpu", ("build_ms", "search_ms")),
                        ("value_wide", ("build_ms", "search_ms")), ("value_cfg3_rank_of_8", ("rank_ms", "one_gpu_ms", "one_gpu_over_rank", "received_bytes_from_peers")),
                        ("value_hard_data", ("build_ms",)), ("value_host_inputs", ("build_ms",)),
                        ("value_all_bounds_off", ("build_ms",))):
        leg = _leg(full, key, fields)
        if leg:
            others[key.replace("value_", "")] = leg
    sweep = full.get("geometry_sweep")
    if isinstance(sweep, dict) and sweep.get("legs"):
        others["geometry_sweep_ms"] = [_num(l.get("item_graph_ms"), 4) for l in sweep["legs"]]
    pub = full.get("published_equivalents")
    if isinstance(pub, dict):
        others["published_equivalents"] = {k: {f: _num(v.get(f), 5) for f in ("gpu_ms", "oracle_cpu_ms", "reference_published_ms") if f in v}
                                           for k, v in pub.items() if isinstance(v, dict)}
    if others:
        line["other_legs_ms"] = others
    for k in ("item_graph_nnz", "knn_uncertified_rows", "search_uncertified_rows", "commit"):
        if k in full:
            line[k] = full[k]
    if full.get("sharded"):
        line["sharded"] = sanitize(full["sharded"])
    line["detail"] = full.get("detail_file", "bench_detail.json")
    line = sanitize(line)
    text = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(", ", ": "))
    if len(text) >= LINE_LIMIT:
        # never happens with the fields above (free text is truncated); if it does, drop the optional parts rather than the contract
        for k in ("other_legs_ms", "sharded", "roofline_search_hbm_frac", "commit"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(", ", ": "))
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def write_detail(full, root):
    """The whole record, pretty enough to diff: bench_detail.json next to bench.py (+ gpurun_out/ when present)."""
    full = sanitize(full)
    text = json.dumps(full, allow_nan=False, indent=1, sort_keys=False)
    written = []
    for d in (root, os.path.join(root, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(text + "\n")
            written.append(os.path.join(d, "bench_detail.json"))
        except OSError:
            pass
    return written
