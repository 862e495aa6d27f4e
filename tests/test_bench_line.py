"""bench.py's stdout line must stay readable by the driver: one line, strict JSON, below 6 KB, the contract's fields present.

Round 5's line was 23 KB and BENCH_r05.json recorded "parsed": null.  The stub here IS that line (profiles/r05u_bench_1M_default.json,
the full record of a real run) plus hostile values (NaN, inf, long free text, numpy scalars)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_line  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full_record():
    with open(os.path.join(ROOT, "profiles", "r05u_bench_1M_default.json")) as f:
        text = f.read().strip().splitlines()[-1]
    return json.loads(text)


def test_round5_record_was_too_large_and_the_compact_line_is_not():
    full = _full_record()
    assert len(json.dumps(full)) > 20000          # what the driver could not parse
    text = bench_line.compact_line(full)
    assert "\n" not in text and len(text) < bench_line.LINE_LIMIT and len(text) < 4000
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == round(full["value"], -int(np.floor(np.log10(full["value"]))) + 6)
    assert abs(line["ms_per_step"] - full["ms_per_step"]) < 1e-3
    r = line["roofline"]
    assert set(r) >= {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"}
    assert len(r["kernel"]) <= 80 and r["kernel"].startswith("pair_join_f16_kernel")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    assert line["config"]["workload"].startswith("configs[2]") and "model" not in line["config"]
    for k in ("roofline_scan_frac", "roofline_search_frac", "roofline_spmm_frac"):
        assert 0.0 < line[k] < 1.0


def test_hostile_values_still_give_one_strict_json_line():
    full = _full_record()
    full["value"] = np.float64(full["value"])
    full["roofline"]["traffic"] = float("nan")
    full["roofline"]["kernel"] = "k" * 5000
    full["config"]["workload"] = "w" * 50000
    full["config"]["transport"] = "t" * 50000
    full["cpu_baseline"]["sample"] = "s" * 50000
    full["search_qps"] = float("inf")
    full["geometry_sweep"]["legs"] = [{"item_graph_ms": float("nan")}] * 2000
    full["sharded"] = {"x": "y" * 20000}
    text = bench_line.compact_line(full)
    assert len(text) < bench_line.LINE_LIMIT
    line = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # NaN / Infinity would raise
    assert line["roofline"]["traffic"] is None and line["search_qps"] is None
    for k in CONTRACT:
        assert k in line, k


def test_detail_file_holds_everything(tmp_path):
    full = _full_record()
    full["roofline"]["traffic"] = float("nan")
    written = bench_line.write_detail(full, str(tmp_path))
    assert written == [str(tmp_path / "bench_detail.json")]
    back = json.loads((tmp_path / "bench_detail.json").read_text())
    assert set(back) == set(full) and back["roofline"]["traffic"] is None
    assert back["geometry_sweep"] == bench_line.sanitize(full["geometry_sweep"])
