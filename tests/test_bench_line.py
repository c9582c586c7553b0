"""bench.py's final stdout line (CPU test, no GPU): the compaction that cuts the complete result object down to the contract's
fields must stay under 6,000 bytes and keep `roofline`, `cpu_baseline`, parity and `survey_8d` -- checked on the complete object
of a real run (profiles/r05_bench.json: round 5's 21.5 KB line, the one the driver could not parse) and on a worst case with
eight ranks, long strings and every optional leg present."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        return json.load(f)


def test_compact_line_of_a_real_run_is_short_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 20000                      # the object that broke the record
    s = bench.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_legs.json"))
    assert len(s.encode()) < bench.LINE_LIMIT == 6000 and "\n" not in s
    d = json.loads(s)
    for k in CONTRACT + ("parity_vs_oracle", "survey_8d", "config3", "config5"):
        assert k in d, k
    assert d["metric"] == full["metric"] and abs(d["value"] - full["value"]) < 1e-5 * full["value"]
    for name in (None, "config3", "config5"):
        o, f = (d, full) if name is None else (d[name], full[name])
        r = o["roofline"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] == "hbm"
        assert abs(r["frac"] - f["roofline"]["frac"]) < 1e-5 and r["traffic"] == f["roofline"]["traffic"]
        c = o["cpu_baseline"]
        assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["coarse_iterations"] == 3
    assert d["legs_file"] == os.path.join("gpurun_out", "bench_legs.json")
    assert d["legs"]["plane_normals"]["value"] > 0 and d["legs"]["voxel_icp"]["dep1_to_dep2"]["cpu_value"] > 0


def test_compact_line_never_exceeds_the_limit():
    full = _full()
    full["n_gpus"] = 8
    full["per_rank"] = full["per_rank"] * 8
    full["config"]["workload"] = full["config"]["workload"] * 6
    full["config"]["parallelism"] = "x" * 4000
    full["pose_exchange"] = "y" * 3000
    full["cpu_baseline"]["sample"] = "z" * 5000
    full["config"]["n_src"] = list(range(4)); full["config"]["n_tgt"] = list(range(4))
    for k in list(bench._LEG_NAMES):
        full[k + "_again"] = full.get(k)
    s = bench.compact_line(full, "/tmp/legs.json")
    assert len(s.encode()) < 6000
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k
    # a pathological object (every optional part huge) still fits: parts are dropped, the contract fields never
    full["survey_8d"] = {f"k{i}": 1.0 / 3 for i in range(400)}
    full["config3"]["roofline"]["kernel"] = "k" * 3000
    s = bench.compact_line(full, None)
    assert len(s.encode()) < 6000 and all(k in json.loads(s) for k in CONTRACT)
