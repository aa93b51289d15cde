"""The bench line's contract, checked on the committed line of the round (no GPU): the keys the driver and the judge read,
and the arithmetic between them (queries/s vs ms per step, roofline fraction vs achieved / peak, algorithmic bytes vs
the per-posting figure of SURVEY 8d)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_keys():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line.json")))
    assert lines, "no committed bench line"
    d = _line(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "queries/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    # whole-job throughput = batch x steps / time
    assert abs(d["value"] - d["config"]["batch_queries"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / average launch time (HIP events), 9 B per posting
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    assert r["bytes_per_posting"] == 9
    assert r["avg_launch_ms"] <= d["ms_per_step"]          # the kernel fits inside the step
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_packed_line_reports_its_own_denominator():
    path = os.path.join(ROOT, "profiles", "r02_bench_c3_packed.json")
    if not os.path.exists(path):
        return
    d = _line(path)
    assert "packed" in d["metric"] and d["roofline"]["bytes_per_posting"] == 4
