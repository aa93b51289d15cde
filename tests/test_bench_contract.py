"""The bench line's contract, checked on CODE (no GPU): bench.roofline_record builds the `roofline` object from measured
numbers -- its arithmetic (fraction = achieved / peak; algorithmic bytes = the per-posting figure of SURVEY 8d; a pruned
kernel's algorithmic rate is reported as an EFFECTIVE figure and its fraction is the physical one) is asserted on synthetic
inputs.  The committed line of the round is checked for the keys the driver and the judge read."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_roofline_record_exhaustive_scan_is_an_algorithmic_bandwidth_fraction():
    # 4.05 G postings x 9 B in 10.1 ms
    r = bench.roofline_record("bm25_scan_kernel", False, 10.1, 4.05e9 * 9, 9, 8, 31.4e9, "x2 wide reads")
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["effective"] is False
    assert abs(r["achieved"] - 4.05e9 * 9 / 10.1e-3 / 1e9) < 0.1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["effective_frac"] is None and r["effective_achieved"] is None
    assert abs(r["frac_at_8B_per_posting"] - r["frac"] * 8 / 9) < 1e-3
    assert abs(r["physical_achieved"] - 31.4e9 / 10.1e-3 / 1e9) < 0.1 and r["traffic"] == 31.4e9
    assert "static profile" in r["traffic_source"]


def test_roofline_record_pruned_kernel_reports_the_physical_fraction():
    r = bench.roofline_record("bm25_maxscore_kernel", True, 2.9, 4.05e9 * 9, 9, 8, 7.0e9, "counted + streamed bytes")
    assert r["effective"] is True
    # frac / achieved: what the kernel physically fetched; the effective (algorithmic) rate beside it may exceed the peak
    assert abs(r["achieved"] - 7.0e9 / 2.9e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4 and r["frac"] < 1.0
    assert abs(r["effective_achieved"] - 4.05e9 * 9 / 2.9e-3 / 1e9) < 0.1 and r["effective_frac"] > 1.0
    assert r["achieved_is"].startswith("physical")
    # without a PMC profile there is no physical figure: the algorithmic one stays, flagged effective
    r2 = bench.roofline_record("bm25_maxscore_kernel", True, 2.9, 4.05e9 * 9, 9, 8, None)
    assert r2["traffic"] is None and r2["physical_frac"] is None and r2["traffic_source"] is None
    assert abs(r2["frac"] - r2["effective_frac"]) < 1e-9 and r2["achieved_is"].startswith("algorithmic")


def test_usable_cpus_is_positive():
    assert bench.usable_cpus() >= 1


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
    assert r["bytes_per_posting"] == 9
    assert r["avg_launch_ms"] <= d["ms_per_step"]          # the kernel fits inside the step
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_committed_c4_line_keeps_the_contract_and_its_roofline_arithmetic():
    """The exact-kNN line (bench.py --workload C4): the contract keys; throughput = queries per step / step time; the roofline
    fraction is the PHYSICAL one (the fp16 sketch's bytes the nomination kernel streams), the algorithmic fp32 bytes of SURVEY 8d
    stand beside it as effective_*; the answer was verified against fp64 over every row and against the CPU port."""
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_c4_q32.json")))
    assert lines
    d = _line(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    c, r = d["config"], d["roofline"]
    assert d["dtype"] == "f32" and d["unit"] == "queries/s" and "workload" in c and "model" not in c
    assert abs(d["value"] - c["queries_per_step"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if r["kernel"] == "knn_sketch_kernel":
        rows, dim = c["rows_per_gpu"], c["dim"]
        steps16 = ((dim + 31) // 32 + 3) // 4 * 4
        assert r["effective"] is True and r["streamed_bytes_per_launch"] == rows * steps16 * 64
        assert r["algorithmic_bytes_per_launch"] == rows * dim * 4
        assert abs(r["achieved"] - r["streamed_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
        assert abs(r["effective_achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1.0
        assert r["frac"] < 1.0 < r["effective_frac"]
    assert d["verify"]["agrees_with_fp64"] is True and d["verify"]["rows"] == c["rows_per_gpu"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["agrees_with_device"] is True and cb["value"] > 0
    cl = d.get("closed_loop")
    if cl:
        assert "nrtgpu_knn_exact_coalesced" in cl["entry"] and cl["64"]["qps"] > 10 * cl["1"]["qps"] / 2


def test_knn_roofline_record_arithmetic():
    """bench.knn_roofline_record on synthetic numbers: from the fp16 sketch the fraction is physical (2 bytes per element, steps of 32
    dimensions padded to groups of four) with the fp32 bytes beside it as effective; from the fp32 rows they coincide and the matrix
    fraction is reported."""
    r = bench.knn_roofline_record(10_000_000, 768, 32, 2.7, True, 3.0, 0)
    assert r["kernel"] == "knn_sketch_kernel" and r["effective"] is True
    assert r["streamed_bytes_per_launch"] == 10_000_000 * 24 * 64 == 10_000_000 * 768 * 2
    assert r["algorithmic_bytes_per_launch"] == 10_000_000 * 768 * 4
    assert abs(r["achieved"] - 15.36e9 / 2.7e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert abs(r["effective_frac"] - 2 * r["frac"]) < 1e-3 and r["second_passes"] == 0
    # the sketch path's matrix-core figures (round 5): useful fp16 multiply-adds against the dense fp16 peak
    assert r["mfma_dtype"] == "f16" and r["mfma_peak_tflops"] == 2500.0
    assert abs(r["mfma_tflops"] - 2 * 10_000_000 * 768 * 32 / 2.7e-3 / 1e12) < 0.01 and abs(r["mfma_frac"] - r["mfma_tflops"] / 2500.0) < 1e-4
    assert r["traffic"] is None
    # 96 dimensions: 3 steps of 32, padded to 4 -> 256 bytes per row streamed, 384 algorithmic
    r = bench.knn_roofline_record(1000, 96, 1, 1.0, True, 3.0, 1)
    assert r["streamed_bytes_per_launch"] == 1000 * 256 and r["algorithmic_bytes_per_launch"] == 1000 * 384 and r["second_passes"] == 1
    r = bench.knn_roofline_record(10_000_000, 768, 32, 6.0, False, 6.0, 0)
    assert r["kernel"] == "knn_score_kernel" and r["effective"] is False and r["effective_frac"] is None
    assert r["streamed_bytes_per_launch"] == r["algorithmic_bytes_per_launch"] == 10_000_000 * 768 * 4
    assert abs(r["mfma_tflops"] - 2 * 10_000_000 * 768 * 32 / 6e-3 / 1e12) < 0.01
    assert abs(r["mfma_frac"] - r["mfma_tflops"] / 157.3) < 1e-3 and r["mfma_dtype"] == "f32"
