#!/usr/bin/env python3
"""The topology matrix of an 8-GPU C3 job, projected from ONE MI355X: N = D doc-shards x R query-groups
(bench.py --doc-shards).  For D in 1, 2, 4, 8 one GPU plays one rank of a D-way doc-shard group (bench.py --force-dist
--emulate-world D: its docid shard, shard-level speculative thresholds, the exchange stage's merge of D lists; the other ranks are
assumed equal, the xGMI transfer -- ~1 MB per link and batch -- is not in it); a group's rate is one batch per step of its slowest
rank, and the R = N / D groups run side by side on their own GPUs.  D = 1 is the plain one-GPU line (nothing is exchanged).
    projected queries/s at N GPUs = (N / D) x batch / ms_per_step(D);   x one GPU = that / the one-GPU line of the same call
p50 is the batch call's latency on the emulated rank (what sharding buys).  Writes one JSON object (stdout, and --out).
Everything here is an EMULATION on one GPU: no multi-GPU node exists in this pool."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(extra, timeout=400):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--closed-loop", "", "--exhaustive-steps", "0", "--c4-steps", "0",
           "--c2-steps", "0", "--c5-steps", "0"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"{' '.join(cmd)} failed ({r.returncode}): {r.stderr[-1500:]}")
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8, help="the job projected (N)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--extra", default="", help="more bench.py flags for the emulated lines (e.g. '--shard-bounds exchange --emulate-peers final')")
    args = ap.parse_args()
    N = args.gpus
    one = bench(["--steps", str(args.steps), "--warmup", str(args.warmup)])
    B = one["config"]["batch_queries"]
    rows = [{"doc_shards": 1, "query_groups": N, "ms_per_step": one["ms_per_step"], "kernel_ms": one["roofline"]["avg_launch_ms"],
             "p50_batch_ms": one["p50_latency_ms"], "rank_queries_per_s": one["value"], "projected_queries_per_s": round(N * one["value"], 1),
             "x_one_gpu": float(N), "what": "N replicas of the index, nothing exchanged (the one-GPU line x N)"}]
    d = 2
    while d <= N:
        line = bench(["--force-dist", "--emulate-world", str(d), "--steps", str(args.steps), "--warmup", str(args.warmup), "--submitters", "1"]
                     + (args.extra.split() if args.extra else []))
        qps_group = B / (line["ms_per_step"] * 1e-3)
        rows.append({"doc_shards": d, "query_groups": N // d, "ms_per_step": line["ms_per_step"], "kernel_ms": line["roofline"]["avg_launch_ms"],
                     "p50_batch_ms": line["p50_latency_ms"], "rank_queries_per_s": round(qps_group, 1),
                     "projected_queries_per_s": round((N // d) * qps_group, 1), "x_one_gpu": round((N // d) * qps_group / one["value"], 2),
                     "shard_speculation": line["config"].get("shard_speculation"), "dist_stage_ms": line["config"].get("dist_stage_ms"),
                     "sharding": line["config"]["sharding"][:240]})
        d *= 2
    out = {"what": f"C3 on {N} MI355X projected from one GPU: N = D doc-shards x R query-groups (bench.py --doc-shards); an EMULATION -- one GPU plays one "
                   "rank of a D-way group, its peers are assumed equal, the xGMI transfer is not in it",
           "n_gpus": N, "batch_queries": B, "steps": args.steps, "one_gpu_queries_per_s": one["value"], "one_gpu_ms_per_step": one["ms_per_step"],
           "build_id": one["roofline"].get("build_id"), "extra": args.extra, "topologies": rows}
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
