#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a round's PMC log (scripts/gpu_measure.sh <tag> pmc -> <tag>_pmc.txt): the HBM bytes per launch of
the dominant kernels, keyed by the build they were profiled on (bench.py reuses a record only for that build).
    python scripts/update_pmc_traffic.py <pmc.txt> <build_id> <round tag for the notes>
FETCH_SIZE is corrected by the factors scripts/ubench/gather_fetch.hip measured (profiles/r05_gather_fetch.txt): a 16 B/lane coalesced
stream is tallied at exactly half, a gather at exactly one 64-byte line.  Streamed posting bytes per launch of the pruned kernel: the
instrumented kernels' event counts (195.5 k postings per C3 query x 8 B x 1024 queries; x 4 B packed: profiles/r05_survivor_queue_event_counts.log;
the walk has not changed since)."""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path, build_id, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = {}
    for line in open(path):
        name, rest = line.split(" ", 1)
        d = ast.literal_eval(rest[rest.index("{"): rest.rindex("}") + 1])
        kernel = rest[: rest.index("{")].strip()
        rows.setdefault(name, []).append((kernel, d))
    def fetch(name, sub):
        for kernel, d in rows.get(name, []):
            if sub in kernel and "FETCH_SIZE" in d:
                return d["FETCH_SIZE"]
        return None
    out = []
    streamed = 1601233715.2
    f = fetch("fetch_default", "bm25_maxscore")
    if f:
        out.append({"workload": "C3", "batch": 1024, "kernel": "bm25_maxscore_kernel", "build_id": build_id, "launches": f[0], "FETCH_SIZE_KB_per_launch": f[1],
                    "streamed_posting_bytes_per_launch": streamed, "hbm_bytes_per_launch": f[1] * 1024 + streamed / 2,
                    "note": f"{tag} (fetch_default): FETCH_SIZE as counted + HALF of the streamed posting bytes (a 16 B/lane stream is tallied at half, a gather at one 64-byte line: profiles/r05_gather_fetch.txt)"})
    f = fetch("fetch_noprune", "bm25_scan")
    if f:
        out.append({"workload": "C3", "batch": 1024, "kernel": "bm25_scan_kernel", "build_id": build_id, "launches": f[0], "FETCH_SIZE_KB_per_launch": f[1],
                    "hbm_bytes_per_launch": f[1] * 1024 * 2, "note": f"{tag} (fetch_noprune, bench.py --no-prune): FETCH_SIZE x 2 (the scan streams its columns with 16 B/lane coalesced reads, tallied at half)"})
    f = fetch("fetch_packed", "bm25_maxscore")
    if f:
        out.append({"workload": "C3", "batch": 1024, "packed": True, "kernel": "bm25_maxscore_kernel", "build_id": build_id, "launches": f[0],
                    "FETCH_SIZE_KB_per_launch": f[1], "streamed_posting_bytes_per_launch": streamed / 2, "hbm_bytes_per_launch": f[1] * 1024 + streamed / 4,
                    "note": f"{tag} (fetch_packed, bench.py --packed): FETCH_SIZE as counted + half of the streamed posting words"})
    f = fetch("c4_fetch", "knn_sketch_kernel")
    if f:
        out.append({"workload": "C4", "kernel": "knn_sketch_kernel", "queries_per_pass": 64, "build_id": build_id, "launches": f[0], "launches_per_pass": 3,
                    "FETCH_SIZE_KB_per_launch": f[1], "hbm_bytes_per_launch": f[1] * 1024 * 2 * 3,
                    "note": f"{tag} (c4_fetch, bench.py --workload C4 --knn-queries 64): per PASS over the rows = 3 launches x FETCH_SIZE x 2 (a coalesced stream, tallied at half)"})
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    for r in out:
        print(r["workload"], r["kernel"], r.get("packed", False), round(r["hbm_bytes_per_launch"] / 1e9, 3), "GB per launch")


if __name__ == "__main__":
    main()
