#!/usr/bin/env python3
"""In which ORDER should the items of a MaxScore launch be dispatched?  (CPU, no GPU.)  The hardware hands workgroups to free CUs
in index order and an item occupies a CU (one workgroup per CU: the 160 KB of LDS), so a batch of 1024 C3 items on 256 CUs is a
list-scheduling problem; planner.cpp sorts the items longest-first by `postings + tiles x const` -- the EXHAUSTIVE scan's cost.
Input: per-query event counts of the walk model (scripts/cpu_maxscore_walk_sim.py, kernel order), whose groups and docs track the
measured kernel times: model cost of an item = 0.25 groups / mean + 0.75 docs / mean.
Output: how the plan-time keys correlate with that cost, and the makespan of 1024 items (drawn with replacement) on 256 CUs
dispatched in each key's order, against the balanced load.
    python scripts/cpu_launch_order_sim.py log [log ...]"""
import ast
import heapq
import re
import sys

import numpy as np


def read(paths):
    rows = []
    for p in paths:
        cur = None
        for line in open(p):
            m = re.match(r"q(\d+) df (\[.*?\]) P (\d+)(.*)", line)
            if not m:
                m2 = re.match(r"\s+kernel order\s+theta [\d.]+ (\{.*?\})", line)
                if m2 and cur is not None:
                    cur.update(ast.literal_eval(m2.group(1)))
                    rows.append(cur)
                    cur = None
                continue
            cur = dict(q=int(m.group(1)), df=ast.literal_eval(m.group(2)), P=int(m.group(3)))
            m3 = re.search(r"kernel order theta [\d.]+ (\{.*?\})", m.group(4))
            if m3:
                cur.update(ast.literal_eval(m3.group(1)))
                rows.append(cur)
                cur = None
    seen, out = set(), []
    for r in rows:
        if r["q"] not in seen:
            seen.add(r["q"])
            out.append(r)
    return out


def makespan(costs, order, cus=256):
    free = [0.0] * cus
    heapq.heapify(free)
    end = 0.0
    for i in order:
        t = heapq.heappop(free) + costs[i]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


def main():
    rows = read(sys.argv[1:])
    docs = np.array([r["docs"] for r in rows], float)
    groups = np.array([r["groups"] for r in rows], float)
    cost = 0.25 * groups / groups.mean() + 0.75 * docs / docs.mean()
    keys = {
        "all postings (planner.cpp today)": np.array([r["P"] for r in rows], float),
        "postings of the rarest clause": np.array([sum(r["df"][:1]) for r in rows], float),
        "postings of the 2 rarest clauses": np.array([sum(r["df"][:2]) for r in rows], float),
        "postings of the 3 rarest clauses": np.array([sum(r["df"][:3]) for r in rows], float),
        "2 rarest + 1/16 of the third": np.array([sum(r["df"][:2]) + r["df"][2] / 16 for r in rows], float),
    }
    print(f"{len(rows)} queries; model cost of an item: min {cost.min():.2f}, median {np.median(cost):.2f}, p90 {np.percentile(cost, 90):.2f}, "
          f"max {cost.max():.2f} of the mean")
    for name, v in keys.items():
        print(f"  correlation with the model cost: {np.corrcoef(v, cost)[0, 1]:+.2f}  {name}")
    rng = np.random.default_rng(7)
    res = {name: [] for name in list(keys) + ["batch order (no sort)", "the model cost itself (ideal key)"]}
    for trial in range(20):
        pick = rng.integers(0, len(rows), 1024)
        c = cost[pick]
        balanced = c.sum() / 256
        for name, v in keys.items():
            res[name].append(makespan(c, np.argsort(-v[pick], kind="stable")) / balanced)
        res["batch order (no sort)"].append(makespan(c, np.arange(1024)) / balanced)
        res["the model cost itself (ideal key)"].append(makespan(c, np.argsort(-c, kind="stable")) / balanced)
    print("makespan of 1024 items on 256 CUs / balanced load (20 draws: mean, min-max):")
    for name, v in res.items():
        print(f"  {np.mean(v):.3f}  ({min(v):.3f}-{max(v):.3f})  longest-first by {name}")
    # Cutting the heavy queries: the walk model prices a query cut into two items over halves of its doc range, each warming
    # up its own theta (the pessimistic end), at 0.53-0.58 of the whole per half (SPLIT_ITEMS=2 on the four heaviest queries).
    # Rule: the key (two heaviest clauses) predicts the cost by a straight line; a query predicted above T x the mean goes in
    # n = ceil(prediction / T) items of (1 / n) x 1.14 of its cost each; order: longest-first by the predicted item cost.
    key = keys["postings of the 2 rarest clauses"]
    A = np.stack([key, np.ones(len(key))], axis=1)
    coef, *_ = np.linalg.lstsq(A, cost, rcond=None)
    pred = A @ coef
    print(f"the key as a predictor: cost ~ {coef[0] * key.mean():.2f} x key / mean key + {coef[1]:.2f}; relative error of the prediction, p50 / p90: "
          f"{np.median(np.abs(pred - cost) / cost):.2f} / {np.percentile(np.abs(pred - cost) / cost, 90):.2f}")
    print("with heavy queries cut (balanced load = the UNCUT batch's: the cuts' overhead counts against them):")
    for T in (4.0, 3.0, 2.5, 2.0, 1.5):
        out, n_items = [], []
        rng = np.random.default_rng(7)
        for trial in range(20):
            pick = rng.integers(0, len(rows), 1024)
            c, pr = cost[pick], pred[pick]
            balanced = c.sum() / 256
            ic, ip = [], []
            for ci, pi in zip(c, pr):
                n = max(1, int(np.ceil(pi / T)))
                f = 1.0 if n == 1 else 1.14 / n
                ic += [ci * f] * n
                ip += [pi / n] * n
            ic, ip = np.array(ic), np.array(ip)
            out.append(makespan(ic, np.argsort(-ip, kind="stable")) / balanced)
            n_items.append(len(ic))
        print(f"  {np.mean(out):.3f}  ({min(out):.3f}-{max(out):.3f})  cut above {T} x the mean: {np.mean(n_items):.0f} items")


main()
