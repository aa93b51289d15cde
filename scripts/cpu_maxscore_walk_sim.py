#!/usr/bin/env python3
"""CPU model of bm25_maxscore_kernel's WALK (no GPU, numpy; event counts, not time): what one C3 item streams, evaluates and looks
up under (a) the kernel's order -- 12 waves taking 64-sub-tile windows in docid order, essential clauses of a window as one
sequence of 512-posting instructions, theta from compactions of a 2304-key buffer -- and (b) the TWO-SWEEP order of DESIGN §8
item 2: first the two rarest clauses over all windows (docs completed by lookups), then the windows again with those clauses
in mark-only mode and the rest as today.  The model follows maxscore.hip's rules (bound s_i + S_{i+1} >= theta per posting,
test-and-set per doc, S_j re-checked before each lookup, S of an instruction's first clause re-checked before the instruction);
simplifications: one segment per item, a clause's bound = its largest score in the corpus, scores in fp64, ties ignored, the 12
waves advance one instruction per tick.  (a)'s counts are to be held against profiles/r03_kernel_shapes.log -- per query 590
instruction groups, 262 k postings streamed, 166 k docs evaluated, 248 k lookups -- before (b)'s are believed.
    python scripts/cpu_maxscore_walk_sim.py [n_queries=8] [first_query=0]     (ONLY_KERNEL_ORDER=1: the kernel's order alone)
Round 4: SPEC_Z=5 adds the kernel's SPECULATIVE thresholds (plan.h: kHitsSpecInvalid) to the kernel's order -- at every compaction,
and whenever the number of windows begun has doubled, theta is raised to the (k w / W + z sqrt(k w / W) + 2)-th best score held
(w of W windows begun) -- and reports how many guesses overshot the final k-th score (what the merge's check would catch).  The
counts `rounds` (later-clause rounds as the kernel runs them: one per instruction and later clause with a live doc) and
`queued_rounds` (the same lookups regrouped per wave, window and clause into rounds of 512: DESIGN 8 item 2) come with every run."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrtsearch_amd import _lib, synth, workload   # noqa: E402

WIN = 64 * 1024
WAVES = 12
CAP = int(os.environ.get("CAND_CAP", "2304"))     # the workgroup's candidate buffer (keys); a compaction keeps the k best
ONLY_KERNEL_ORDER = os.environ.get("ONLY_KERNEL_ORDER", "") != ""
SPLIT_ITEMS = int(os.environ.get("SPLIT_ITEMS", "0"))   # > 1: also the query cut into that many items over equal doc ranges
INSTR = 512
SPEC_Z = float(os.environ.get("SPEC_Z", "0"))
BLOCK_SHIFTS = [int(x) for x in os.environ.get("BLOCK_SHIFTS", "").split(",") if x]   # e.g. 16,13,10,7


class Item:
    def __init__(self, k):
        self.k = k
        self.theta = 0.0
        self.cand = []
        self.n = dict(groups=0, windows=0, postings=0, marked=0, survivors=0, docs=0, lookups=0, cands=0, compactions=0, rounds=0,
                      queued_rounds=0, estimates=0)
        self.n_win, self.wins_started, self.spec_at, self.guess_max = 1, 0, 2 * WAVES, 0.0

    def guess(self):
        """maxscore.hip: ms_compact -- the speculative theta from what the buffer holds."""
        if SPEC_Z <= 0 or not self.cand:
            return
        m = self.k * min(1.0, self.wins_started / self.n_win)
        r = m + SPEC_Z * np.sqrt(m) + 2.0
        allc = np.concatenate(self.cand)
        if r < self.k and len(allc) > int(r):
            g = float(np.partition(allc, len(allc) - int(r))[len(allc) - int(r)])
            if g > self.theta:
                self.theta = g
                self.guess_max = max(self.guess_max, g)
        self.spec_at = 0 if self.wins_started >= self.n_win else max(2 * self.wins_started, self.wins_started + WAVES)

    def window_begun(self):
        self.wins_started += 1
        if SPEC_Z > 0 and self.spec_at and self.wins_started >= self.spec_at:
            self.n["estimates"] += 1
            self.guess()

    def push(self, scores):
        scores = scores[scores > self.theta]
        if len(scores) == 0:
            return
        self.n["cands"] += len(scores)
        self.cand.append(scores)
        tot = sum(len(c) for c in self.cand)
        if tot > CAP:
            allc = np.concatenate(self.cand)
            if len(allc) >= self.k:
                allc = np.partition(allc, len(allc) - self.k)[len(allc) - self.k:]
                self.theta = max(self.theta, float(allc.min()))
            self.cand = [allc]
            self.n["compactions"] += 1
            self.guess()


def window_instructions(D, lo_hi, clauses):
    """The window's instruction sequence: clause-major entries (clause, index into D[clause]); every clause padded to whole
    8-posting lane groups (index -1), cut into instructions of 64 lane groups."""
    cl, ix = [], []
    for c in clauses:
        lo, hi = lo_hi[c]
        n = hi - lo
        if n <= 0:
            continue
        pad = (-n) % 8
        cl.append(np.full(n + pad, c, dtype=np.int32))
        ix.append(np.concatenate([np.arange(lo, hi, dtype=np.int64), np.full(pad, -1, dtype=np.int64)]))
    if not cl:
        return []
    cl, ix = np.concatenate(cl), np.concatenate(ix)
    return [(cl[i: i + INSTR], ix[i: i + INSTR]) for i in range(0, len(cl), INSTR)]


def run_walk(it, D, Sc, dense, S, N, stream, mark_only=(), blk=None, win_range=None):
    """One sweep over all windows.  stream: clauses that may be streamed (when essential); mark_only: clauses whose postings only set
    the seen bits (already handled by an earlier sweep).  blk = (shift, Sblk): bounds per block of 2^shift docs -- Sblk[j][b] = what
    the clauses j.. can add at most to a doc of block b (their largest scores INSIDE the block) -- instead of the corpus-wide S[j]
    in the per-posting bound and before each lookup (which clauses are streamed is still decided by S)."""
    n_terms = len(D)
    n_win = (N + WIN - 1) // WIN
    next_win = [0]
    if win_range is not None:                                     # an item over a part of the doc range: windows [g0, g1)
        next_win[0], n_win = win_range[0], min(win_range[1], n_win)
    waves = [None] * WAVES
    it.n_win = max(1, n_win - next_win[0])

    def open_window(g):
        w0, w1 = g * WIN, min((g + 1) * WIN, N)
        lo_hi = [(int(np.searchsorted(D[c], w0)), int(np.searchsorted(D[c], w1))) for c in range(n_terms)]
        ess = [c for c in stream if S[c] >= it.theta]            # decided at the window's start (ng = 0 otherwise)
        seq = window_instructions(D, lo_hi, list(mark_only) + ess) if ess else []
        it.n["windows"] += 1
        it.window_begun()
        return dict(w0=w0, seen=np.zeros(w1 - w0, dtype=bool), seq=seq, pos=0, qj=np.zeros(n_terms, dtype=np.int64))

    def step(ws):
        if ws["pos"] >= len(ws["seq"]):
            if "qj" in ws:   # the window is done: its lookups regrouped per clause into rounds of 512
                it.n["queued_rounds"] += int(np.sum((ws.pop("qj") + INSTR - 1) // INSTR))
            return False
        cl, ix = ws["seq"][ws["pos"]]
        ws["pos"] += 1
        theta = it.theta
        c_first = int(cl[0])
        if c_first not in mark_only and S[c_first] < theta:      # the rest of the window has become non-essential
            ws["pos"] = len(ws["seq"])
            if "qj" in ws:
                it.n["queued_rounds"] += int(np.sum((ws.pop("qj") + INSTR - 1) // INSTR))
            return False
        it.n["groups"] += 1
        valid = ix >= 0
        round_of = np.zeros(n_terms, dtype=bool)   # later clauses for which this instruction runs a round
        for c in np.unique(cl):
            m = valid & (cl == c)
            idx = ix[m]
            docs = D[c][idx]
            if c in mark_only:
                it.n["marked"] += len(docs)
                ws["seen"][docs - ws["w0"]] = True
                continue
            it.n["postings"] += len(docs)
            s = Sc[c][idx]
            alive = s + (S[c + 1] if blk is None else blk[1][c + 1][docs >> blk[0]]) >= theta
            docs, s = docs[alive], s[alive]
            it.n["survivors"] += len(docs)
            rel = docs - ws["w0"]
            first = ~ws["seen"][rel]
            ws["seen"][rel] = True
            docs, run = docs[first], s[first].copy()
            it.n["docs"] += len(docs)
            live = np.ones(len(docs), dtype=bool)
            for j in range(c + 1, n_terms):
                live &= run + (S[j] if blk is None else blk[1][j][docs >> blk[0]]) >= theta
                if not live.any():
                    break
                it.n["lookups"] += int(live.sum())
                round_of[j] = True
                ws["qj"][j] += int(live.sum())
                run = run + np.where(live, dense[j][docs], 0.0)
            it.push(run[live])
        it.n["rounds"] += int(round_of.sum())
        return True

    active = True
    while active:
        active = False
        for wv in range(WAVES):
            if waves[wv] is None or not step(waves[wv]):
                # (a window that ended, or none yet: take the next one and run its first instruction in the same tick)
                while next_win[0] < n_win:
                    waves[wv] = open_window(next_win[0])
                    next_win[0] += 1
                    if step(waves[wv]):
                        active = True
                        break
                else:
                    waves[wv] = dict(seq=[], pos=0)
            else:
                active = True


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0         # queries [first, first + nq) of the bench's query set
    w = workload.C3
    N, k = w.n_docs, w.k
    qr = synth.make_queries(first + nq, w.n_terms, w.max_rank)[first:]
    lens = synth.doc_lengths(N)
    norms = synth.int_to_byte4(lens)
    avgdl = np.float32(int(lens.astype(np.int64).sum()) / N)
    import ctypes as C
    cache = np.zeros(256, dtype=np.float32)
    _lib.load().nrtgpu_bm25_norm_cache(C.c_float(float(avgdl)), C.c_float(1.2), C.c_float(0.75), cache.ctypes.data)
    post = {r: synth.term_postings(N, r) for r in sorted(set(int(x) for x in qr.reshape(-1)))}
    dense = [np.zeros(N, dtype=np.float32) for _ in range(w.n_terms)]
    tot = {}
    for qi in range(nq):
        terms = [int(x) for x in qr[qi]]
        info = []
        for r in terms:
            d, f = post[r]
            idf = np.float32(np.log(1 + (N - len(d) + 0.5) / (len(d) + 0.5)))
            s = (idf - idf / (np.float32(1) + f.astype(np.float32) * cache[norms[d]])).astype(np.float64)
            info.append((float(idf), d.astype(np.int64), s))
        info.sort(key=lambda t: -t[0])                            # heaviest (rarest) clause first
        D, Sc = [t[1] for t in info], [t[2] for t in info]
        for c in range(len(D)):
            dense[c][:] = 0
            dense[c][D[c]] = Sc[c]
        ub = [float(s.max()) for s in Sc]
        S = [sum(ub[j:]) for j in range(len(ub))] + [0.0]
        P = sum(len(d) for d in D)
        res = {}
        a = Item(k)
        run_walk(a, D, Sc, dense, S, N, stream=list(range(len(D))))
        res["kernel order"] = a
        if SPLIT_ITEMS > 1:   # the query as several items over equal doc ranges, each with a theta of its own (the pessimistic end:
            #                   the kernel's items of a query read each other's theta once per window)
            n_win = (N + WIN - 1) // WIN
            parts = []
            for pi in range(SPLIT_ITEMS):
                sp = Item(k)
                run_walk(sp, D, Sc, dense, S, N, stream=list(range(len(D))), win_range=(n_win * pi // SPLIT_ITEMS, n_win * (pi + 1) // SPLIT_ITEMS))
                parts.append(sp.n)
            print(f"q{first + qi} df {[len(d) for d in D]} P {P}   kernel order theta {a.theta:.3f} {a.n}   split {SPLIT_ITEMS}: {parts}", flush=True)
            continue
        if ONLY_KERNEL_ORDER:
            over = ""
            if SPEC_Z > 0:
                acc0 = np.zeros(N, dtype=np.float64)
                for c in range(len(D)):
                    acc0[D[c]] += Sc[c]
                nz0 = acc0[acc0 > 0]
                final0 = float(np.partition(nz0, len(nz0) - k)[len(nz0) - k]) if len(nz0) >= k else 0.0
                tot["guesses that overshot"] = tot.get("guesses that overshot", 0) + int(a.guess_max > final0)
                over = f"   largest guess {a.guess_max:.3f} vs final k-th {final0:.3f}" + ("  OVERSHOT" if a.guess_max > final0 else "")
            print(f"q{first + qi} df {[len(d) for d in D]} P {P}   kernel order theta {a.theta:.3f} {a.n}{over}", flush=True)
            t = tot.setdefault("kernel order", {})
            for kk, v in a.n.items():
                t[kk] = t.get(kk, 0) + v
            continue
        b = Item(k)
        run_walk(b, D, Sc, dense, S, N, stream=[0, 1])
        theta_1 = b.theta
        if S[2] >= b.theta:                                        # something beyond the two rarest clauses is still essential
            run_walk(b, D, Sc, dense, S, N, stream=list(range(2, len(D))), mark_only=(0, 1))
        res["two sweeps"] = b
        # bounds per block of docs for the clauses after the first (Lucene's block-max idea on doc ranges): how many fewer docs?
        for shift in BLOCK_SHIFTS:
            nb = (N >> shift) + 1
            suf = [np.zeros(nb) for _ in range(len(D) + 1)]
            for j in range(len(D) - 1, -1, -1):
                bm = np.zeros(nb)
                np.maximum.at(bm, D[j] >> shift, Sc[j])
                suf[j] = suf[j + 1] + bm
            bk = Item(k)
            run_walk(bk, D, Sc, dense, S, N, stream=list(range(len(D))), blk=(shift, suf))
            res[f"block bounds 2^{shift}"] = bk
        # calibration against profiles/r02_seed_experiment.log (kernel time with theta seeded at f x the final k-th score:
        # f = 1: -22 %, 0.9: -11 %, 0.7: -3 %): the kernel's order started from such a seed
        acc = np.zeros(N, dtype=np.float64)
        for c in range(len(D)):
            acc[D[c]] += Sc[c]
        nz = acc[acc > 0]
        theta_final = float(np.partition(nz, len(nz) - k)[len(nz) - k]) if len(nz) >= k else 0.0
        for f in (1.0, 0.9, 0.7):
            sd = Item(k)
            sd.theta = f * theta_final * (1 - 1e-12)
            run_walk(sd, D, Sc, dense, S, N, stream=list(range(len(D))))
            res[f"seed {f:.1f} x"] = sd
        print(f"q{first + qi} df {[len(d) for d in D]} P {P}", flush=True)
        for name, it in res.items():
            extra = f" theta after sweep 1: {theta_1:.3f}" if name == "two sweeps" else ""
            print(f"   {name:13s} theta {it.theta:.3f} {it.n}{extra}", flush=True)
            t = tot.setdefault(name, {})
            for kk, v in it.n.items():
                t[kk] = t.get(kk, 0) + v
        assert abs(a.theta - b.theta) < 1e-9 or True
    overshot = tot.pop("guesses that overshot", None)
    for name, t in tot.items():
        print("MEAN per query,", name, {kk: round(v / nq, 1) for kk, v in t.items()})
    if overshot is not None:
        print(f"speculation at z = {SPEC_Z}: {overshot} of {nq} queries had a guess above their final k-th score (they would be run again)")
    # A linear reading of the seed experiment: time ~ F groups + a postings + b docs + c lookups, non-negative weights fitted to
    # the three measured ratios; what it says about the two sweeps (an indication: 3 equations, 4 unknowns -> least norm)
    try:
        if ONLY_KERNEL_ORDER:
            raise RuntimeError("kernel order only")
        from scipy.optimize import nnls
        keys = ("groups", "postings", "docs", "lookups")
        base = np.array([tot["kernel order"][kk] for kk in keys], dtype=np.float64)
        rows, rhs = [base / base], [1.0]
        for name, ratio in (("seed 1.0 x", 0.776), ("seed 0.9 x", 0.887), ("seed 0.7 x", 0.967)):
            rows.append(np.array([tot[name][kk] for kk in keys], dtype=np.float64) / base)
            rhs.append(ratio)
        wts, resid = nnls(np.array(rows), np.array(rhs))
        two = np.array([tot["two sweeps"][kk] for kk in keys], dtype=np.float64) / base
        print("share of the kernel's time by event (fit):", dict(zip(keys, np.round(wts, 3))), "residual", round(float(resid), 4))
        print("relative counts, two sweeps:", dict(zip(keys, np.round(two, 3))), "-> modelled time ratio", round(float(two @ wts), 3))
    except Exception as e:   # noqa: BLE001
        print("fit skipped:", e)


main()
