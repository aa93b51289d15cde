"""The rule behind the MaxScore route's speculative thresholds (plan.h: kHitsSpecInvalid; maxscore.hip: ms_compact), in isolation
and without a GPU: a collector that has seen a fraction g of the docs -- spread like a random sample -- holds about m = k g of
the final top-k, so its (m + z sqrt(m) + 2)-th best score lies below the final k-th best, z standard deviations deep.  The
kernel's guesses are CHECKED (the merge compares the merged list's k-th key with the largest guess) and failed queries are run
again, so this rule decides how often a second pass is paid -- never what is returned.  Here: the failure rate of the rule
itself on random scores, by margin; and that the check the merge makes is exactly "did the guess overshoot"."""
import numpy as np


def guess_rank(k: int, g: float, z: float) -> int:
    """maxscore.hip: ms_compact -- 0: no guess (the rank would not be below k)."""
    m = k * min(1.0, g)
    r = m + z * np.sqrt(m) + 2.0
    return int(r) if r < k else 0


def trial(rng, n_docs, k, g, z):
    scores = rng.lognormal(size=n_docs).astype(np.float32)
    seen = scores[: int(n_docs * g)]                    # (the docs arrive in random order: a prefix is a sample)
    r = guess_rank(k, g, z)
    if r == 0 or r > len(seen):
        return None
    guess = np.partition(seen, len(seen) - r)[len(seen) - r]          # the r-th best of what was seen
    kth = np.partition(scores, n_docs - k)[n_docs - k]                # the final k-th best
    return bool(guess > kth)                                          # overshoot: something competitive would be skipped


def test_the_rank_is_deeper_than_the_expected_share_and_never_reaches_k():
    for k in (10, 100, 1000):
        for g in (0.001, 0.01, 0.1, 0.5, 0.9):
            for z in (0.0, 3.0, 5.0):
                r = guess_rank(k, g, z)
                assert r == 0 or (k * g < r < k)
    assert guess_rank(1000, 1.0, 5.0) == 0 and guess_rank(1000, 0.99, 5.0) == 0      # nothing left to guess about
    assert guess_rank(1000, 12 / 159, 5.0) == 120                                       # C3: every wave has begun one window of 159


def test_failure_rate_by_margin():
    rng = np.random.Generator(np.random.PCG64(7))
    n_docs, k = 60_000, 1000
    fails = {0.0: 0, 3.0: 0, 5.0: 0}
    trials = 0
    for g in (0.02, 0.075, 0.15, 0.3, 0.6):
        for _ in range(60):
            trials += 1
            for z in fails:
                out = trial(rng, n_docs, k, g, z)
                fails[z] += int(bool(out))
    assert fails[5.0] == 0, fails                      # five standard deviations: not in 300 trials (nominally 3e-7 each)
    assert fails[3.0] <= 3, fails                      # three: the nominal 0.13 % (108 of 122 880 queries on the GPU: profiles/r04_speculation_ab.log)
    assert fails[0.0] > trials // 5, fails             # no margin: a coin flip -- the margin is what makes the guess usable


def test_what_the_merge_checks_is_the_overshoot():
    """merge_topk_kernel: the merged list stands iff its k-th key reaches the largest guess.  With every doc scoring >= the guess
    kept and everything below it possibly skipped, that is the case exactly when the guess did not overshoot the true k-th best."""
    rng = np.random.Generator(np.random.PCG64(11))
    for _ in range(200):
        n, k = 5000, 50
        scores = rng.random(n).astype(np.float32)
        true_kth = np.sort(scores)[::-1][k - 1]
        guess = np.float32(rng.choice([true_kth * 0.9, true_kth, np.nextafter(true_kth, np.float32(2.0)), true_kth * 1.05]))
        kept = np.sort(scores[scores >= guess])[::-1]          # what a walk that skips everything below the guess can still deliver
        stands = len(kept) >= k and kept[k - 1] >= guess        # the merge's check
        assert stands == bool(guess <= true_kth)
        if stands:
            assert kept[:k].tolist() == np.sort(scores)[::-1][:k].tolist()     # ... and then the list IS the exact top-k


def shard_trial(rng, n_docs, k, world, f, z, sizes=None):
    """One search shared by `world` shards (search.cpp: enqueue_search, spec_world): every shard guesses the k-th score of the WHOLE
    search from the fraction f of ITS docs it has seen -- the windows of all shards in the denominator, g = f / world -- and the
    largest guess is checked against the merged list.  -> did any shard's guess overshoot the global k-th best?"""
    scores = rng.lognormal(size=n_docs).astype(np.float32)
    kth = np.partition(scores, n_docs - k)[n_docs - k]
    bounds = np.linspace(0, n_docs, world + 1).astype(int) if sizes is None else np.concatenate([[0], np.cumsum(sizes)])
    worst = 0.0
    for s in range(world):
        mine = scores[bounds[s]: bounds[s + 1]]
        seen = mine[: int(len(mine) * f)]
        r = guess_rank(k, f / world, z)
        if r == 0 or r > len(seen):
            continue
        worst = max(worst, float(np.partition(seen, len(seen) - r)[len(seen) - r]))
    return worst > kth


def test_shards_guessing_the_global_threshold():
    """a shard's docs are a 1 / W sample of the index.  Equal shards, five standard deviations: at most one overshoot in 8 x 240
    shard-guesses; without the margin most searches have a shard that overshoots (which is why the margin is there); and a shard
    that holds more of the index than it says -- spec_world counts shards OF THIS SHARD'S SIZE (include/nrtgpu.h) -- guesses too
    high: what the check against the merged list and the re-run are for."""
    rng = np.random.Generator(np.random.PCG64(23))
    n_docs, k, world = 160_000, 1000, 8
    over = {0.0: 0, 5.0: 0}
    trials = 0
    for f in (0.1, 0.25, 0.5, 1.0):
        for _ in range(60):
            trials += 1
            for z in over:
                over[z] += int(shard_trial(rng, n_docs, k, world, f, z))
    # (with this seed ONE of the 1920 shard-guesses overshoots: a shard that had seen 61 of the top 1000 where 31 were expected,
    #  5.3 standard deviations -- a count's tail is heavier than a normal's; 60 000 further guesses, other seeds: none.  The check
    #  against the merged list is what makes such a guess cost a re-run instead of an answer.)
    assert over[5.0] <= 1, over
    assert over[0.0] > trials // 2, over
    # the guess at the end of a shard's walk: rank k / W + z sqrt(k / W) + 2 of its own docs (183 of 1000 at W = 8, z = 5)
    assert guess_rank(1000, 1.0 / 8, 5.0) == 182 and guess_rank(1000, 1.0 / 2, 5.0) == 613
    # unequal shards under the equal-shards assumption: the first shard holds half of the index and says "one of eight"
    sizes = [n_docs // 2] + [n_docs // 14] * 7
    sizes[-1] += n_docs - sum(sizes)
    lopsided = sum(int(shard_trial(rng, n_docs, k, world, 1.0, 5.0, sizes)) for _ in range(40))
    assert lopsided >= 36, lopsided      # it holds ~500 of the top 1000 and guesses its 182nd best: caught by the check, paid as a re-run
