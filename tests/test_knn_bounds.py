"""Soundness of the rounding bounds the exact vector search certifies its answers with (nrtsearch_amd/csrc/vectors.cpp:
bound_of / bound16_of, plan.h: knn_result_upper / knn_estimate_lower), checked on the CPU: the two kinds of ESTIMATE the matrix-core
passes produce are emulated in numpy -- an fp32 dot product summed in another order than the oracle's, and the fp16 sketch (rows and
query scaled by a power of two, rounded to fp16, products exact, fp32 accumulation) -- and |estimate - result| must stay inside the
bound for every row, where the result is the oracle's scalar left-to-right fp32 sum.  The formulas are restated here from the C++;
what the test pins is the mathematics (worst-case data included), the GPU tests pin the code."""
import numpy as np
import pytest

U = 2.0 ** -24


def gamma(dim):
    return (dim + 4) * U


def seq_dot(oracle, q, v):
    """The oracle's order of summation (nrt_oracle_vector_score: scalar, left to right, every product and sum rounded to fp32),
    restated: the oracle exports scores, and a score map would hide a small dot product behind its "+ 1"."""
    acc = np.float32(0.0)
    p = (q * v).astype(np.float32)      # each product rounded to fp32 (elementwise: no fused multiply-add)
    for x in p:
        acc = np.float32(acc + x)
    return float(acc)


def pow2_scale(x_absmax):
    if not (x_absmax > 0):
        return 1.0
    _, e = np.frexp(np.float32(x_absmax))     # x < 2^e
    return float(np.ldexp(1.0, 14 - int(e)))


def sketch_dots(q, rows):
    """knn_sketch_kernel's dot products: fp16 operands, exact products, fp32 accumulation (numpy's order), unscaled."""
    sv, sq = pow2_scale(np.abs(rows).max()), pow2_scale(np.abs(q).max())
    r16 = (rows * np.float32(sv)).astype(np.float16).astype(np.float32)
    q16 = (q * np.float32(sq)).astype(np.float16).astype(np.float32)
    acc = (r16 * q16[None, :]).sum(axis=1, dtype=np.float32)
    return acc.astype(np.float64) / (sv * sq), sv, sq


def e_dot16(q, rows, sv, sq):
    dim = q.shape[0]
    nq = float(np.dot(q.astype(np.float64), q.astype(np.float64)))
    nv_max = float((rows.astype(np.float64) ** 2).sum(axis=1).max())
    e16 = 2.0 ** -10 + 2.0 ** -22 + 4.0 * gamma(dim)
    flush = float(np.abs(q).astype(np.float64).sum()) * 2.0 ** -14 / sv + np.sqrt(dim * nv_max) * 2.0 ** -14 / sq
    return 1.01 * (e16 * np.sqrt(nq * nv_max) + flush)


def datasets(rng, dim):
    n = 400
    yield "normal", rng.standard_normal(dim).astype(np.float32), rng.standard_normal((n, dim)).astype(np.float32)
    yield "all positive (no cancellation: the accumulation error adds up)", \
        np.abs(rng.standard_normal(dim)).astype(np.float32), np.abs(rng.standard_normal((n, dim))).astype(np.float32)
    yield "big", (rng.standard_normal(dim) * 1e4).astype(np.float32), (rng.standard_normal((n, dim)) * 1e4).astype(np.float32)
    yield "tiny", (rng.standard_normal(dim) * 1e-4).astype(np.float32), (rng.standard_normal((n, dim)) * 1e-4).astype(np.float32)
    spiky = (rng.standard_normal((n, dim)) * 1e-5).astype(np.float32)
    spiky[np.arange(n), rng.integers(0, dim, size=n)] = (rng.standard_normal(n) * 300).astype(np.float32)
    yield "spiky rows (most elements under fp16's range after scaling)", rng.standard_normal(dim).astype(np.float32), spiky
    qs = (rng.standard_normal(dim) * 1e-5).astype(np.float32)
    qs[3] = 77.0
    yield "spiky query", qs, rng.standard_normal((n, dim)).astype(np.float32)
    c = (rng.standard_normal(dim) * 12 + 100).astype(np.float32)
    yield "clustered far from the origin", c, (c + rng.standard_normal((n, dim)).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    half = (np.ones(dim) * (1.0 + 2.0 ** -12)).astype(np.float32)   # every element a rounding tie-breaker for fp16
    yield "fp16 half-way values", half, np.tile(half, (n, 1)) * rng.choice([-1.0, 1.0], size=(n, 1)).astype(np.float32)


@pytest.mark.parametrize("dim", [16, 96, 768, 2048])
def test_dot_product_estimates_stay_inside_their_bounds(oracle, dim):
    rng = np.random.default_rng(1000 + dim)
    for name, q, rows in datasets(rng, dim):
        seq = np.array([seq_dot(oracle, q, r) for r in rows[:60]])
        q64, r64 = q.astype(np.float64), rows[:60].astype(np.float64)
        nq, nv = float(q64 @ q64), (r64 ** 2).sum(axis=1)
        # fp32 nominations: another order of the same fp32 sum (numpy: blocked / pairwise): 2 gamma |q||v|
        est32 = (rows[:60] * q[None, :]).sum(axis=1, dtype=np.float32).astype(np.float64)
        assert np.all(np.abs(est32 - seq) <= 2.0 * gamma(dim) * np.sqrt(nq * nv) * (1 + 1e-6) + 1e-300), name
        # fp16 sketch
        est16, sv, sq = sketch_dots(q, rows)
        bound = e_dot16(q, rows, sv, sq) + 2.2 * gamma(dim) * np.sqrt(nq * nv.max())
        err = np.abs(est16[:60] - seq)
        assert np.all(err <= bound), (name, float(err.max()), bound)


def test_the_bound_is_not_vacuous(oracle):
    """For ordinary data the sketch's bound is a small fraction of the cosine's range (so k + max(32, k / 2) nominations certify),
    and the real error is well inside it."""
    rng = np.random.default_rng(5)
    dim = 768
    q, rows = rng.standard_normal(dim).astype(np.float32), rng.standard_normal((300, dim)).astype(np.float32)
    est16, sv, sq = sketch_dots(q, rows)
    seq = np.array([seq_dot(oracle, q, r) for r in rows])
    nq, nv = float(q.astype(np.float64) @ q.astype(np.float64)), (rows.astype(np.float64) ** 2).sum(axis=1)
    for r in range(20):     # seq_dot IS the oracle's sum: through the oracle's own MAXIMUM_INNER_PRODUCT map, bit for bit
        d = np.float32(seq[r])
        want = np.float32(d + np.float32(1.0)) if d >= 0 else np.float32(np.float32(1.0) / np.float32(np.float32(1.0) - d))
        assert np.float32(oracle.vector_score(3, q, rows[r])) == want
    rel = np.abs(est16 - seq) / np.sqrt(nq * nv)
    bound_rel = e_dot16(q, rows, sv, sq) / np.sqrt(nq * nv.max())
    assert bound_rel < 1.3e-3 and rel.max() < bound_rel / 5


def test_score_maps_are_monotone_and_the_euclidean_bound_works_in_distance_units(oracle):
    """knn_result_upper for EUCLIDEAN: a row whose estimated score is <= m has d2 >= 1/m - 1 - E, hence a result <= 1 / (1 + that)."""
    rng = np.random.default_rng(6)
    dim = 128
    q = (rng.standard_normal(dim) * 10).astype(np.float32)
    rows = (q + rng.standard_normal((200, dim)).astype(np.float32) * np.float32(0.5)).astype(np.float32)
    q64, r64 = q.astype(np.float64), rows.astype(np.float64)
    nq, nv = float(q64 @ q64), (r64 ** 2).sum(axis=1)
    d2_est = np.maximum(np.float32(nq) + nv.astype(np.float32) - np.float32(2.0) * (rows * q[None, :]).sum(axis=1, dtype=np.float32), np.float32(0))
    e_d2 = 4.5 * gamma(dim) * (nq + nv.max()) + 4 * U
    for r in range(200):
        s_exact = float(oracle.vector_score(2, q, rows[r]))
        m = 1.0 / (1.0 + float(d2_est[r]))
        d2 = 1.0 / m - 1.0
        lo = d2 - (e_d2 + 32 * U * (1.0 + d2))
        upper = 1.0 / (1.0 + max(lo, 0.0)) * (1 + 1e-6)
        assert s_exact <= upper, (r, s_exact, upper)
