"""Batched MMR over binary16 rows: with 16 or more pools the reference-order Gram matrices are computed on the f32 matrix cores
(csrc/exact.hip gram_mfma_f32_kernel: a K = 1 f32 matrix instruction is one step s = s + fl(a * b) of dot_product when a * b is
exact, which it is for binary16 values).  Picks AND logged MMR values must be bit-identical to the oracle's mmr_diversify
(rag_engine.rs:767-839) -- pool sizes around the 32-row tile edges, ragged batches, widths that are not a multiple of the
16-byte row unit, duplicated rows, NaN / Inf rows, awkward relevance."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def check_pools(oracle, rows, prow, psc, sizes, order, mmr, nsel, k, lam, what, step=1):
    for q in range(0, prow.shape[0], step):
        sz = int(sizes[q])
        if sz == 0:
            assert int(nsel[q]) == 0, (what, q)
            continue
        worder, wmmr = oracle.mmr(rows[prow[q, :sz].astype(np.int64)], psc[q, :sz], k, lam)
        assert int(nsel[q]) == len(worder), (what, q, int(nsel[q]), len(worder))
        assert np.array_equal(order[q, :nsel[q]], worder), (what, q)
        assert np.array_equal(bits(mmr[q, 1:nsel[q]]), bits(wmmr[1:])), (what, q)


@pytest.mark.parametrize("dim,P,m", [(64, 33, 17), (100, 64, 16), (1024, 65, 19), (260, 129, 16), (768, 300, 16), (1024, 308, 24),
                                     (128, 513, 16), (72, 1024, 16), (1004, 31, 40)])
def test_batched_mmr_on_the_f32_matrix_cores_matches_the_oracle(rlr, oracle, dim, P, m):
    n = max(3 * P, 2000)
    rows = oracle.synth_rows(n, dim, seed=9100 + P, n_clusters=7, f16=True)
    ix = rlr.GpuIndex(dim, "f16")
    ix.upload(rows)
    rng = np.random.default_rng(P * 31 + dim)
    for lam in (0.3, 0.7, 1.0):
        k = int(rng.choice([1, 5, min(100, P), min(160, P)]))   # (the oracle's literal loop is O(k^2 P dim) on one core)
        prow = np.zeros((m, P), np.uint64)
        psc = np.zeros((m, P), np.float32)
        sizes = np.full(m, P, np.uint32)
        for q in range(m):                    # pools as a search builds them: the best P rows of a query, relevance = 0.7 cos
            qv = oracle.normalize(oracle.synth_query(dim, seed=9200 + 17 * q + P))
            r, c = ix.search_topk(qv, P)
            prow[q], psc[q] = r[0], (np.float32(0.7) * c[0]).astype(np.float32)
        sizes[1] = max(1, P // 3)             # ragged
        sizes[2] = 1
        sizes[8] = 0
        sizes[9] = min(P, 32)
        sizes[3] = min(P, 33)
        order, mmr, nsel = ix.mmr_select_batch(prow, psc, sizes, k, lam)
        # every pool below 300 candidates, every third from there on (pools 0, 3, 6, 9: full, ragged and tile-edge sizes among them)
        check_pools(oracle, rows, prow, psc, sizes, order, mmr, nsel, k, lam, ("search pools", dim, P, lam, k), step=1 if P < 300 else 3)
    ix.close()


def test_batched_mmr_f16_awkward_pools(rlr, oracle):
    """duplicated rows (exact similarity ties, MMR ties decided by the visiting order), relevance with ties / zeros of both
    signs / non-finite values, rows holding NaN or Inf (non-finite similarities are skipped, :803), a bundle of near-parallel
    rows, lambda at both ends"""
    dim, P, m = 128, 96, 18
    rows = oracle.synth_rows(4000, dim, seed=9301, n_clusters=3, f16=True)
    rows[10:40] = rows[3]
    base = rows[50].copy()
    for i in range(60, 160):
        rows[i] = oracle.round_f16(oracle.normalize(base + np.float32(0.002) * rows[i]))
    rows[200, 5] = np.nan
    rows[201, 7] = np.inf
    rows[202, :] = 0.0
    rows[203, :] = oracle.round_f16(np.full(dim, 6.0e-8, np.float32))  # binary16 subnormals only (2^-24)
    ix = rlr.GpuIndex(dim, "f16")
    ix.upload(rows)
    rng = np.random.default_rng(77)
    prow = rng.integers(300, 4000, size=(m, P)).astype(np.uint64)
    psc = np.sort(rng.random((m, P)).astype(np.float32), axis=1)[:, ::-1].copy()
    sizes = np.full(m, P, np.uint32)
    prow[0, :40] = np.arange(3, 43)
    prow[1, :] = np.arange(60, 60 + P)
    prow[2, 7] = 200
    prow[3, 9] = 201
    prow[3, 0] = 201                                          # the Inf row is the first pick: every similarity to it matters
    psc[4, 10:20] = psc[4, 10]
    psc[5, 3], psc[5, 4], psc[5, 5], psc[5, 6] = np.float32(0.0), np.float32(-0.0), np.float32(np.nan), np.float32(np.inf)
    psc[6, :] = np.float32(0.5)
    prow[7, :] = np.tile(np.arange(10, 14), P // 4)
    prow[8, 5], prow[8, 6] = 202, 203
    for lam, k in ((0.0001, 50), (0.5, 96), (0.999, 30), (1.0, 96)):
        order, mmr, nsel = ix.mmr_select_batch(prow, psc, sizes, k, lam)
        check_pools(oracle, rows, prow, psc, sizes, order, mmr, nsel, k, lam, ("awkward", lam, k))
    ix.close()
