"""Size-independent properties at BASELINE.json's full per-GPU sizes, far beyond what the oracle can scan
in seconds (corpora generated on the device): config 3 = 10 M x 768 f32 (30.7 GB) with 256 batched queries,
config 5's per-GPU share = 6.25 M x 1024 binary16 with 1024 batched queries and MMR 0.7.  Checked:
sortedness, self-consistency of the emitted scores with the reference-order re-score, top-k-ness against a
random sample, idempotence, single == batched == sharded, MMR batch == MMR single, a delete round trip."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

N, DIM, K, SEED = 10_000_000, 768, 100, 0x5EED0003


@pytest.fixture(scope="module")
def big(rlr):
    ix = rlr.GpuIndex(DIM)
    ix.fill_synthetic(N, seed=SEED)
    yield ix
    ix.close()


def _queries(rlr, n, seed=5):
    rng = np.random.default_rng(seed)
    return np.stack([rlr.normalize(rng.standard_normal(DIM).astype(np.float32)) for _ in range(n)])


def test_sorted_exact_and_topk_against_sample(rlr, big):
    qs = _queries(rlr, 3)
    rng = np.random.default_rng(1)
    for q in qs:
        r, c = big.search_topk(q, K)
        r, c = r[0], c[0]
        assert len(r) == K and len(set(r.tolist())) == K
        key = list(zip((-c).tolist(), r.tolist()))
        assert key == sorted(key)                                       # (score desc, row asc)
        assert np.array_equal(bits(c), bits(big.score_rows(q, r)))      # emitted == reference-order dot of that row
        sample = rng.choice(N, size=200_000, replace=False).astype(np.uint64)
        s = big.score_rows(q, sample)
        outside = ~np.isin(sample, r)
        assert s[outside].max() <= c[-1]                                # nothing in the sample beats the k-th result
        inside = sample[~outside]
        if inside.size:                                                  # and sampled members carry the same score
            pos = {int(x): i for i, x in enumerate(r)}
            assert all(bits([s[np.where(sample == x)[0][0]]])[0] == bits([c[pos[int(x)]]])[0] for x in inside)
        r2, c2 = big.search_topk(q, K)                                   # idempotent
        assert np.array_equal(r2[0], r) and np.array_equal(bits(c2[0]), bits(c))


def test_batched_equals_single_at_scale(rlr, big):
    qs = _queries(rlr, 256, seed=6)                                      # BASELINE config 3: 256 batched queries
    big.profile_read(reset=True)
    rb, cb = big.search_topk(qs, K)
    prof = big.profile_read()
    assert prof.n_batches == 1 and prof.n_batch_fallbacks == 0
    for i in range(0, 256, 37):
        r1, c1 = big.search_topk(qs[i], K)
        assert np.array_equal(r1[0], rb[i]) and np.array_equal(bits(c1[0]), bits(cb[i]))
    big.enable_batch_image(True)                                         # and through the nomination image
    ri, ci = big.search_topk(qs, K)
    big.enable_batch_image(False)
    assert np.array_equal(ri, rb) and np.array_equal(bits(ci), bits(cb))


def test_sharded_equals_single_at_scale(rlr, big):
    qs = _queries(rlr, 2, seed=7)
    mi = rlr.MultiGpuIndex(DIM, [0, 0, 0, 0])
    mi.fill_synthetic(N, seed=SEED)
    for q in qs:
        r1, c1 = big.search_topk(q, K)
        rm, cm = mi.search_topk(q, K)
        assert np.array_equal(r1, rm) and np.array_equal(bits(c1), bits(cm))
    mi.close()


def test_config5_shard_batched_search_and_mmr(rlr):
    """6.25 M x 1024 binary16 rows (50 M / 8 GPUs), 1024 batched queries, pool 300, MMR lambda 0.7"""
    n, dim, nq, k, lam = 6_250_000, 1024, 1024, 100, 0.7
    ix = rlr.GpuIndex(dim, "f16")
    ix.fill_synthetic(n, seed=0x5EED0005, n_clusters=4096)
    rng = np.random.default_rng(9)
    qs = np.stack([rlr.normalize(rng.standard_normal(dim).astype(np.float32)) for _ in range(nq)])
    pool = max(3 * k, k + 10)
    r, c = ix.search_topk(qs, pool)
    assert r.shape == (nq, pool)
    for i in (0, 511, 1023):                                             # batched == single, scores == reference order
        r1, c1 = ix.search_topk(qs[i], pool)
        assert np.array_equal(r1[0], r[i]) and np.array_equal(bits(c1[0]), bits(c[i]))
        assert np.array_equal(bits(c[i]), bits(ix.score_rows(qs[i], r[i])))
        assert (np.diff(c[i]) <= 0).all()
    scores = (np.float32(0.7) * c).astype(np.float32)
    order, mmr, cnt = ix.mmr_select_batch(r, scores, np.full(nq, pool, np.uint32), k, lam)
    assert (cnt == k).all()
    for i in (0, 300, 1023):                                             # batched MMR == single MMR
        o1, m1 = ix.mmr_select(r[i], scores[i], k, lam)
        assert np.array_equal(o1, order[i, :k]) and np.array_equal(bits(m1[1:]), bits(mmr[i, 1:k]))
        assert order[i, 0] == 0 and len(set(order[i, :k].tolist())) == k  # first pick = best score, no repeats
    ix.close()


def test_config4_shard_single_query_12_5m_rows(rlr):
    """BASELINE config 4's per-GPU share: 12.5 M x 768 f32 rows (100 M / 8 GPUs = 38.4 GB), single queries, as rank 3 of
    8 holds it (row0 = 3 x 12.5 M of the same synthetic stream).  The same size-independent properties as at 10 M rows,
    plus: the shard's answer is what a merge needs -- reference-order scores, (score desc, row asc) order."""
    n, row0 = 12_500_000, 3 * 12_500_000
    ix = rlr.GpuIndex(DIM)
    ix.fill_synthetic(n, seed=0x5EED0004, row0=row0)
    rng = np.random.default_rng(44)
    qs = _queries(rlr, 2, seed=44)
    for q in qs:
        r, c = ix.search_topk(q, K)
        r, c = r[0], c[0]
        assert len(r) == K and len(set(r.tolist())) == K and r.max() < n
        key = list(zip((-c).tolist(), r.tolist()))
        assert key == sorted(key)
        assert np.array_equal(bits(c), bits(ix.score_rows(q, r)))
        sample = rng.choice(n, size=200_000, replace=False).astype(np.uint64)
        s = ix.score_rows(q, sample)
        assert s[~np.isin(sample, r)].max() <= c[-1]
        r2, c2 = ix.search_topk(q, K)
        assert np.array_equal(r2[0], r) and np.array_equal(bits(c2[0]), bits(c))
    # the opt-in nomination copies give the same answer at this size too
    base = [ix.search_topk(q, K) for q in qs]
    for kw in (dict(on=True, single_query=True), dict(on=False, q8=True)):
        ix.enable_batch_image(**kw)
        for q, (rb, cb) in zip(qs, base):
            r, c = ix.search_topk(q, K)
            assert np.array_equal(r, rb) and np.array_equal(bits(c), bits(cb)), kw
        ix.enable_batch_image(False)
    ix.close()


def test_delete_round_trip(rlr):
    n = 1_000_000
    ix = rlr.GpuIndex(DIM)
    ix.fill_synthetic(n, seed=SEED + 9)
    q = _queries(rlr, 1, seed=8)[0]
    r, c = ix.search_topk(q, 20)
    dead = np.array([r[0][0], r[0][3], r[0][7]], dtype=np.uint64)
    ix.delete_rows(dead)
    r2, c2 = ix.search_topk(q, 17)
    keep = [i for i in range(20) if i not in (0, 3, 7)]
    want_rows = np.array([int(r[0][i]) - int((dead < r[0][i]).sum()) for i in keep], dtype=np.uint64)
    assert np.array_equal(r2[0], want_rows)
    assert np.array_equal(bits(c2[0]), bits(c[0][keep]))
    ix.close()
