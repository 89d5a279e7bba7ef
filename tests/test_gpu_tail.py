"""The fused select -> re-score -> sort tail of a single-query search (csrc/tail.hip) against the oracle, in each of its
modes: DIRECT (one pass: collect + re-score where the candidates are found), REFINE (digit-2 histogram, then collect +
re-score + last-workgroup sort) and the split five-launch pipeline it replaced (RLR_TAIL=0, still the path of rows too
wide for the staged re-score).  The switches are read when an index is created, so one process can hold all three."""
import os

import numpy as np
import pytest

from conftest import bits
from test_gpu_parity import make_index, oracle_topk

pytestmark = pytest.mark.gpu

MODES = {"direct": {"RLR_TAIL": "1", "RLR_TAIL_DIRECT_MAX": "4096"},
         "refine": {"RLR_TAIL": "1", "RLR_TAIL_DIRECT_MAX": "0"},
         "default": {},
         "split": {"RLR_TAIL": "0"}}


class tail_mode:
    def __init__(self, name):
        self.env = MODES[name]

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("RLR_TAIL", "RLR_TAIL_DIRECT_MAX")}
        for k in self.old:
            os.environ.pop(k, None)
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def check(ix, oracle, rows, qn, k):
    r, c = ix.search_topk(qn, k)
    wr, wc = oracle_topk(oracle, rows, qn, k)
    assert r.shape[1] == min(k, rows.shape[0])
    assert np.array_equal(r[0], wr)
    assert np.array_equal(bits(c[0]), bits(wc))


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("n,dim,dtype,k", [(20000, 768, "f32", 100), (4099, 384, "f32", 20), (70000, 1024, "f16", 300),
                                           (130, 100, "f32", 130), (3, 64, "f32", 2), (9000, 1536, "f32", 7),
                                           (5000, 2304, "f16", 50), (2500, 4096, "f32", 33)])
def test_tail_modes_bit_exact(rlr, oracle, mode, n, dim, dtype, k):
    rows = oracle.synth_rows(n, dim, seed=900 + n, n_clusters=11, f16=(dtype == "f16"))
    with tail_mode(mode):
        ix = make_index(rlr, rows, dtype=dtype)
    try:
        for qi in range(3):
            check(ix, oracle, rows, oracle.normalize(oracle.synth_query(dim, seed=7000 + qi)), k)
        # several queries in one call: every query slot has its own selection state
        qs = np.stack([oracle.normalize(oracle.synth_query(dim, seed=7100 + i)) for i in range(3)])
        if n >= 4096:
            os.environ["RLR_BATCH_MIN"] = "1000000"  # keep the call on the single-query pipelines
        try:
            with tail_mode(mode):
                ix2 = make_index(rlr, rows, dtype=dtype)
        finally:
            os.environ.pop("RLR_BATCH_MIN", None)
        r, c = ix2.search_topk(qs, k)
        for i in range(3):
            wr, wc = oracle_topk(oracle, rows, qs[i], k)
            assert np.array_equal(r[i], wr) and np.array_equal(bits(c[i]), bits(wc))
        ix2.close()
    finally:
        ix.close()


@pytest.mark.parametrize("mode", ["direct", "refine", "default"])
def test_tail_duplicates_ties_and_nan_rows(rlr, oracle, mode):
    """exact duplicates of the best rows (adjacent: one workgroup's slice holds them all), a NaN row, a zero row"""
    n, dim, k = 30000, 768, 100
    rows = oracle.synth_rows(n, dim, seed=31, n_clusters=5)
    qn = oracle.normalize(oracle.synth_query(dim, seed=32))
    e = oracle.scan(rows, qn)
    best = int(np.argmax(np.where(np.isnan(e), -np.inf, e)))
    rows[1000:1400] = rows[best]          # 400 exact ties next to each other
    rows[20000:20040] = rows[best]
    rows[77] = np.nan
    rows[78] = 0.0
    with tail_mode(mode):
        ix = make_index(rlr, rows)
    try:
        for kk in (1, 100, 300, 441, 500):
            check(ix, oracle, rows, qn, kk)
    finally:
        ix.close()


@pytest.mark.parametrize("mode", ["direct", "refine"])
def test_tail_local_list_overflow_takes_the_large_candidate_path(rlr, oracle, mode):
    """more than 1024 candidates inside one workgroup's slice (3000 adjacent copies of the best row): the local list
    overflows, the finish reports a band overflow and the host's large-candidate path answers -- same result.  (1.2 M short
    rows: stage 2's 256 workgroups then own ~4700 rows each; stage 1's never own more than 1024, their lists cannot overflow.)"""
    n, dim, k = 1_200_000, 64, 50
    rows = oracle.synth_rows(n, dim, seed=41)
    qn = oracle.normalize(oracle.synth_query(dim, seed=42))
    e = oracle.scan(rows, qn)
    best = int(np.argmax(e))
    rows[5000:8000] = rows[best]
    with tail_mode(mode):
        ix = make_index(rlr, rows)
    try:
        ix.profile_read(reset=True)
        check(ix, oracle, rows, qn, k)
        assert ix.profile_read().n_retries == (1 if mode == "refine" else 0)
        check(ix, oracle, rows, oracle.normalize(oracle.synth_query(dim, seed=43)), k)  # the counters were reset: next query fine
    finally:
        ix.close()


def test_tail_state_survives_interleaved_entry_points(rlr, oracle):
    """search, search_with_diversity (no sort/emit launch: the pool kernel orders the candidates), text-less engine search and
    score_rows share contexts; the tail's counters must be back at zero after each"""
    n, dim = 12000, 768
    rows = oracle.synth_rows(n, dim, seed=51, n_clusters=40)
    eng = rlr.RagEngine(dim, "f32")
    try:
        eng.index.upload(rows)
        eng._chunks = [rlr.DocumentChunk(str(i), "s", "", i) for i in range(n)]
        for i in range(4):
            q = oracle.synth_query(dim, seed=60 + i)
            got = eng.search_with_diversity(q, 20, 0.3)
            want = oracle.search_with_diversity(rows, q, 20, 0.3)[0]
            assert [g.row for g in got] == list(want)
            check(eng.index, oracle, rows, oracle.normalize(q), 100)
            eng.index.score_rows(oracle.normalize(q), [0, 5, 7])
            got = eng.search(q, 10)
            assert [g.row for g in got] == list(oracle.search(rows, q, 10)[0])
    finally:
        eng.close()


def test_synthetic_generator_flags_match_the_oracle(rlr, oracle):
    """the two generator modes bench.py's hostile configurations use: bit 31 of n_clusters = tight clusters (twin of
    rlr_o_synth_raw), bit 30 = the last 1 % of the rows repeat the first 1 % (a fill-level rule: two generator ranges)"""
    n, dim = 3000, 96
    ix = rlr.GpuIndex(dim)
    try:
        ix.fill_synthetic(n, seed=77, n_clusters=5 | 0x80000000)
        want = oracle.synth_rows(n, dim, seed=77, n_clusters=5 | 0x80000000)
        assert np.array_equal(bits(ix.fetch_rows(np.arange(n))), bits(want))
        # near-copies inside a cluster: the best cosine to ANOTHER row is ~0.999
        r, c = ix.search_topk(want[0], 2)
        assert int(r[0][0]) == 0 and float(c[0][1]) > 0.99
        ix.fill_synthetic(n, seed=78, row0=10, n_clusters=0x40000000)
        base = oracle.synth_rows(n, dim, seed=78, row0=10)
        want = np.concatenate([base[: n - n // 100], base[: n // 100]])
        assert np.array_equal(bits(ix.fetch_rows(np.arange(n))), bits(want))
    finally:
        ix.close()


def test_probe_bandwidth_modes(rlr):
    """the measured-peak probes of bench.py's roofline: plausible rates, every mode, errors on bad input"""
    ix = rlr.GpuIndex(768)
    try:
        ix.fill_synthetic(400_000, seed=5)                      # 1.2 GB
        for mode in (0, 1, 2, 3):
            gbps, ms = ix.probe_bandwidth(mode, 3)
            assert 200.0 < gbps < 16000.0 and ms > 0, (mode, gbps, ms)
        with pytest.raises(Exception):
            ix.probe_bandwidth(7, 1)
        r, c = ix.search_topk(rlr.normalize(np.ones(768, np.float32)), 10)   # the index still answers afterwards
        assert r.shape == (1, 10)
    finally:
        ix.close()
