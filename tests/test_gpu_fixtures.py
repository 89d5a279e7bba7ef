"""The committed search-level fixtures against the HIP path, through the C ABI.  The corpus is generated ON THE DEVICE
(rlr_index_fill_synthetic) and checked by its sha256, the expected rows and f32 bit patterns come from
tests/golden/search_fixtures.json: no oracle code runs in this test."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "search_fixtures.json")) as _f:
    DOC = json.load(_f)


def _u32(v):
    return np.asarray(v, dtype=np.uint32)


@pytest.mark.parametrize("corpus", DOC["corpora"], ids=[c["name"] for c in DOC["corpora"]])
def test_gpu_reproduces_the_committed_fixtures(rlr, corpus):
    n, dim, dtype = corpus["n"], corpus["dim"], corpus["dtype"]
    eng = rlr.RagEngine(dim, dtype)
    try:
        eng.index.fill_synthetic(n, seed=corpus["seed"], n_clusters=corpus["n_clusters"])
        eng._chunks = [rlr.DocumentChunk(str(i), "synthetic", "", i) for i in range(n)]
        eng._row_of = {str(i): i for i in range(n)}
        stored = eng.index.fetch_rows(np.arange(n))
        assert hashlib.sha256(np.ascontiguousarray(stored).tobytes()).hexdigest() == corpus["rows_sha256"], \
            "the device corpus generator no longer produces the committed corpus"
        q = _u32(corpus["query_bits"]).view(np.float32)          # raw (un-normalised) query, as the engine API takes it
        assert hashlib.sha256(np.ascontiguousarray(q).tobytes()).hexdigest() == corpus["query_sha256"]
        pool = None
        for call in corpus["calls"]:
            a, e, kind = call["args"], call["expect"], call["kind"]
            ctx = f"{corpus['name']}: {kind} {a}"
            w = None
            if "w_e" in a:
                w = rlr.QueryWeights(embedding=a["w_e"], lexical=a["w_l"])
            lex = [(str(r), s) for r, s in a.get("lex", [])]
            if kind == "search":
                got = eng.search(q, a["top_k"], weights=w, lexical=lex, stage=a.get("stage", 0))
            elif kind == "search_with_diversity":
                got = eng.search_with_diversity(q, a["top_k"], a["diversity"], weights=w, lexical=lex)
            elif kind == "embedding_candidates":
                ec = eng.get_embedding_candidates(q, a["count"])
                assert [int(c) for c, _ in ec] == e["rows"], ctx
                assert np.array_equal(bits([s for _, s in ec]), _u32(e["score_bits"])), ctx
                continue
            elif kind == "mmr":
                assert pool is not None
                order, mm = eng.index.mmr_select(np.array([g.row for g in pool], np.uint64),
                                                 np.array([g.score for g in pool], np.float32), a["top_k"], a["lambda"])
                assert [int(x) for x in order] == e["order"], ctx
                assert np.array_equal(bits(mm[1:]), _u32(e["mmr_bits_from_second_pick"])), ctx
                continue
            else:
                raise AssertionError(f"unknown fixture call {kind}")
            if kind == "search" and a["top_k"] == 300:
                pool = got
            assert [g.row for g in got] == e["rows"], ctx
            assert np.array_equal(bits([g.score for g in got]), _u32(e["score_bits"])), ctx
            assert np.array_equal(bits([g.embedding_score for g in got]), _u32(e["embedding_bits"])), ctx
            assert np.array_equal(bits([g.lexical_score for g in got]), _u32(e["lexical_bits"])), ctx
    finally:
        eng.close()
