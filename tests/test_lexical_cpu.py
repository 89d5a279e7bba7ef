"""Oracle of the lexical term (oracle/lexical.py) against the one case the reference's own tests
hold for LexicalIndex, plus hand-computed BM25 values and the tokenizer's edge cases."""
import importlib
import math

import numpy as np

from oracle import lexical as OL


def test_reference_contains_and_drop_stale():
    """rag_engine.rs:2295-2326 test_lexical_index_contains_and_drop_stale"""
    ix = OL.LexicalIndex()
    ix.add_chunk("chunk1", "hello world")
    ix.add_chunk("chunk2", "foo bar baz")
    ix.add_chunk("chunk3", "test document")
    assert ix.contains("chunk1") and ix.contains("chunk2") and ix.contains("chunk3")
    assert not ix.contains("chunk4")
    ix.drop_stale({"chunk1", "chunk2"})
    assert ix.contains("chunk1") and ix.contains("chunk2")
    assert not ix.contains("chunk3")
    assert ix.total_docs == 2 and ix.total_length == 5


def test_tokenize_rules():
    assert OL.tokenize("Hello, World! it's a DB-9 plug") == ["hello", "world", "plug"]   # < 3 bytes dropped
    assert OL.tokenize("naïve café ÉTÉ") == ["naïve", "café", "été"]                      # Unicode lower-casing
    assert OL.tokenize("né") == ["né"]              # 2 chars but 3 bytes: kept (`token.len()` counts bytes)
    assert OL.tokenize("ab  __ c") == []
    assert OL.tokenize("x86_64 abc123") == ["x86", "abc123"]


def test_bm25_hand_computed():
    ix = OL.LexicalIndex()
    ix.add_chunk("a", "alpha beta beta gamma")        # len 4
    ix.add_chunk("b", "alpha delta")                  # len 2
    ix.add_chunk("c", "epsilon zeta eta theta iota")  # len 5 ("eta" has 3 bytes)
    assert ix.total_docs == 3 and ix.total_length == 11
    res = dict(ix.score("beta", 10))
    avg = np.float32(11) / np.float32(3)
    idf = np.float32(math.log((3 - 1 + 0.5) / (1 + 0.5)))
    want = idf * (np.float32(2) * np.float32(2.5)) / (np.float32(2) + np.float32(1.5) * (np.float32(0.25) + np.float32(0.75) * (np.float32(4) / avg)))
    assert set(res) == {"a"} and abs(float(res["a"]) - float(want)) < 1e-6
    # "alpha" is in 2 of 3 docs: idf = ln(1.5/2.5) < 0 -> floored at 0 -> both documents score exactly 0.0
    res = ix.score("alpha", 10)
    assert [c for c, _ in res] == ["a", "b"] and all(float(s) == 0.0 for _, s in res)
    assert ix.score("alpha", 10, keep_zero=False) == []
    assert ix.score("it is", 10) == [] and ix.score("unknownterm", 10) == []
    # limit truncation, sorted descending
    res = ix.score("beta delta epsilon", 2)
    assert len(res) == 2 and float(res[0][1]) >= float(res[1][1])


def test_re_add_replaces_and_empty_text_is_not_indexed():
    ix = OL.LexicalIndex()
    ix.add_chunk("a", "one two three")
    ix.add_chunk("a", "four five")
    assert ix.total_docs == 1 and ix.total_length == 2 and "one" not in ix.term_postings
    ix.add_chunk("b", "a b c")     # no token survives
    assert not ix.contains("b") and ix.total_docs == 1
    ix.remove_chunk("a")
    assert ix.total_docs == 0 and ix.total_length == 0 and ix.score("four", 5) == []


def test_library_tokenizer_ascii_exact(rlr):
    lex = importlib.import_module("rust-local-rag_amd.lexical")
    for text in ["Hello, World! it's a DB-9 plug", "x86_64 abc123", "", "ab cd", "The Quick brown-fox_jumps"]:
        assert lex.tokenize_ascii(text) == OL.tokenize(text)
        assert lex.tokenize(text) == OL.tokenize(text)
    assert lex.tokenize("naïve café ÉTÉ né") == OL.tokenize("naïve café ÉTÉ né")


def _bm25_vectors():
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "bm25_vectors.json")) as f:
        return json.load(f), root


def test_oracle_against_the_hand_derived_bm25_vectors():
    """tests/golden/bm25_vectors.json: BM25 values derived with exact rational arithmetic + correctly rounded ln
    (make_bm25_vectors.py: no numpy, no libm, none of the project's code) pin `LexicalIndex::score`'s VALUES
    (rag_engine.rs:2169-2225), which no reference test does."""
    doc, _ = _bm25_vectors()
    n_pos = 0
    for case in doc["cases"]:
        ix = OL.LexicalIndex()
        for r, tokens in enumerate(case["docs"]):
            ix.add_chunk(r, " ".join(tokens), rank=r)
        for q in case["queries"]:
            res = ix.score(" ".join(q["terms"]), q["limit"], keep_zero=False)
            assert [r for r, _ in res] == q["rows"], (case["name"], q["terms"])
            assert [int(np.float32(s).view(np.uint32)) for _, s in res] == q["score_bits"], (case["name"], q["terms"])
            n_pos += len(res)
    assert n_pos >= 50


def test_bm25_vector_file_is_what_its_generator_writes():
    import importlib.util
    import json
    import os
    doc, root = _bm25_vectors()
    spec = importlib.util.spec_from_file_location("make_bm25_vectors", os.path.join(root, "tests", "golden", "make_bm25_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert json.loads(json.dumps(mod.build())) == doc
