"""CPU restatement of the reference's LexicalIndex (BM25) -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/rag_engine.rs:
  LexicalIndex fields            :2083-2090      add_chunk      :2106-2137
  remove_chunk                   :2139-2167      score          :2169-2225
  contains / drop_stale          :2227-2237      tokenize       :2242-2247
Pure-Python loops with numpy float32 scalars (every operation rounds to binary32, as Rust's f32
arithmetic does) and the C library's logf for `f32::ln` -- small cases only.

Pinning: the reference's own tests hold ONE case for this structure
(`test_lexical_index_contains_and_drop_stale`, :2295-2326), reproduced in tests/test_lexical_cpu.py.
BM25 score VALUES are pinned by no reference test or fixture; since round 2 they are pinned by
tests/golden/bm25_vectors.json -- vectors derived with exact rational arithmetic and a correctly
rounded ln (tests/golden/make_bm25_vectors.py: no numpy, no libm, none of this project's code)
from the formula as the reference spells it (:2187-2217).  This file and the GPU index must both
reproduce them bit for bit (tests/test_lexical_cpu.py, tests/test_gpu_lexical.py).
Two things the reference leaves to HashMap iteration order are fixed here the way the GPU library
defines them: query terms accumulate in order of first occurrence, ties order by insertion rank
(the row number in the tests).
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.argtypes = [ctypes.c_float]
_libm.logf.restype = ctypes.c_float
f32 = np.float32


def tokenize(text: str):
    """:2242-2247  split(|c| !c.is_alphanumeric()).filter(len >= 3 bytes).map(to_lowercase)"""
    tokens, start = [], None
    for i, ch in enumerate(text):
        if ch.isalnum():
            if start is None:
                start = i
        elif start is not None:
            tokens.append(text[start:i])
            start = None
    if start is not None:
        tokens.append(text[start:])
    return [t.lower() for t in tokens if len(t.encode("utf-8")) >= 3]


class LexicalIndex:
    def __init__(self):
        self.term_postings = {}   # term -> {id: count}
        self.doc_lengths = {}
        self.doc_terms = {}
        self.total_docs = 0
        self.total_length = 0
        self.rank = {}            # id -> insertion rank (tie order; the reference's is arbitrary)
        self._next = 0

    def add_chunk(self, cid, text, rank=None):
        if cid in self.doc_terms:
            self.remove_chunk(cid)
        self.rank[cid] = self._next if rank is None else rank
        self._next += 1
        tokens = tokenize(text)
        if not tokens:
            return
        counts = {}
        for t in tokens:
            counts[t] = counts.get(t, 0) + 1
        doc_length = sum(counts.values())
        if doc_length == 0:
            return
        for term, c in counts.items():
            self.term_postings.setdefault(term, {})[cid] = c
        self.doc_lengths[cid] = doc_length
        self.doc_terms[cid] = counts
        self.total_docs += 1
        self.total_length += doc_length

    def remove_chunk(self, cid):
        counts = self.doc_terms.pop(cid, None)
        if counts is not None:
            for term in counts:
                p = self.term_postings.get(term)
                if p is not None:
                    p.pop(cid, None)
                    if not p:
                        del self.term_postings[term]
            length = self.doc_lengths.pop(cid, None)
            if length is not None:
                self.total_length = self.total_length - length if self.total_length >= length else 0
            if self.total_docs > 0:
                self.total_docs -= 1
        else:
            self.doc_lengths.pop(cid, None)
        if self.total_docs == 0:
            self.total_length = 0

    def contains(self, cid):
        return cid in self.doc_terms

    def drop_stale(self, valid_ids):
        for cid in [c for c in self.doc_terms if c not in valid_ids]:
            self.remove_chunk(cid)

    def score(self, query, limit, keep_zero=True):
        """-> [(id, f32 score)] sorted (score desc, rank asc), truncated to `limit` (0 = all)."""
        if self.total_docs == 0:
            return []
        tokens = tokenize(query)
        if not tokens:
            return []
        unique = list(dict.fromkeys(tokens))
        avg = f32(self.total_length) / f32(self.total_docs)
        k1, b = f32(1.5), f32(0.75)
        scores = {}
        for term in unique:
            postings = self.term_postings.get(term)
            if postings is None:
                continue
            df = f32(len(postings))
            ratio = (f32(self.total_docs) - df + f32(0.5)) / (df + f32(0.5))
            idf = f32(_libm.logf(float(ratio)))
            idf = idf if idf > 0 else f32(0.0)      # f32::max(0.0): NaN -> 0.0
            for cid, tf_i in postings.items():
                dl = f32(self.doc_lengths.get(cid, 0))
                if dl == 0:
                    continue
                tf = f32(tf_i)
                denom = tf + k1 * (f32(1.0) - b + b * (dl / avg))
                if denom == 0:
                    continue
                s = idf * (tf * (k1 + f32(1.0))) / denom
                scores[cid] = f32(scores.get(cid, f32(0.0)) + s)
        res = [(c, s) for c, s in scores.items() if keep_zero or s > 0]
        res.sort(key=lambda cs: (-float(cs[1]), self.rank[cs[0]]))
        if limit > 0 and len(res) > limit:
            res = res[:limit]
        return res
