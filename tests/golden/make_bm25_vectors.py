#!/usr/bin/env python3
"""Writes tests/golden/bm25_vectors.json: BM25 score vectors derived BY HAND-STYLE EXACT ARITHMETIC, independent of
oracle/lexical.py and of the library.

    python tests/golden/make_bm25_vectors.py

The reference pins no BM25 value (its one LexicalIndex test, rag_engine.rs:2295-2326, checks contains / drop_stale).
These vectors pin the VALUES of `LexicalIndex::score` (rag_engine.rs:2169-2225) as data: every f32 operation of

    avg   = total_length as f32 / total_docs as f32                                              (:2187-2191)
    idf   = ((N - df + 0.5) / (df + 0.5)).ln().max(0.0)                                           (:2199-2202)
    denom = tf + k1 * (1.0 - b + b * (doc_length / avg))        k1 = 1.5, b = 0.75               (:2211)
    score = idf * (tf * (k1 + 1.0)) / denom ;   scores[doc] += score                              (:2216-2217)

is evaluated with exact rational arithmetic (fractions.Fraction) and then rounded to the nearest binary32 (ties to
even) -- which IS IEEE-754 arithmetic, by definition -- and `ln` with 80-digit decimal arithmetic rounded once to
binary32 (the correctly rounded result; every case below keeps the true value at least 1/16 ulp away from a rounding
boundary, so any logf with the usual < 1 ulp error agrees).  No numpy, no libm, none of the project's code.

What the reference leaves to HashMap order is fixed the way the build defines it: query terms accumulate in order of
first occurrence; results order by (score desc, row asc); rows whose score is 0 (idf floored at 0) are not reported
(they add nothing to the hybrid score: l = 0 / max = 0, rag_engine.rs:527-530).
"""
from __future__ import annotations

import json
import os
import struct
from decimal import Decimal, getcontext
from fractions import Fraction

getcontext().prec = 80
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "bm25_vectors.json")


def to_f32(x: Fraction) -> Fraction:
    """round-to-nearest-even of an exact rational to binary32 (normal range; 0 stays 0), returned exactly"""
    if x == 0:
        return Fraction(0)
    sign = -1 if x < 0 else 1
    a = abs(x)
    e = 0
    while a >= 2:
        a /= 2
        e += 1
    while a < 1:
        a *= 2
        e -= 1
    assert -126 <= e <= 127, "outside the normal range"
    scaled = a * (1 << 23)                      # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return sign * Fraction(n, 1 << 23) * (Fraction(2) ** e)


def bits(x: Fraction) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]   # x is exactly representable: float() is exact


def ln_f32(r: Fraction) -> Fraction:
    """correctly rounded binary32 of ln(r); asserts the true value is not within 1/16 ulp of a rounding boundary"""
    d = (Decimal(r.numerator) / Decimal(r.denominator)).ln()
    exact = Fraction(d)                          # 80 digits: far more than the 1/16-ulp margin needs
    y = to_f32(exact)
    if y != 0:
        ulp = abs(to_f32(y * (1 + Fraction(1, 1 << 22))) - y) or Fraction(1, 1 << 149)
        mid_dist = abs(abs(exact - y) - ulp / 2)
        assert mid_dist > ulp / 16, f"ln({r}) sits too close to a rounding boundary: pick another case"
    return y


def bm25(docs, query_terms, limit):
    """docs: list of token lists (row = position); -> [(row, score as exact Fraction of an f32)]"""
    lengths = [len(d) for d in docs]
    rows_with_tokens = [r for r, d in enumerate(docs) if d]
    n_docs = len(rows_with_tokens)
    total_len = sum(lengths)
    if n_docs == 0:
        return []
    f = lambda v: to_f32(Fraction(v))
    avg = to_f32(f(total_len) / f(n_docs))
    k1, b = f(Fraction(3, 2)), f(Fraction(3, 4))
    scores = {}
    seen = []
    for t in query_terms:
        if t not in seen:
            seen.append(t)
    for term in seen:
        postings = [(r, d.count(term)) for r, d in enumerate(docs) if term in d]
        if not postings:
            continue
        df = f(len(postings))
        t1 = to_f32(f(n_docs) - df)
        t2 = to_f32(t1 + Fraction(1, 2))
        t3 = to_f32(df + Fraction(1, 2))
        ratio = to_f32(t2 / t3)
        idf = ln_f32(ratio)
        if idf < 0:
            idf = Fraction(0)                    # .max(0.0)
        for r, tf_i in postings:
            dl, tf = f(lengths[r]), f(tf_i)
            q = to_f32(dl / avg)
            m = to_f32(b * q)
            a = to_f32(Fraction(1) - b)          # 1.0 - b
            s1 = to_f32(a + m)
            kk = to_f32(k1 * s1)
            denom = to_f32(tf + kk)
            n1 = to_f32(k1 + 1)                  # k1 + 1.0
            n2 = to_f32(tf * n1)
            n3 = to_f32(idf * n2)
            score = to_f32(n3 / denom)
            scores[r] = to_f32(scores.get(r, Fraction(0)) + score)
    res = [(r, s) for r, s in scores.items() if s > 0]
    res.sort(key=lambda rs: (-rs[1], rs[0]))
    return res[:limit] if limit > 0 else res


def lcg_corpus(n_docs, vocab, seed):
    x = seed
    docs = []
    for _ in range(n_docs):
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        m = 1 + (x >> 8) % 9
        d = []
        for _ in range(m):
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF
            # squared index: low-numbered terms are frequent, high-numbered ones rare
            i = ((x >> 10) % len(vocab)) * ((x >> 20) % len(vocab)) // len(vocab)
            d.append(vocab[i])
        docs.append(d)
    return docs


def build():
    cases = []
    small = [["alpha", "beta", "beta", "gamma"], ["alpha", "delta"], ["epsilon", "zeta", "eta", "theta", "iota"],
             ["beta"], ["alpha", "alpha", "alpha", "kappa", "lambda", "beta"], [], ["omega", "omega"]]
    vocab = [f"w{i:02d}x" for i in range(24)]
    big = lcg_corpus(60, vocab, 20261004)
    for name, docs, queries in (
        ("seven_chunks_one_empty", small,
         [(["beta"], 10), (["alpha"], 10), (["omega"], 10), (["alpha", "beta"], 10), (["beta", "alpha"], 10),
          (["kappa", "beta", "kappa", "missing"], 2), (["gamma", "delta", "iota"], 0), (["missing"], 5)]),
        ("sixty_chunks_zipf", big,
         [([vocab[0]], 0), ([vocab[5]], 10), ([vocab[1], vocab[2], vocab[3]], 15), ([vocab[9], vocab[0], vocab[15]], 5),
          ([vocab[3], vocab[1], vocab[2]], 15), ([vocab[20], vocab[21], vocab[22], vocab[23]], 0)]),
    ):
        qs = []
        for terms, limit in queries:
            res = bm25(docs, terms, limit)
            qs.append({"terms": terms, "limit": limit, "rows": [r for r, _ in res], "score_bits": [bits(s) for _, s in res]})
        cases.append({"name": name, "docs": docs, "queries": qs})
    return {"format": 1, "what": "hand-derived (exact rational + correctly rounded) BM25 vectors; see make_bm25_vectors.py",
            "reference": "src/rag_engine.rs:2169-2225", "cases": cases}


if __name__ == "__main__":
    doc = build()
    with open(OUT, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", sum(len(c["queries"]) for c in doc["cases"]), "queries")
