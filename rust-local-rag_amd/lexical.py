"""BM25 term of the hybrid score with the postings in HBM.

Mirror of the reference's `LexicalIndex` (src/rag_engine.rs:2083-2237) keyed by index row instead
of chunk id.  Tokenisation (`tokenize`, :2242-2247) is host-language work and stays here -- Python's
str methods carry the Unicode tables Rust's `char::is_alphanumeric` / `to_lowercase` use -- while the
dictionary, the postings and the BM25 arithmetic live behind include/rlr_lexical.h.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _native as N


def tokenize(text: str) -> List[str]:
    """rag_engine.rs:2242-2247: split at non-alphanumeric chars, keep tokens of >= 3 BYTES, lower-case.
    (`str.isalnum` is Unicode L*/N*; Rust's is Alphabetic || Numeric -- they differ only on combining
    marks with the Other_Alphabetic property.)"""
    out, cur = [], []
    for ch in text:
        if ch.isalnum():
            cur.append(ch)
        else:
            if cur:
                out.append("".join(cur))
                cur = []
    if cur:
        out.append("".join(cur))
    return [t.lower() for t in out if len(t.encode("utf-8")) >= 3]


class LexicalIndex:
    MAX_LIMIT = 8192

    def __init__(self, device: int = 0):
        self._L = N.lib()
        h = C.c_void_p()
        N.check(self._L.rlr_lexical_create(device, C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.rlr_lexical_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _joined(tokens: Sequence[str]) -> bytes:
        return " ".join(tokens).encode("utf-8")

    def add_chunk(self, row: int, text: str) -> None:
        """LexicalIndex::add_chunk :2106-2137"""
        self.add_tokens(row, tokenize(text))

    def add_tokens(self, row: int, tokens: Sequence[str]) -> None:
        b = self._joined(tokens)
        N.check(self._L.rlr_lexical_add_chunk(self._h, row, b, len(b)))

    def remove_rows(self, rows) -> None:
        """remove_chunk :2139-2167 + the row compaction of GpuIndex.delete_rows"""
        r = np.ascontiguousarray(rows, dtype=np.uint64).ravel()
        if r.size:
            N.check(self._L.rlr_lexical_remove_rows(self._h, r.ctypes.data_as(N.u64p), r.size))

    def clear(self) -> None:
        N.check(self._L.rlr_lexical_clear(self._h))

    def contains(self, row: int) -> bool:
        st = self._L.rlr_lexical_contains(self._h, row)
        if st < 0:
            N.check(st)
        return bool(st)

    def info(self) -> dict:
        v = [C.c_uint64() for _ in range(4)]
        N.check(self._L.rlr_lexical_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("total_docs", "total_length", "n_terms", "n_postings"), (x.value for x in v)))

    def segments(self) -> dict:
        """device posting segments: what the last commits rebuilt (appends rebuild only the appended segment)"""
        v = [C.c_uint64() for _ in range(5)]
        N.check(self._L.rlr_lexical_segments(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("main_postings", "appended_postings", "full_rebuilds", "append_rebuilds", "select_retries"),
                        (x.value for x in v)))

    def score(self, query: str, limit: int) -> Tuple[np.ndarray, np.ndarray]:
        """LexicalIndex::score :2169-2225 -> (rows u64, scores f32), (score desc, row asc)"""
        return self.score_tokens(tokenize(query), limit)

    def score_tokens(self, tokens: Sequence[str], limit: int) -> Tuple[np.ndarray, np.ndarray]:
        b = self._joined(tokens)
        cap = self.MAX_LIMIT if limit == 0 else min(limit, self.MAX_LIMIT)
        rows = np.zeros(max(cap, 1), dtype=np.uint64)
        sc = np.zeros(max(cap, 1), dtype=np.float32)
        n = C.c_uint32()
        N.check(self._L.rlr_lexical_score(self._h, b, len(b), limit, rows.ctypes.data_as(N.u64p),
                                          sc.ctypes.data_as(N.f32p), C.byref(n)))
        return rows[: n.value], sc[: n.value]


def tokenize_ascii(text: str) -> List[str]:
    """the library's own tokenizer (rlr_tokenize_ascii): exact for ASCII text"""
    b = text.encode("utf-8")
    need = C.c_size_t()
    buf = C.create_string_buffer(max(len(b), 1))
    N.check(N.lib().rlr_tokenize_ascii(b, len(b), buf, len(b), C.byref(need)))
    s = buf.raw[: need.value].decode("utf-8")
    return s.split(" ") if s else []
