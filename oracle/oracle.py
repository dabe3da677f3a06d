"""ctypes front end of the CPU oracle (oracle/bm25_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py``; never by the product package.

PARITY UNPINNED: the reference holds no golden vectors for this path (its arithmetic runs in
PostgreSQL); the oracle restates the SQL of app/batch/bm25_stats.py:9-27 and
app/api/search_service.py:13-101 and is cross-checked against sqlite3 and hand-worked cases.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbm25_oracle.so")

SRC_WEB, SRC_NEWS = 1, 2


class Totals(C.Structure):
    _fields_ = [("n_done", C.c_uint64), ("n_wc", C.c_uint64), ("sum_wc", C.c_uint64), ("n_postings", C.c_uint64)]


class CorpusSpec(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_terms", C.c_uint32),
        ("doc_begin", C.c_uint32),
        ("doc_end", C.c_uint32),
        ("article_begin", C.c_uint32),
        ("article_end", C.c_uint32),
        ("zipf_s", C.c_double),
        ("body_min", C.c_uint32),
        ("body_max", C.c_uint32),
        ("title_len", C.c_uint32),
        ("desc_len", C.c_uint32),
    ]


def build_library() -> str:
    """Compiles the oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "bm25_oracle.c")
    if not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_library())
        p32, p64, p8, pd, pf = (C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8),
                                C.POINTER(C.c_double), C.POINTER(C.c_float))
        L.orc_zipf_cdf.argtypes = [C.c_uint32, C.c_double, p32]
        L.orc_gen_corpus.restype = C.c_uint64
        L.orc_gen_corpus.argtypes = [C.POINTER(CorpusSpec), p32, p32, p32, p32, p8, p32]
        L.orc_gen_queries.restype = C.c_int
        L.orc_gen_queries.argtypes = [C.c_uint64, C.c_uint32, p32, C.c_uint32, C.c_uint32, C.c_uint32, p32]
        L.orc_build.restype = C.c_void_p
        L.orc_build.argtypes = [p32, p32, p32, p8, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, p32, p8]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_finalize.argtypes = [C.c_void_p, p32, C.POINTER(Totals)]
        for name, rt in [("orc_df", p32), ("orc_idf", pd), ("orc_term_rows", p32), ("orc_doc_len", p32),
                         ("orc_row_ptr", p64), ("orc_news_ptr", p64), ("orc_post_unit", p32),
                         ("orc_post_weight", p32)]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_get_totals.argtypes = [C.c_void_p, C.POINTER(Totals)]
        L.orc_n_docs_stat.restype = C.c_uint64
        L.orc_n_docs_stat.argtypes = [C.c_void_p]
        L.orc_avgdl.restype = C.c_double
        L.orc_avgdl.argtypes = [C.c_void_p]
        L.orc_set_idf.argtypes = [C.c_void_p, pd, p8]
        for name in ("orc_search_rows", "orc_search_csr"):
            f = getattr(L, name)
            f.restype = C.c_uint32
            f.argtypes = [C.c_void_p, C.c_int, p32, C.c_uint32, C.c_uint32, p32, C.c_uint32, p32, pd, p32]
        _lib = L
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def zipf_cdf(n_terms: int, zipf_s: float = 1.0) -> np.ndarray:
    cdf = np.empty(n_terms, dtype=np.uint32)
    lib().orc_zipf_cdf(n_terms, zipf_s, _p(cdf, C.c_uint32))
    return cdf


def make_spec(seed=20260227, n_terms=5000, n_docs=1000, n_articles=0, doc_begin=0, article_begin=0,
              zipf_s=1.0, body_min=100, body_max=500, title_len=8, desc_len=24) -> CorpusSpec:
    return CorpusSpec(seed, n_terms, doc_begin, doc_begin + n_docs, article_begin, article_begin + n_articles,
                      zipf_s, body_min, body_max, title_len, desc_len)


def gen_corpus(spec: CorpusSpec, cdf: np.ndarray | None = None) -> dict:
    """Rows of the synthetic corpus as numpy arrays (unit-major)."""
    if cdf is None:
        cdf = zipf_cdf(spec.n_terms, spec.zipf_s)
    n = lib().orc_gen_corpus(C.byref(spec), _p(cdf, C.c_uint32), None, None, None, None, None)
    term = np.empty(n, np.uint32)
    unit = np.empty(n, np.uint32)
    tf = np.empty(n, np.uint32)
    meta = np.empty(n, np.uint8)
    wc = np.empty(spec.doc_end - spec.doc_begin, np.uint32)
    lib().orc_gen_corpus(C.byref(spec), _p(cdf, C.c_uint32), _p(term, C.c_uint32), _p(unit, C.c_uint32),
                         _p(tf, C.c_uint32), _p(meta, C.c_uint8), _p(wc, C.c_uint32))
    return dict(term_id=term, unit_id=unit, tf=tf, meta=meta, word_count=wc, n_terms=spec.n_terms,
                n_docs=spec.doc_end - spec.doc_begin, n_articles=spec.article_end - spec.article_begin)


def gen_queries(seed: int, n_terms: int, cdf: np.ndarray, min_rank: int, n_queries: int, tpq: int) -> np.ndarray:
    out = np.empty((n_queries, tpq), np.uint32)
    rc = lib().orc_gen_queries(seed, n_terms, _p(cdf, C.c_uint32), min_rank, n_queries, tpq, _p(out, C.c_uint32))
    if rc:
        raise ValueError("vocabulary too small for the requested queries")
    return out


class OracleIndex:
    """Statistics job + scoring over one set of token rows."""

    def __init__(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None):
        self._keep = [np.ascontiguousarray(term_id, np.uint32), np.ascontiguousarray(unit_id, np.uint32),
                      np.ascontiguousarray(tf, np.uint32), np.ascontiguousarray(meta, np.uint8)]
        wc = None if word_count is None else np.ascontiguousarray(word_count, np.uint32)
        fl = None if doc_flags is None else np.ascontiguousarray(doc_flags, np.uint8)
        self.n_terms, self.n_docs, self.n_articles = int(n_terms), int(n_docs), int(n_articles)
        t, u, f, m = self._keep
        self._h = lib().orc_build(_p(t, C.c_uint32), _p(u, C.c_uint32), _p(f, C.c_uint32), _p(m, C.c_uint8),
                                  len(t), self.n_terms, self.n_docs, self.n_articles, _p(wc, C.c_uint32),
                                  _p(fl, C.c_uint8))
        if not self._h:
            raise ValueError("oracle: row out of range")

    def close(self):
        if self._h:
            lib().orc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, fn, n, dtype):
        ptr = getattr(lib(), fn)(self._h)
        if n == 0:
            return np.empty(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    @property
    def totals(self) -> Totals:
        t = Totals()
        lib().orc_get_totals(self._h, C.byref(t))
        return t

    @property
    def df(self):
        return self._arr("orc_df", self.n_terms, np.uint32)

    @property
    def idf(self):
        return self._arr("orc_idf", self.n_terms, np.float64)

    @property
    def term_rows(self):
        return self._arr("orc_term_rows", self.n_terms, np.uint32)

    @property
    def doc_len(self):
        return self._arr("orc_doc_len", self.n_docs, np.uint32)

    @property
    def row_ptr(self):
        return self._arr("orc_row_ptr", self.n_terms + 1, np.uint64)

    @property
    def news_ptr(self):
        return self._arr("orc_news_ptr", self.n_terms, np.uint64)

    @property
    def post_unit(self):
        return self._arr("orc_post_unit", int(self.totals.n_postings), np.uint32)

    @property
    def post_weight(self):
        return self._arr("orc_post_weight", int(self.totals.n_postings), np.uint32)

    @property
    def n_docs_stat(self) -> int:
        return int(lib().orc_n_docs_stat(self._h))

    @property
    def avgdl(self) -> float:
        return float(lib().orc_avgdl(self._h))

    def finalize(self, df_global=None, totals_global: Totals | None = None):
        dfg = None if df_global is None else np.ascontiguousarray(df_global, np.uint32)
        lib().orc_finalize(self._h, _p(dfg, C.c_uint32), None if totals_global is None else C.byref(totals_global))

    def set_idf(self, idf, present=None):
        idf = np.ascontiguousarray(idf, np.float64)
        pr = None if present is None else np.ascontiguousarray(present, np.uint8)
        lib().orc_set_idf(self._h, _p(idf, C.c_double), _p(pr, C.c_uint8))

    def search(self, terms, k, source=SRC_WEB, url_rank=None, unit_base=0, level="rows"):
        """Top-k of one query -> (unit[n], score[n], matched[n]); level 'rows' follows the SQL
        formula row by row, 'csr' scores the merged integer postings."""
        terms = np.ascontiguousarray(terms, np.uint32)
        ur = None if url_rank is None else np.ascontiguousarray(url_rank, np.uint32)
        ou = np.empty(k, np.uint32)
        osc = np.empty(k, np.float64)
        om = np.empty(k, np.uint32)
        fn = lib().orc_search_rows if level == "rows" else lib().orc_search_csr
        n = fn(self._h, source, _p(terms, C.c_uint32), len(terms), k, _p(ur, C.c_uint32), unit_base,
               _p(ou, C.c_uint32), _p(osc, C.c_double), _p(om, C.c_uint32))
        return ou[:n].copy(), osc[:n].copy(), om[:n].copy()
