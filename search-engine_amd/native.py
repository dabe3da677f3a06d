"""ctypes binding of libsebm25.so (the C ABI in include/sebm25.h).

The library is the product: every build / search call below runs hand-written gfx950 kernels.
There is no CPU fallback — if the shared object is missing or no GPU is visible the calls
raise.  Arrays may be numpy arrays (host memory) or torch tensors (host or ``cuda``); only
their addresses cross the ABI.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.realpath(__file__))
LIB_PATH = os.environ.get("SEBM25_LIB") or os.path.join(_HERE, "lib", "libsebm25.so")

SRC_WEB, SRC_NEWS = 1, 2
FIELD_TITLE, FIELD_DESCRIPTION, FIELD_BODY = 1, 2, 4
ROWS_UNIT_MAJOR = 1
DOC_DONE, DOC_WC_NULL = 1, 2
PAD_UNIT = 0xFFFFFFFF

_ERR_NAMES = {1: "INVALID", 2: "RANGE", 3: "HIP", 4: "NOMEM", 5: "STATE", 6: "ORDER", 7: "INTERNAL"}


def meta(source_type, field):
    """meta byte of a token row: source_type | field << 2."""
    return np.uint8(source_type | (field << 2))


class Sebm25Error(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"sebm25 error {code} ({_ERR_NAMES.get(code, '?')}): {message}")
        self.code = code


class Rows(C.Structure):
    _fields_ = [
        ("term_id", C.c_void_p), ("unit_id", C.c_void_p), ("tf", C.c_void_p), ("meta", C.c_void_p),
        ("n_rows", C.c_uint64), ("n_terms", C.c_uint32), ("n_docs", C.c_uint32), ("n_articles", C.c_uint32),
        ("word_count", C.c_void_p), ("doc_flags", C.c_void_p), ("flags", C.c_uint32),
    ]


class DocTotals(C.Structure):
    _fields_ = [("n_done", C.c_uint64), ("n_wc", C.c_uint64), ("sum_wc", C.c_uint64), ("n_postings", C.c_uint64)]


class IndexInfo(C.Structure):
    _fields_ = [
        ("n_rows", C.c_uint64), ("n_postings", C.c_uint64), ("n_web_postings", C.c_uint64),
        ("n_terms", C.c_uint32), ("n_docs", C.c_uint32), ("n_articles", C.c_uint32),
        ("term_bits", C.c_uint32), ("unit_bits", C.c_uint32), ("tf_bits", C.c_uint32), ("sort_passes", C.c_uint32),
        ("n_segments", C.c_uint32), ("n_dead_rows", C.c_uint64), ("n_update_rows", C.c_uint64),
        ("fused_postings", C.c_uint32),
    ]


class CorpusSpec(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_terms", C.c_uint32), ("doc_begin", C.c_uint32), ("doc_end", C.c_uint32),
        ("article_begin", C.c_uint32), ("article_end", C.c_uint32), ("zipf_s", C.c_double),
        ("body_min", C.c_uint32), ("body_max", C.c_uint32), ("title_len", C.c_uint32), ("desc_len", C.c_uint32),
    ]


# every symbol include/sebm25.h declares
EXPORTS = [
    "sebm25_create", "sebm25_destroy", "sebm25_last_error", "sebm25_set_stream", "sebm25_build_local",
    "sebm25_local_stats", "sebm25_finalize", "sebm25_build", "sebm25_update", "sebm25_update_local", "sebm25_compact", "sebm25_get_stats", "sebm25_get_doc_len",
    "sebm25_set_idf", "sebm25_set_url_rank", "sebm25_set_url_rank_appended", "sebm25_set_unit_base", "sebm25_get_info", "sebm25_get_postings",
    "sebm25_search", "sebm25_search_noalloc", "sebm25_merge_topk", "sebm25_get_timings", "sebm25_enable_timing", "sebm25_zipf_cdf",
    "sebm25_gen_corpus", "sebm25_gen_queries",
]

_lib = None


def lib():
    """Loads the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C search-engine_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  A process must run on ONE HIP
    # runtime: if this library pulled in the system runtime first, torch's later initialisation
    # finds no GPU.  Importing torch first makes the dynamic loader resolve this library's
    # libamdhip64.so.7 dependency to the copy torch has already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.sebm25_create.argtypes = [i32, C.POINTER(vp)]
    L.sebm25_destroy.argtypes = [vp]
    L.sebm25_destroy.restype = None
    L.sebm25_last_error.argtypes = [vp]
    L.sebm25_last_error.restype = C.c_char_p
    L.sebm25_set_stream.argtypes = [vp, vp]
    L.sebm25_build_local.argtypes = [vp, C.POINTER(Rows)]
    L.sebm25_local_stats.argtypes = [vp, vp, C.POINTER(DocTotals)]
    L.sebm25_finalize.argtypes = [vp, vp, C.POINTER(DocTotals)]
    L.sebm25_build.argtypes = [vp, C.POINTER(Rows)]
    L.sebm25_update.argtypes = [vp, C.POINTER(Rows), vp, u32, vp, u32]
    L.sebm25_update_local.argtypes = [vp, C.POINTER(Rows), vp, u32, vp, u32]
    L.sebm25_compact.argtypes = [vp]
    L.sebm25_get_stats.argtypes = [vp, vp, vp, vp, C.POINTER(u64), C.POINTER(C.c_double)]
    L.sebm25_get_doc_len.argtypes = [vp, vp]
    L.sebm25_set_idf.argtypes = [vp, vp, vp]
    L.sebm25_set_url_rank.argtypes = [vp, i32, vp]
    L.sebm25_set_url_rank_appended.argtypes = [vp, i32, u32, u32, vp, u32]
    L.sebm25_set_unit_base.argtypes = [vp, u32, u32]
    L.sebm25_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.sebm25_get_postings.argtypes = [vp, vp, vp, vp, vp]
    L.sebm25_search.argtypes = [vp, i32, vp, vp, u32, u32, vp, vp, vp, vp, vp]
    L.sebm25_search_noalloc.argtypes = [vp, i32, vp, vp, u32, u32, vp, vp, vp, vp, vp]
    L.sebm25_merge_topk.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sebm25_get_timings.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), u32, C.POINTER(u32)]
    L.sebm25_enable_timing.argtypes = [vp, i32]
    L.sebm25_zipf_cdf.argtypes = [u32, C.c_double, vp]
    L.sebm25_gen_corpus.argtypes = [vp, C.POINTER(CorpusSpec), C.POINTER(u64), vp, vp, vp, vp, vp]
    L.sebm25_gen_queries.argtypes = [u64, u32, C.c_double, u32, u32, u32, vp]
    for name in EXPORTS:
        if name not in ("sebm25_destroy", "sebm25_last_error"):
            getattr(L, name).restype = i32
    _lib = L
    return L


def _addr(x):
    """Address of a numpy array / torch tensor / None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    if isinstance(x, int):
        return x
    raise TypeError(f"cannot take the address of {type(x)!r}")


def _check_dtype(x, name, np_dtype):
    if x is None:
        return
    want = np.dtype(np_dtype)
    if isinstance(x, np.ndarray):
        ok = x.dtype == want
    elif hasattr(x, "data_ptr"):
        # torch has no uint32: int32 storage with the same bits is accepted for 32-bit columns
        ok = x.element_size() == want.itemsize
    else:
        ok = True
    if not ok:
        raise TypeError(f"{name}: expected {want.itemsize}-byte elements ({want})")


def zipf_cdf(n_terms: int, zipf_s: float = 1.0) -> np.ndarray:
    cdf = np.empty(n_terms, np.uint32)
    rc = lib().sebm25_zipf_cdf(n_terms, zipf_s, cdf.ctypes.data)
    if rc:
        raise Sebm25Error(rc, lib().sebm25_last_error(None).decode())
    return cdf


def gen_queries(seed: int, n_terms: int, n_queries: int, terms_per_query: int = 3, min_rank: int = 100,
                zipf_s: float = 1.0) -> np.ndarray:
    out = np.empty((n_queries, terms_per_query), np.uint32)
    rc = lib().sebm25_gen_queries(seed, n_terms, zipf_s, min_rank, n_queries, terms_per_query, out.ctypes.data)
    if rc:
        raise Sebm25Error(rc, lib().sebm25_last_error(None).decode())
    return out


class Index:
    """One GPU-resident inverted index + its term statistics (one handle of the C ABI)."""

    def __init__(self, device: int = 0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.sebm25_create(device, C.byref(h))
        if rc:
            raise Sebm25Error(rc, self._L.sebm25_last_error(None).decode())
        self._h = h
        self.device = device
        self.n_terms = self.n_docs = self.n_articles = 0

    # -- plumbing
    def _check(self, rc):
        if rc:
            raise Sebm25Error(rc, self._L.sebm25_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.sebm25_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle: int | None):
        self._check(self._L.sebm25_set_stream(self._h, stream_handle))

    def enable_timing(self, on: bool = True):
        self._check(self._L.sebm25_enable_timing(self._h, 1 if on else 0))

    def timings(self) -> list[tuple[str, float]]:
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = C.c_uint32()
        self._check(self._L.sebm25_get_timings(self._h, names, ms, cap, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, cap))]

    # -- build
    def _rows(self, term_id, unit_id, tf, meta_, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major):
        for x, nm, dt in ((term_id, "term_id", np.uint32), (unit_id, "unit_id", np.uint32), (tf, "tf", np.uint32),
                          (meta_, "meta", np.uint8), (word_count, "word_count", np.uint32),
                          (doc_flags, "doc_flags", np.uint8)):
            _check_dtype(x, nm, dt)
        n = int(term_id.shape[0]) if term_id is not None else 0
        for x in (unit_id, tf, meta_):
            if (int(x.shape[0]) if x is not None else 0) != n:
                raise ValueError("row arrays must have equal length")
        return Rows(_addr(term_id), _addr(unit_id), _addr(tf), _addr(meta_), n, int(n_terms), int(n_docs),
                    int(n_articles), _addr(word_count), _addr(doc_flags), ROWS_UNIT_MAJOR if unit_major else 0)

    def _built(self, r: Rows):
        # sizes of the output arrays of stats() / doc_len() / postings(): only what the library accepted
        self.n_terms, self.n_docs, self.n_articles = int(r.n_terms), int(r.n_docs), int(r.n_articles)

    def build(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
              unit_major=True):
        """The whole statistics job on one GPU (sort, CSR, df, totals, idf)."""
        r = self._rows(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major)
        self._check(self._L.sebm25_build(self._h, C.byref(r)))
        self._built(r)

    def build_local(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
                    unit_major=True):
        """This shard's part of the job; follow with local_stats -> (sum across shards) -> finalize."""
        r = self._rows(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major)
        self._check(self._L.sebm25_build_local(self._h, C.byref(r)))
        self._built(r)

    def update(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None,
               changed_docs=None, changed_articles=None, unit_major=False, local=False):
        """Incremental update (sebm25_update): the listed units lose every row the index holds for them and get
        the rows given here; n_terms / n_docs / n_articles and word_count / doc_flags describe the tables as they
        are now.  local=True leaves the statistics to a following local_stats / finalize (sharded indexes)."""
        r = self._rows(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major)
        cd = np.ascontiguousarray(changed_docs if changed_docs is not None else [], np.uint32)
        ca = np.ascontiguousarray(changed_articles if changed_articles is not None else [], np.uint32)
        fn = self._L.sebm25_update_local if local else self._L.sebm25_update
        self._check(fn(self._h, C.byref(r), cd.ctypes.data if len(cd) else None, len(cd),
                       ca.ctypes.data if len(ca) else None, len(ca)))
        self._built(r)

    def compact(self):
        """Folds the update segments into one (sebm25_compact); results do not change."""
        self._check(self._L.sebm25_compact(self._h))

    def local_stats(self, df_out=None):
        """-> (df, DocTotals).  df_out may be a preallocated numpy array / torch tensor."""
        if df_out is None:
            df_out = np.empty(self.n_terms, np.uint32)
        t = DocTotals()
        self._check(self._L.sebm25_local_stats(self._h, _addr(df_out), C.byref(t)))
        return df_out, t

    def finalize(self, df_global=None, totals_global: DocTotals | None = None):
        self._check(self._L.sebm25_finalize(self._h, _addr(df_global),
                                            None if totals_global is None else C.byref(totals_global)))

    def stats(self):
        """-> dict(df, idf, term_rows, n_docs, avgdl): the content of term_statistics."""
        df = np.empty(self.n_terms, np.uint32)
        idf = np.empty(self.n_terms, np.float64)
        tr = np.empty(self.n_terms, np.uint32)
        n = C.c_uint64()
        avg = C.c_double()
        self._check(self._L.sebm25_get_stats(self._h, df.ctypes.data, idf.ctypes.data, tr.ctypes.data, C.byref(n),
                                             C.byref(avg)))
        return dict(df=df, idf=idf, term_rows=tr, n_docs=int(n.value), avgdl=float(avg.value))

    def doc_len(self) -> np.ndarray:
        out = np.empty(self.n_docs, np.uint32)
        self._check(self._L.sebm25_get_doc_len(self._h, out.ctypes.data))
        return out

    def info(self) -> IndexInfo:
        i = IndexInfo()
        self._check(self._L.sebm25_get_info(self._h, C.byref(i)))
        return i

    def postings(self):
        """-> (row_ptr[V+1], news_ptr[V], post_unit[P], post_weight_tenths[P]) as numpy arrays."""
        i = self.info()
        rp = np.empty(self.n_terms + 1, np.uint64)
        npz = np.empty(self.n_terms, np.uint64)
        pu = np.empty(i.n_postings, np.uint32)
        pw = np.empty(i.n_postings, np.uint32)
        self._check(self._L.sebm25_get_postings(self._h, rp.ctypes.data, npz.ctypes.data, pu.ctypes.data, pw.ctypes.data))
        return rp, npz, pu, pw

    def set_idf(self, idf, present=None):
        _check_dtype(idf, "idf", np.float64)
        _check_dtype(present, "present", np.uint8)
        self._check(self._L.sebm25_set_idf(self._h, _addr(idf), _addr(present)))

    def set_url_rank(self, source: int, rank):
        _check_dtype(rank, "rank", np.uint32)
        self._check(self._L.sebm25_set_url_rank(self._h, source, _addr(rank)))

    def set_url_rank_appended(self, source: int, n_base: int, shift: int, keys):
        """Tie-break keys for an index whose first n_base units are numbered in url order: (unit + 1) << shift for
        those, keys[i] for unit n_base + i (sebm25_set_url_rank_appended)."""
        _check_dtype(keys, "keys", np.uint32)
        n_keys = int(keys.numel() if hasattr(keys, "numel") else len(keys))
        self._check(self._L.sebm25_set_url_rank_appended(self._h, source, int(n_base), int(shift), _addr(keys), n_keys))

    def set_unit_base(self, web_base: int = 0, news_base: int = 0):
        self._check(self._L.sebm25_set_unit_base(self._h, web_base, news_base))

    # -- search
    def search_noalloc(self, queries, k: int, out, source: int = SRC_WEB):
        """`search` through the entry point that allocates nothing (sebm25_search_noalloc): `out`
        must hold preallocated device tensors (unit, score, matched, count and optionally tie);
        at most 1024 queries and k <= 256 per call."""
        return self.search(queries, k, source=source, out=out, _entry="sebm25_search_noalloc")

    def search(self, queries, k: int, source: int = SRC_WEB, out=None, want_tie: bool = False, _entry="sebm25_search"):
        """Batched top-k.  ``queries``: list of term-id sequences, or a 2-D uint32 array (one
        query per row).  -> dict(unit[Q,k], score[Q,k], matched[Q,k], count[Q], tie[Q,k]?).
        ``out`` may supply preallocated arrays/tensors under the same keys."""
        if isinstance(queries, np.ndarray) and queries.ndim == 2:
            q_terms = np.ascontiguousarray(queries, np.uint32).reshape(-1)
            nq, t = queries.shape
            q_off = np.arange(0, (nq + 1) * t, t, dtype=np.uint64)
        else:
            nq = len(queries)
            lens = np.fromiter((len(q) for q in queries), np.uint64, nq)
            q_off = np.zeros(nq + 1, np.uint64)
            np.cumsum(lens, out=q_off[1:])
            q_terms = (np.concatenate([np.asarray(q, np.uint32).reshape(-1) for q in queries]).astype(np.uint32)
                       if nq and q_off[-1] else np.zeros(0, np.uint32))
        if out is None:
            out = dict(unit=np.empty((nq, k), np.uint32), score=np.empty((nq, k), np.float64),
                       matched=np.empty((nq, k), np.uint32), count=np.empty(nq, np.uint32))
            if want_tie:
                out["tie"] = np.empty((nq, k), np.uint32)
        self._check(getattr(self._L, _entry)(self._h, source, q_terms.ctypes.data if len(q_terms) else None,
                                          q_off.ctypes.data, nq, k, _addr(out["unit"]), _addr(out["score"]),
                                          _addr(out["matched"]), _addr(out.get("tie")), _addr(out["count"])))
        return out

    def merge_topk(self, n_lists, n_queries, k, unit, score, matched, tie, count, out=None):
        if out is None:
            out = dict(unit=np.empty((n_queries, k), np.uint32), score=np.empty((n_queries, k), np.float64),
                       matched=np.empty((n_queries, k), np.uint32), tie=np.empty((n_queries, k), np.uint32),
                       count=np.empty(n_queries, np.uint32))
        self._check(self._L.sebm25_merge_topk(self._h, n_lists, n_queries, k, _addr(unit), _addr(score), _addr(matched),
                                              _addr(tie), _addr(count), _addr(out["unit"]), _addr(out["score"]),
                                              _addr(out["matched"]), _addr(out.get("tie")), _addr(out["count"])))
        return out

    # -- synthetic corpus
    def gen_corpus_rows(self, spec: CorpusSpec) -> int:
        n = C.c_uint64()
        self._check(self._L.sebm25_gen_corpus(self._h, C.byref(spec), C.byref(n), None, None, None, None, None))
        return int(n.value)

    def gen_corpus(self, spec: CorpusSpec, term_id, unit_id, tf, meta_, word_count=None) -> int:
        n = C.c_uint64()
        self._check(self._L.sebm25_gen_corpus(self._h, C.byref(spec), C.byref(n), _addr(term_id), _addr(unit_id),
                                              _addr(tf), _addr(meta_), _addr(word_count)))
        return int(n.value)


def make_spec(seed=20260227, n_terms=500_000, n_docs=1_000_000, n_articles=0, doc_begin=0, article_begin=0,
              zipf_s=1.0, body_min=100, body_max=500, title_len=8, desc_len=24) -> CorpusSpec:
    return CorpusSpec(seed, n_terms, doc_begin, doc_begin + n_docs, article_begin, article_begin + n_articles, zipf_s,
                      body_min, body_max, title_len, desc_len)
