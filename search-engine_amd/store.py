"""Export / import between the reference's SQL tables and the arrays the GPU index takes.

Input side: the `tokens` table (alembic/versions/0001_initial_schema.py:65-74 +
0007_unify_news_tokens.py:19-27; writers app/crawler/worker.py:229-239 and
app/batch/news_fetcher.py:303-311), `documents.word_count/status` (0001:42,45) and the url columns
used for ORDER BY ... url ASC.  Output side: the `term_statistics` table (0001:110-115), replaced
in one transaction exactly as app/batch/bm25_stats.py:15-27 does through app/common/db.py:27, so
unmodified readers keep working.

The export is one snapshot (a single REPEATABLE READ / deferred transaction, so rows the crawler
commits meanwhile are either all seen or not at all), streamed in chunks and converted column by
column with numpy: no per-row Python work.  The term dictionary is derived from the streamed rows
themselves.  Rows come in heap order (no ORDER BY: sorting 1e8+ rows inside the database costs far
more than the two extra radix passes the GPU needs for rows that are not unit-major).

`SqlStore` speaks DB-API 2.0 and is exercised here with the stdlib sqlite3 module; on PostgreSQL
(psycopg 3, `SqlStore.from_postgres_env`, import-gated: psycopg is absent from the build container)
the rows leave through `COPY (SELECT ...) TO STDOUT` and `term_statistics` is refilled through
`COPY ... FROM STDIN`.
"""
from __future__ import annotations

import io
import re
import itertools
import os
from dataclasses import dataclass, field

import numpy as np

try:  # optional: the fast COPY text reader and the native term dictionary (imported here, on the importing thread:
    # a first import of pyarrow from a worker thread of a process that holds torch has crashed the interpreter)
    import pyarrow as _pa
    import pyarrow.compute as _pc
    from pyarrow import csv as _pacsv
except ImportError:  # pragma: no cover - pyarrow is present in the build container
    _pa = _pc = _pacsv = None

TOKEN_COLUMNS = ("COALESCE(doc_id, -1), term, field, frequency, COALESCE(article_url, ''), COALESCE(source_type, 1)")

# what a failed text decode of row metadata looks like: the reference falls back to a statement without
# text columns on psycopg's CharacterNotInRepertoire (app/api/search_service.py:321)
TEXT_DECODE_ERRORS: tuple = (UnicodeError,)
try:  # pragma: no cover - psycopg is not in the build container
    import psycopg.errors as _pg_errors  # type: ignore

    TEXT_DECODE_ERRORS = (UnicodeError, _pg_errors.CharacterNotInRepertoire)
except Exception:  # noqa: BLE001
    pass


@dataclass
class TokenExport:
    """The `tokens` table as structure-of-arrays plus the host-side dictionaries."""
    term_id: np.ndarray      # uint32[R]
    unit_id: np.ndarray      # uint32[R]  dense doc index (web rows) / article index (news rows)
    tf: np.ndarray           # uint32[R]
    meta: np.ndarray         # uint8[R]   source_type | field << 2
    terms: list              # term_id -> TEXT term
    doc_ids: np.ndarray      # int64[n_docs]  dense doc index -> documents.id
    word_count: np.ndarray   # uint32[n_docs]
    doc_flags: np.ndarray    # uint8[n_docs]  1 = status 'done', 2 = word_count IS NULL
    doc_urls: list           # dense doc index -> url (None for a doc id that only `tokens` knows)
    article_urls: list       # article index -> url
    unit_major: bool = True
    term_index: dict = field(default_factory=dict)
    # rows that reach no unit (a news row whose article is gone): the statistics job still groups them by
    # term (bm25_stats.py:23-24 reads `tokens` alone), so their terms get a term_statistics row
    extra_term_rows: np.ndarray | None = None
    # dense doc indices were handed out in ascending url order: a doc's index is its ORDER BY url ASC rank,
    # so the index needs no separate rank array for web ties
    url_ordered: bool = False

    @property
    def n_terms(self):
        return max(len(self.terms), 1)

    @property
    def n_docs(self):
        return len(self.doc_ids)

    @property
    def n_articles(self):
        return len(self.article_urls)


def url_ranks(urls, key=None) -> np.ndarray:
    """Rank of each url in ascending order (the GPU's tie-break key).  `key` can reproduce a
    database collation; the default is code-point order (PostgreSQL "C" collation).  A unit without
    a url (None) ranks last."""
    urls = ["\U0010ffff" if u is None else u for u in urls]
    if key is None:
        order = np.argsort(np.array(urls, dtype="U"), kind="stable") if urls else np.zeros(0, np.int64)
    else:
        order = np.asarray(sorted(range(len(urls)), key=lambda i: key(urls[i])), np.int64)
    rank = np.empty(len(urls), np.uint32)
    rank[order] = np.arange(len(urls), dtype=np.uint32)
    return rank


class _Coded:
    """A string column as codes into its distinct values (what a dictionary-encoding parser hands over).  The
    distinct values are a Python list or, from the pyarrow parser, an arrow string array (`arrow`), turned into a
    list only when somebody asks for it."""

    __slots__ = ("codes", "_uniq", "arrow")

    def __init__(self, codes, uniq=None, arrow=None):
        self.codes = codes      # int64[n]
        self._uniq = uniq       # list of str, or None while only `arrow` holds the values
        self.arrow = arrow

    @property
    def uniq(self):
        if self._uniq is None:
            self._uniq = self.arrow.to_pylist()
        return self._uniq

    def __len__(self):
        return len(self.codes)


_COPY_ESCAPES = {"\\\\": "\\", "\\t": "\t", "\\n": "\n", "\\r": "\r", "\\b": "\b", "\\f": "\f", "\\v": "\v"}
_COPY_ESCAPE_RE = re.compile(r"\\[\\tnrbfv]")


def _copy_unescape(strings: list) -> list:
    """COPY text escapes backslash, tab, newline and a few control characters inside a value."""
    if not any("\\" in u for u in strings):
        return strings
    return [_COPY_ESCAPE_RE.sub(lambda m: _COPY_ESCAPES[m.group(0)], u) for u in strings]


class UrlRanker:
    """The rank of every unit in ascending url order (the tie-break key of a search), kept current while urls are
    appended, changed or dropped, without sorting everything again: the units that changed are taken out of the
    sorted order and put back where they belong (a binary search each), the rest keeps its place.  Same ranks as
    `url_ranks` over the whole list; a unit without a url ranks last."""

    def __init__(self, urls, key=None, presorted: bool = False):
        self.urls = list(urls)
        self._key = key
        n = len(self.urls)
        if presorted:                       # unit indices were handed out in url order
            self.order = np.arange(n, dtype=np.int64)
        else:
            self.order = np.argsort(url_ranks(self.urls, key), kind="stable").astype(np.int64)
        self.identity = presorted           # rank == unit index: the index needs no rank array
        # the first n_base units were numbered in url order; as long as none of them moves (a unit whose url is dropped keeps
        # its place: it has no postings left to rank), only the units appended since need a key of their own (appended_keys)
        self.n_base = n if presorted else 0

    def _k(self, i):
        u = self.urls[i]
        return (1, "", i) if u is None else (0, u if self._key is None else self._key(u), i)

    def apply(self, changed: dict, appended=()) -> bool:
        """changed: unit index -> new url (or None); appended: urls of new units (indices follow the existing ones).
        -> whether any rank changed."""
        moved = [i for i, u in changed.items() if self.urls[i] != u]
        for i in moved:
            if i < self.n_base and changed[i] is not None:
                self.n_base = 0             # a base unit goes somewhere else in url order: every unit needs its rank
            self.urls[i] = changed[i]
        first_new = len(self.urls)
        self.urls.extend(appended)
        moved += list(range(first_new, len(self.urls)))
        if not moved:
            return False
        old = np.asarray([i for i in moved if i < first_new], np.int64)
        rest = self.order[~np.isin(self.order, old)] if len(old) else self.order
        moved.sort(key=self._k)
        pos = np.empty(len(moved), np.int64)
        for j, i in enumerate(moved):       # first place in `rest` whose unit ranks behind unit i
            k, lo, hi = self._k(i), 0, len(rest)
            while lo < hi:
                mid = (lo + hi) // 2
                if self._k(int(rest[mid])) < k:
                    lo = mid + 1
                else:
                    hi = mid
            pos[j] = lo
        new_order = np.insert(rest, pos, np.asarray(moved, np.int64))
        if self.identity and not np.array_equal(new_order, np.arange(len(new_order))):
            self.identity = False
        self.order = new_order
        return True

    def ranks(self) -> np.ndarray:
        rank = np.empty(len(self.order), np.uint32)
        rank[self.order] = np.arange(len(self.order), dtype=np.uint32)
        return rank

    def appended_keys(self, max_shift: int = 8):
        """-> (n_base, shift, keys) when the first n_base units still stand in url order among themselves: tie-break
        keys that order all units as ranks() does, with (unit + 1) << shift for a base unit (no array needed) and keys[i]
        for unit n_base + i, a value between the keys of the base units its url falls between.  None when a base unit
        moved, or when more units fell into one gap than 2^shift - 1 keys."""
        n, nb = len(self.order), self.n_base
        if nb == 0:
            return None
        shift = max_shift
        while shift > 0 and ((nb + 1) << shift) > 0xFFFFFFFF:
            shift -= 1
        if shift == 0:
            return None
        keys = np.zeros(n - nb, np.uint32)
        gap, r = 0, 0                                   # base units seen so far, appended units since the last one
        for u in self.order.tolist():
            if u < nb:
                if self.urls[u] is None:
                    continue                            # dropped: stands at the end of the order, ranks nothing
                if u + 1 <= gap:
                    return None                         # base units out of order (cannot happen while n_base is kept)
                gap, r = u + 1, 0
            else:
                r += 1
                if r >= (1 << shift):
                    return None
                keys[u - nb] = (gap << shift) + r
        return nb, shift, keys


def _copy_in_text(terms, df, idf, avgdl, rows_per_block: int = 1 << 16):
    """COPY text-format lines for `term_statistics`, a block of rows at a time: floats as their shortest
    round-trip decimal (the server parses them back to the same float8), text escaped as COPY expects."""
    avg = repr(float(avgdl))
    esc = str.maketrans({"\\": "\\\\", "\t": "\\t", "\n": "\\n", "\r": "\\r"})
    for i in range(0, len(terms), rows_per_block):
        j = i + rows_per_block
        yield "".join(f"{t.translate(esc)}\t{d}\t{x!r}\t{avg}\n" for t, d, x in zip(terms[i:j], df[i:j], idf[i:j])).encode()


def _parse_copy_text(buf: bytes):
    """Columns of a block of `COPY ... TO STDOUT` text-format lines (tab-separated, one row per line).
    With pyarrow: its multi-threaded reader, the two text columns dictionary-encoded in native code, so that
    Python sees none of a block's strings (the assembler merges the block's dictionary into its own in arrow arrays):
    2.3x the rate of the pandas parser.  Without it: pandas."""
    if _pa is None:
        return _parse_copy_text_pandas(buf)
    pa, pc, pacsv = _pa, _pc, _pacsv
    names = ["d", "t", "f", "q", "a", "s"]
    tbl = pacsv.read_csv(
        pa.BufferReader(buf),
        read_options=pacsv.ReadOptions(column_names=names, use_threads=True, block_size=8 << 20),
        parse_options=pacsv.ParseOptions(delimiter="\t", quote_char=False, double_quote=False, escape_char=False,
                                         newlines_in_values=False),
        convert_options=pacsv.ConvertOptions(column_types={"d": pa.int64(), "t": pa.string(), "f": pa.int64(), "q": pa.int64(),
                                                           "a": pa.string(), "s": pa.int64()},
                                             null_values=[], strings_can_be_null=False, quoted_strings_can_be_null=False))

    def coded(col):
        enc = pc.dictionary_encode(tbl[col].combine_chunks())
        codes = enc.indices.to_numpy(zero_copy_only=False).astype(np.int64)
        values = enc.dictionary
        if len(values) and pc.any(pc.match_substring(values, "\\")).as_py():   # escapes are rare: only then through Python
            return _Coded(codes, _copy_unescape(values.to_pylist()))
        return _Coded(codes, arrow=values)

    num = lambda col: tbl[col].to_numpy()  # noqa: E731
    return (num("d"), coded("t"), num("f"), num("q"), coded("a"), num("s"))


def _parse_copy_text_pandas(buf: bytes):
    import pandas as pd

    df = pd.read_csv(io.BytesIO(buf), sep="\t", header=None, quoting=3, na_filter=False, engine="c",
                     dtype={0: np.int64, 1: str, 2: np.int64, 3: np.int64, 4: str, 5: np.int64}, encoding="utf-8")
    cols = []
    for c in (1, 4):
        codes, uniq = pd.factorize(df[c].to_numpy(dtype=object))
        cols.append(_Coded(codes.astype(np.int64), _copy_unescape(uniq.tolist())))
    return (df[0].to_numpy(), cols[0], df[2].to_numpy(), df[3].to_numpy(), cols[1], df[5].to_numpy())


def _factorize(strings):
    """-> (codes int64[n], distinct strings as a list): hash-based, one pass, no sort."""
    if isinstance(strings, _Coded):
        return strings.codes, strings.uniq
    import pandas as pd

    codes, uniq = pd.factorize(np.asarray(strings, dtype=object))
    return codes.astype(np.int64), uniq.tolist()


class _Assembler:
    """Turns streamed column chunks into the arrays of a TokenExport."""

    def __init__(self, doc_ids: np.ndarray, art_index: dict):
        self.doc_ids = doc_ids
        self.art_index = art_index
        self.term_index: dict = {}       # term -> provisional id (order of first appearance), when pyarrow is absent
        # with pyarrow: the distinct terms so far as one arrow string array, same ids
        self.term_arrow = _pa.array([], _pa.string()) if _pa is not None else None
        self.parts: list = []            # (term_id, unit, tf, meta)
        self.unknown_docs: list = []     # int64 arrays of doc ids that `documents` does not hold
        self.orphan_terms: list = []     # term ids of rows without a unit
        self.rows_in = 0

    def add(self, doc_id, term, fld, freq, art, st):
        n = len(doc_id)
        if n == 0:
            return
        self.rows_in += n
        doc_id = np.asarray(doc_id, np.int64)
        st = np.asarray(st, np.int64)
        # ---- terms: unique strings of the chunk, then one dictionary access per distinct term
        if self.term_arrow is not None:
            inv, ids = self._term_ids_arrow(term)
        else:
            inv, uniq = _factorize(term)
            ti = self.term_index
            ids = np.fromiter((ti.setdefault(t, len(ti)) for t in uniq), np.int64, len(uniq))
        term_id = ids[inv]
        news = st == 2
        unit = np.zeros(n, np.int64)
        keep = np.ones(n, bool)
        # ---- web rows: position of documents.id in the sorted id column; ids only `tokens` knows are kept
        # apart and given a (not 'done') doc entry at the end, so COUNT(DISTINCT doc_id) still sees them
        web = ~news
        if web.any():
            d = doc_id[web]
            pos = np.searchsorted(self.doc_ids, d)
            pos_c = np.minimum(pos, max(len(self.doc_ids) - 1, 0))
            known = (self.doc_ids[pos_c] == d) if len(self.doc_ids) else np.zeros(len(d), bool)
            u = np.where(known, pos_c, -1 - d)          # unknown: encoded doc id, resolved in finish()
            null_doc = d < 0                            # a web row with NULL doc_id matches nothing and counts nowhere
            unit[web] = u
            if (~known & ~null_doc).any():
                self.unknown_docs.append(np.unique(d[~known & ~null_doc]))
            if null_doc.any():
                k = keep[web]
                k[null_doc] = False
                keep[web] = k
                self.orphan_terms.append(term_id[web][null_doc])
        # ---- news rows: article index by url
        if news.any():
            if isinstance(art, _Coded):
                ia, ua = art.codes[news], art.uniq
            else:
                ia, ua = _factorize(np.asarray(art, dtype=object)[news])
            ai = self.art_index
            amap = np.fromiter((ai.get(x, -1) for x in ua), np.int64, len(ua))
            a_unit = amap[ia]
            unit[news] = a_unit
            gone = a_unit < 0                           # JOIN news_articles drops it (search_service.py:90)
            if gone.any():
                k = keep[news]
                k[gone] = False
                keep[news] = k
                self.orphan_terms.append(term_id[news][gone])
        tf = np.maximum(np.asarray(freq, np.int64), 0)
        meta = (np.where(news, 2, 1) | ((np.asarray(fld, np.int64) & 63) << 2)).astype(np.uint8)
        if not keep.all():
            term_id, unit, tf, meta = term_id[keep], unit[keep], tf[keep], meta[keep]
        self.parts.append((term_id, unit, tf.astype(np.uint32), meta))

    def _term_ids_arrow(self, term):
        """-> (codes of the chunk's rows, provisional id per code): the chunk's distinct terms are looked up in, and
        the new ones appended to, the arrow array of all distinct terms; no Python string is made."""
        pa, pc = _pa, _pc
        if isinstance(term, _Coded):
            inv = term.codes
            values = term.arrow if term.arrow is not None else pa.array(term.uniq, pa.string())
        else:
            inv, uniq = _factorize(term)
            values = pa.array(uniq, pa.string())
        known = self.term_arrow
        where = pc.index_in(values, value_set=known) if len(known) else pa.nulls(len(values), pa.int32())
        is_new = pc.is_null(where).to_numpy(zero_copy_only=False)
        ids = where.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64)
        n_new = int(is_new.sum())
        if n_new:
            ids[is_new] = len(known) + np.arange(n_new, dtype=np.int64)
            self.term_arrow = pa.concat_arrays([known, values.filter(pa.array(is_new))])
        return inv, ids

    def finish(self, docs_word_count, docs_flags, doc_urls, article_urls, alloc, collation_key=None) -> TokenExport:
        R = sum(len(p[0]) for p in self.parts)
        term_id = alloc(R, np.uint32)
        unit_id = alloc(R, np.uint32)
        tf = alloc(R, np.uint32)
        meta = alloc(R, np.uint8)
        # final term ids: rank of the term among the distinct terms (deterministic whatever the row order)
        if self.term_arrow is not None:
            terms_seen = self.term_arrow.to_pylist()
            order = _pc.sort_indices(self.term_arrow).to_numpy(zero_copy_only=False).astype(np.int64)   # UTF-8 byte order = code-point order
        else:
            terms_seen = list(self.term_index)
            order = np.argsort(np.array(terms_seen, dtype="U"), kind="stable") if terms_seen else np.zeros(0, np.int64)
        rank = np.empty(len(terms_seen), np.int64)
        rank[order] = np.arange(len(terms_seen))
        terms = [terms_seen[i] for i in order.tolist()]
        doc_ids, word_count, doc_flags = self.doc_ids, docs_word_count, docs_flags
        remap = None
        if self.unknown_docs:
            extra = np.unique(np.concatenate(self.unknown_docs))
            all_ids = np.union1d(doc_ids, extra)
            old_pos = np.searchsorted(all_ids, doc_ids)
            wc2 = np.zeros(len(all_ids), np.uint32)
            fl2 = np.zeros(len(all_ids), np.uint8)      # not 'done': counts in df, never scored, not in N
            wc2[old_pos], fl2[old_pos] = word_count, doc_flags
            urls2 = [None] * len(all_ids)
            for i, p in enumerate(old_pos.tolist()):
                urls2[p] = doc_urls[i]
            remap = (old_pos, all_ids)
            doc_ids, word_count, doc_flags, doc_urls = all_ids, wc2, fl2, urls2
        # dense doc index = rank of the doc's url (search ties break on url ASC: search_service.py:38)
        doc_urls = list(doc_urls)
        url_rank = url_ranks(doc_urls, collation_key).astype(np.int64)
        by_rank = np.argsort(url_rank, kind="stable")
        doc_ids, word_count, doc_flags = doc_ids[by_rank], word_count[by_rank], doc_flags[by_rank]
        doc_urls = [doc_urls[i] for i in by_rank.tolist()]
        o = 0
        for t_id, unit, tf_c, meta_c in self.parts:
            n = len(t_id)
            if n == 0:
                continue
            if remap is not None:
                web = (meta_c & 3) == 1
                uw = unit[web]
                known = uw >= 0
                uw2 = np.empty(len(uw), np.int64)
                uw2[known] = remap[0][uw[known]]
                uw2[~known] = np.searchsorted(remap[1], -1 - uw[~known])
                unit = unit.copy()
                unit[web] = uw2
            web = (meta_c & 3) == 1
            if web.any():
                if web.all():
                    unit = url_rank[unit]
                else:
                    unit = unit.copy()
                    unit[web] = url_rank[unit[web]]
            term_id[o:o + n] = rank[t_id]
            unit_id[o:o + n] = unit
            tf[o:o + n] = tf_c
            meta[o:o + n] = meta_c
            o += n
        key = ((meta & 3).astype(np.int64) << 32) | unit_id
        ordered = bool(R < 2 or (np.diff(key) >= 0).all())
        extra_rows = None
        if self.orphan_terms:
            extra_rows = np.bincount(rank[np.concatenate(self.orphan_terms)], minlength=max(len(terms), 1)).astype(np.int64)
        return TokenExport(term_id, unit_id, tf, meta, terms, doc_ids, word_count, doc_flags, doc_urls,
                           list(article_urls), ordered, {t: i for i, t in enumerate(terms)}, extra_rows, True)


class SqlStore:
    """DB-API access to the reference's tables.  `paramstyle`: '?' (sqlite3) or '%s' (psycopg, which also
    selects the PostgreSQL forms: REPEATABLE READ snapshot, COPY out / in, TRUNCATE)."""

    def __init__(self, connect, paramstyle="?", chunk_rows: int = 1 << 20):
        self._connect = connect
        self._p = paramstyle
        self.postgres = paramstyle == "%s"
        self.chunk_rows = chunk_rows

    @classmethod
    def from_postgres_env(cls):
        """Connection parameters from POSTGRES_* as app/common/db.py:12-19 reads them."""
        try:
            import psycopg  # type: ignore
        except ImportError as e:  # pragma: no cover - psycopg is not in the build container
            raise RuntimeError("psycopg is required to reach PostgreSQL") from e
        env = os.environ
        info = (f"postgresql://{env['POSTGRES_USER']}:{env['POSTGRES_PASSWORD']}@{env['POSTGRES_HOST']}:"
                f"{env['POSTGRES_PORT']}/{env['POSTGRES_DB']}")
        return cls(lambda: psycopg.connect(info), "%s")

    # ---- export
    def _begin_snapshot(self, cur):
        if self.postgres:
            # must be the first statement of the transaction psycopg opens implicitly
            cur.execute("SET TRANSACTION ISOLATION LEVEL REPEATABLE READ, READ ONLY")
        else:
            try:
                cur.execute("BEGIN")
            except Exception:  # noqa: BLE001 - a driver that already opened the transaction
                pass

    def _token_chunks(self, conn, cur, where="", params=()):
        """Yields (doc_id, term, field, frequency, article_url, source_type) column chunks in heap order."""
        sql = f"SELECT {TOKEN_COLUMNS} FROM tokens{where}"
        if self.postgres and hasattr(cur, "copy") and not params:
            yield from self._copy_chunks(cur, sql)
            return
        cur.execute(sql, tuple(params))
        while True:
            rows = cur.fetchmany(self.chunk_rows)
            if not rows:
                return
            a = np.array(rows, dtype=object)      # one pass over the driver's tuples; columns are views
            yield (a[:, 0].astype(np.int64), a[:, 1], a[:, 2].astype(np.int64), a[:, 3].astype(np.int64), a[:, 4],
                   a[:, 5].astype(np.int64))

    def _copy_chunks(self, cur, sql, block_bytes: int = 64 << 20):
        """Blocks of whole COPY lines, parsed by two worker threads (the native parser releases the interpreter
        lock) while this thread reads the socket and the caller assembles the block before; results in order."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        pend = []
        size = 0
        jobs: deque = deque()
        with ThreadPoolExecutor(max_workers=2) as pool:
            with cur.copy(f"COPY ({sql}) TO STDOUT") as cp:
                for data in cp:
                    b = bytes(data)
                    pend.append(b)
                    size += len(b)
                    if size >= block_bytes:
                        buf = b"".join(pend)
                        cut = buf.rfind(b"\n") + 1
                        if cut:
                            jobs.append(pool.submit(_parse_copy_text, buf[:cut]))
                        pend, size = [buf[cut:]], len(buf) - cut
                        while len(jobs) > 2:
                            yield jobs.popleft().result()
            buf = b"".join(pend)
            if buf:
                jobs.append(pool.submit(_parse_copy_text, buf))
            while jobs:
                yield jobs.popleft().result()

    def export_tokens(self, alloc=None, collation_key=None) -> TokenExport:
        """The whole `tokens` table + the document columns the job reads, as one snapshot.
        `alloc(n, dtype)` supplies the four row arrays (e.g. pinned host memory); default numpy.
        Dense doc indices follow ascending url order under `collation_key` (default: code points)."""
        alloc = alloc or (lambda n, dt: np.empty(n, dt))
        conn = self._connect()
        try:
            cur = conn.cursor()
            self._begin_snapshot(cur)
            cur.execute("SELECT id, url, word_count, status FROM documents ORDER BY id")
            docs = cur.fetchall()
            dc = list(zip(*docs)) if docs else [(), (), (), ()]
            doc_ids = np.array(dc[0], np.int64)
            wc_null = np.fromiter((w is None for w in dc[2]), bool, len(docs))
            word_count = np.fromiter((w or 0 for w in dc[2]), np.int64, len(docs)).astype(np.uint32)
            done = np.fromiter((str(s) == "done" for s in dc[3]), bool, len(docs))
            doc_flags = (done.astype(np.uint8) | (wc_null.astype(np.uint8) << 1)).astype(np.uint8)
            cur.execute("SELECT url FROM news_articles ORDER BY url")
            article_urls = [r[0] for r in cur.fetchall()]
            asm = _Assembler(doc_ids, {u: i for i, u in enumerate(article_urls)})
            for chunk in self._token_chunks(conn, cur):
                asm.add(*chunk)
            ex = asm.finish(word_count, doc_flags, list(dc[1]), article_urls, alloc, collation_key)
        finally:
            try:
                conn.rollback()
            except Exception:  # noqa: BLE001
                pass
            conn.close()
        return ex

    def export_units(self, ex: TokenExport, doc_ids=(), article_urls=()):
        """The current rows of a few web docs / articles (for an incremental update): -> dict with
        term strings and the raw columns, read in one snapshot."""
        out = dict(doc_rows=[], article_rows=[], docs={}, articles=set())
        conn = self._connect()
        try:
            cur = conn.cursor()
            self._begin_snapshot(cur)
            p = self._p
            doc_ids = [int(d) for d in doc_ids]
            article_urls = list(article_urls)
            for i in range(0, len(doc_ids), 500):
                part = doc_ids[i:i + 500]
                ph = ",".join([p] * len(part))
                cur.execute(f"SELECT id, url, word_count, status FROM documents WHERE id IN ({ph})", part)
                for r in cur.fetchall():
                    out["docs"][int(r[0])] = (r[1], r[2], str(r[3]))
                cur.execute(f"SELECT doc_id, term, field, frequency FROM tokens WHERE doc_id IN ({ph})", part)
                out["doc_rows"].extend(cur.fetchall())
            for i in range(0, len(article_urls), 500):
                part = article_urls[i:i + 500]
                ph = ",".join([p] * len(part))
                cur.execute(f"SELECT url FROM news_articles WHERE url IN ({ph})", part)
                out["articles"].update(r[0] for r in cur.fetchall())
                cur.execute(f"SELECT article_url, term, field, frequency FROM tokens WHERE source_type = 2 AND article_url IN ({ph})", part)
                out["article_rows"].extend(cur.fetchall())
        finally:
            try:
                conn.rollback()
            except Exception:  # noqa: BLE001
                pass
            conn.close()
        return out

    # ---- change capture: which units were written since the job last ran
    CHANGE_TABLE = "sebm25_changed_units"
    _SQLITE_CAPTURE = (
        "CREATE TABLE IF NOT EXISTS sebm25_changed_units (doc_id INTEGER, article_url TEXT)",
        "CREATE TRIGGER IF NOT EXISTS sebm25_tok_ins AFTER INSERT ON tokens BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NEW.doc_id, NEW.article_url); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_tok_del AFTER DELETE ON tokens BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (OLD.doc_id, OLD.article_url); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_tok_upd AFTER UPDATE ON tokens BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (OLD.doc_id, OLD.article_url), (NEW.doc_id, NEW.article_url); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_doc_ins AFTER INSERT ON documents BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NEW.id, NULL); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_doc_upd AFTER UPDATE ON documents BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NEW.id, NULL); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_doc_del AFTER DELETE ON documents BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (OLD.id, NULL); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_art_ins AFTER INSERT ON news_articles BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NULL, NEW.url); END",
        "CREATE TRIGGER IF NOT EXISTS sebm25_art_del AFTER DELETE ON news_articles BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NULL, OLD.url); END",
    )
    # PostgreSQL: statement-level triggers with transition tables, so a crawl that writes a few hundred token rows of one
    # document logs one row.  (This text has not run here: there is no PostgreSQL server in the build image.)
    _POSTGRES_CAPTURE = (
        "CREATE TABLE IF NOT EXISTS sebm25_changed_units (doc_id BIGINT, article_url TEXT)",
        "CREATE OR REPLACE FUNCTION sebm25_note_new_tokens() RETURNS trigger LANGUAGE plpgsql AS $$ BEGIN "
        "INSERT INTO sebm25_changed_units SELECT DISTINCT doc_id, article_url FROM changed; RETURN NULL; END $$",
        "CREATE OR REPLACE FUNCTION sebm25_note_doc() RETURNS trigger LANGUAGE plpgsql AS $$ BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (COALESCE(NEW.id, OLD.id), NULL); RETURN NULL; END $$",
        "CREATE OR REPLACE FUNCTION sebm25_note_article() RETURNS trigger LANGUAGE plpgsql AS $$ BEGIN "
        "INSERT INTO sebm25_changed_units VALUES (NULL, COALESCE(NEW.url, OLD.url)); RETURN NULL; END $$",
        "DROP TRIGGER IF EXISTS sebm25_tok_ins ON tokens",
        "CREATE TRIGGER sebm25_tok_ins AFTER INSERT ON tokens REFERENCING NEW TABLE AS changed "
        "FOR EACH STATEMENT EXECUTE FUNCTION sebm25_note_new_tokens()",
        "DROP TRIGGER IF EXISTS sebm25_tok_del ON tokens",
        "CREATE TRIGGER sebm25_tok_del AFTER DELETE ON tokens REFERENCING OLD TABLE AS changed "
        "FOR EACH STATEMENT EXECUTE FUNCTION sebm25_note_new_tokens()",
        "DROP TRIGGER IF EXISTS sebm25_tok_upd ON tokens",
        "CREATE TRIGGER sebm25_tok_upd AFTER UPDATE ON tokens REFERENCING NEW TABLE AS changed "
        "FOR EACH STATEMENT EXECUTE FUNCTION sebm25_note_new_tokens()",
        "DROP TRIGGER IF EXISTS sebm25_doc_any ON documents",
        "CREATE TRIGGER sebm25_doc_any AFTER INSERT OR UPDATE OR DELETE ON documents FOR EACH ROW EXECUTE FUNCTION sebm25_note_doc()",
        "DROP TRIGGER IF EXISTS sebm25_art_any ON news_articles",
        "CREATE TRIGGER sebm25_art_any AFTER INSERT OR DELETE ON news_articles FOR EACH ROW EXECUTE FUNCTION sebm25_note_article()",
    )

    def install_change_capture(self):
        """Adds (idempotently) a table that lists the units whose token rows, status, word count or url were written,
        kept current by triggers on tokens / documents / news_articles.  The reference's writers need no change: the
        crawler replaces a document's token rows inside its own transaction (app/crawler/worker.py:229-239) and the news
        fetcher an article's (app/batch/news_fetcher.py:303-311); the triggers see both."""
        conn = self._connect()
        try:
            cur = conn.cursor()
            for stmt in (self._POSTGRES_CAPTURE if self.postgres else self._SQLITE_CAPTURE):
                cur.execute(stmt)
            conn.commit()
        finally:
            conn.close()

    def drain_changes(self):
        """-> (doc ids, article urls) written since the last drain, and forgets them, in one transaction: a unit written
        while the drained ones are being processed is listed again for the next run."""
        conn = self._connect()
        try:
            cur = conn.cursor()
            if self.postgres:
                cur.execute("DELETE FROM sebm25_changed_units RETURNING doc_id, article_url")
                rows = cur.fetchall()
            else:
                try:
                    cur.execute("BEGIN IMMEDIATE")
                except Exception:  # noqa: BLE001 - a driver that already opened the transaction
                    pass
                cur.execute("SELECT doc_id, article_url FROM sebm25_changed_units")
                rows = cur.fetchall()
                cur.execute("DELETE FROM sebm25_changed_units")
            conn.commit()
        except Exception:
            conn.rollback()
            raise
        finally:
            conn.close()
        docs = list(dict.fromkeys(int(d) for d, _ in rows if d is not None))
        arts = list(dict.fromkeys(u for _, u in rows if u is not None))
        return docs, arts

    def note_changes(self, doc_ids=(), article_urls=()):
        """Lists units as changed (what a failed run puts back)."""
        conn = self._connect()
        try:
            cur = conn.cursor()
            p = self._p
            cur.executemany(f"INSERT INTO sebm25_changed_units VALUES ({p},{p})",
                            [(int(d), None) for d in doc_ids] + [(None, u) for u in article_urls])
            conn.commit()
        finally:
            conn.close()

    # ---- write back
    def write_term_statistics(self, terms, df, idf, avgdl, term_rows, extra_term_rows=None) -> int:
        """TRUNCATE + INSERT in one transaction: readers see the old table or the new one
        (bm25_stats.py:15-27 + db.py:27).  A row is written for every term that has token rows."""
        n = len(terms)
        rows_of = np.asarray(term_rows[:n], np.int64)
        if extra_term_rows is not None:
            rows_of = rows_of + np.asarray(extra_term_rows[:n], np.int64)
        sel = np.nonzero(rows_of > 0)[0]
        t_sel = [terms[i] for i in sel.tolist()]
        df_sel = np.asarray(df)[sel].astype(np.int64).tolist()
        idf_sel = np.asarray(idf, np.float64)[sel].tolist()
        avgdl = float(avgdl)
        conn = self._connect()
        try:
            cur = conn.cursor()
            if self.postgres:
                cur.execute("TRUNCATE term_statistics")
            else:
                cur.execute("DELETE FROM term_statistics")
            if self.postgres and hasattr(cur, "copy"):
                with cur.copy("COPY term_statistics (term, doc_frequency, idf, avg_doc_len) FROM STDIN") as cp:
                    for block in _copy_in_text(t_sel, df_sel, idf_sel, avgdl):
                        cp.write(block)
            else:
                p = self._p
                cur.executemany(f"INSERT INTO term_statistics(term, doc_frequency, idf, avg_doc_len) VALUES ({p},{p},{p},{p})",
                                zip(t_sel, df_sel, idf_sel, itertools.repeat(avgdl)))
            conn.commit()
            return len(t_sel)
        except Exception:
            conn.rollback()
            raise
        finally:
            conn.close()

    def read_term_statistics(self) -> dict:
        conn = self._connect()
        try:
            cur = conn.cursor()
            cur.execute("SELECT term, idf FROM term_statistics")
            return {r[0]: float(r[1]) for r in cur.fetchall()}
        finally:
            conn.close()

    def term_statistics_version(self):
        """A cheap signature of the table that changes whenever the statistics job has replaced it."""
        conn = self._connect()
        try:
            cur = conn.cursor()
            cur.execute("SELECT COUNT(*), COALESCE(SUM(doc_frequency), 0), COALESCE(MAX(avg_doc_len), 0), "
                        "COALESCE(SUM(idf), 0) FROM term_statistics")
            r = cur.fetchone()
            return tuple(float(x) for x in r)
        finally:
            conn.close()

    # ---- row metadata for result tuples
    def web_rows(self, doc_ids) -> dict:
        """documents.id -> (title, description, url)."""
        return self._by_key("SELECT id, title, COALESCE(description, ''), url FROM documents WHERE id IN ({})", doc_ids)

    NEWS_COLUMNS = ("na.title, COALESCE(na.description, ''), na.url, nf.feed_url, nf.home_url, nf.name, nf.link, "
                    "nf.image, nf.discovered_by_url, nf.last_published, nf.last_fetched, nf.next_fetch_at, "
                    "nf.publish_rate_per_hour, na.author, na.published_at")

    def news_rows(self, urls) -> dict:
        """news_articles.url -> the 15 leading columns of the reference's news query."""
        sql = (f"SELECT na.url, {self.NEWS_COLUMNS} FROM news_articles na "
               "LEFT JOIN news_feeds nf ON nf.feed_url = na.feed_url WHERE na.url IN ({})")
        return self._by_key(sql, urls)

    def _by_key(self, sql, keys) -> dict:
        keys = list(dict.fromkeys(keys))
        if not keys:
            return {}
        out = {}
        conn = self._connect()
        try:
            cur = conn.cursor()
            for i in range(0, len(keys), 500):
                chunk = keys[i:i + 500]
                cur.execute(sql.format(",".join([self._p] * len(chunk))), chunk)
                for r in cur.fetchall():
                    out[r[0]] = tuple(r[1:])
        finally:
            conn.close()
        return out
