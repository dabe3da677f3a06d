"""Export / import between the reference's SQL tables and the arrays the GPU index takes.

Input side: the `tokens` table (alembic/versions/0001_initial_schema.py:65-74 +
0007_unify_news_tokens.py:19-27), `documents.word_count/status` (0001:42,45) and the url columns
used for ORDER BY ... url ASC.  Output side: the `term_statistics` table (0001:110-115), replaced
in one transaction exactly as app/batch/bm25_stats.py:15-27 does through app/common/db.py:27, so
unmodified readers keep working.

`SqlStore` speaks DB-API 2.0 and is exercised here with the stdlib sqlite3 module; the same
statements run on PostgreSQL through psycopg (`SqlStore.from_postgres_env`, which needs psycopg
and is therefore import-gated: it is absent from the build container).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np


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
    doc_urls: list           # dense doc index -> url
    article_urls: list       # article index -> url
    unit_major: bool = True
    term_index: dict = field(default_factory=dict)

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
    database collation; the default is code-point order (PostgreSQL "C" collation)."""
    order = sorted(range(len(urls)), key=(lambda i: key(urls[i])) if key else (lambda i: urls[i]))
    rank = np.empty(len(urls), np.uint32)
    rank[np.asarray(order, np.int64)] = np.arange(len(urls), dtype=np.uint32)
    return rank


class SqlStore:
    """DB-API access to the reference's tables.  `paramstyle`: '?' (sqlite3) or '%s' (psycopg)."""

    def __init__(self, connect, paramstyle="?"):
        self._connect = connect
        self._p = paramstyle

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
    def export_tokens(self) -> TokenExport:
        conn = self._connect()
        try:
            cur = conn.cursor()
            cur.execute("SELECT id, url, word_count, status FROM documents ORDER BY id")
            docs = cur.fetchall()
            doc_ids = np.array([d[0] for d in docs], np.int64)
            doc_index = {int(d[0]): i for i, d in enumerate(docs)}
            word_count = np.array([d[2] or 0 for d in docs], np.uint32)
            doc_flags = np.array([(1 if str(d[3]) == "done" else 0) | (2 if d[2] is None else 0) for d in docs], np.uint8)
            cur.execute("SELECT url FROM news_articles ORDER BY url")
            article_urls = [r[0] for r in cur.fetchall()]
            art_index = {u: i for i, u in enumerate(article_urls)}
            cur.execute("SELECT DISTINCT term FROM tokens ORDER BY term")
            terms = [r[0] for r in cur.fetchall()]
            term_index = {t: i for i, t in enumerate(terms)}
            # unit-major order lets the build sort on term alone: web rows by doc, then news rows
            cur.execute("SELECT doc_id, term, field, frequency, article_url, source_type FROM tokens "
                        "ORDER BY source_type, doc_id, article_url")
            rows = cur.fetchall()
        finally:
            conn.close()
        n = len(rows)
        term_id = np.empty(n, np.uint32)
        unit_id = np.empty(n, np.uint32)
        tf = np.empty(n, np.uint32)
        meta = np.empty(n, np.uint8)
        keep = np.ones(n, bool)
        for i, (doc_id, term, fld, freq, art, st) in enumerate(rows):
            term_id[i] = term_index[term]
            tf[i] = max(int(freq), 0)
            st = int(st or 1)
            meta[i] = st | ((int(fld) & 63) << 2)
            if st == 2:
                if art in art_index:
                    unit_id[i] = art_index[art]
                else:
                    keep[i] = False          # JOIN news_articles drops it (search_service.py:90)
            elif doc_id is not None and int(doc_id) in doc_index:
                unit_id[i] = doc_index[int(doc_id)]
            else:
                keep[i] = False
        if not keep.all():
            term_id, unit_id, tf, meta = term_id[keep], unit_id[keep], tf[keep], meta[keep]
        key = ((meta & 3).astype(np.int64) << 32) | unit_id
        return TokenExport(term_id, unit_id, tf, meta, terms, doc_ids, word_count, doc_flags,
                           [d[1] for d in docs], article_urls, bool((np.diff(key) >= 0).all()), term_index)

    # ---- write back
    def write_term_statistics(self, terms, df, idf, avgdl, term_rows) -> int:
        """TRUNCATE + INSERT in one transaction: readers see the old table or the new one."""
        conn = self._connect()
        try:
            cur = conn.cursor()
            cur.execute("DELETE FROM term_statistics")
            p = self._p
            rows = [(terms[t], int(df[t]), float(idf[t]), float(avgdl)) for t in range(len(terms)) if term_rows[t] > 0]
            cur.executemany(f"INSERT INTO term_statistics(term, doc_frequency, idf, avg_doc_len) VALUES ({p},{p},{p},{p})", rows)
            conn.commit()
            return len(rows)
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
        keys = list(keys)
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
