"""Host-side glue between the SQL tables and one GPU index: rebuild (the statistics job) and
the scoring call that stands where the reference executes its search SQL.

`SearchEngine.score_web / score_news` return rows in exactly the tuple layout the reference's
statements yield (app/api/search_service.py:36 and :69-88), ordered token_score DESC, url ASC and
cut at the candidate limit, so the ranking code above them is unchanged.
"""
from __future__ import annotations

import threading

import numpy as np

from .native import SRC_NEWS, SRC_WEB
from .store import SqlStore, TokenExport, url_ranks


class SearchEngine:
    """One index + the dictionaries needed to talk in term strings and urls.

    index_factory() must return an object with the methods of native.Index (build, stats,
    set_idf, set_url_rank, search).  The default creates a native.Index on GPU `device`."""

    def __init__(self, store: SqlStore, device: int = 0, index_factory=None, collation_key=None):
        self.store = store
        self._factory = index_factory or (lambda: _native_index(device))
        self._collation_key = collation_key
        self._lock = threading.Lock()    # one index handle serves one call at a time
        self.index = None
        self.export: TokenExport | None = None

    # ---- the statistics job: replaces the SQL of app/batch/bm25_stats.py:9-27
    def rebuild(self, write_back: bool = True) -> dict:
        ex = self.store.export_tokens()
        ix = self._factory()
        ix.build(ex.term_id, ex.unit_id, ex.tf, ex.meta, ex.n_terms, ex.n_docs, ex.n_articles, ex.word_count,
                 ex.doc_flags, unit_major=ex.unit_major)
        st = ix.stats()
        if ex.n_docs:
            ix.set_url_rank(SRC_WEB, url_ranks(ex.doc_urls, self._collation_key))
        # articles are exported in url order already: their index is their rank
        if write_back:
            self.store.write_term_statistics(ex.terms, st["df"], st["idf"], st["avgdl"], st["term_rows"])
        with self._lock:
            old, self.index, self.export = self.index, ix, ex    # readers see the old index or the new one
        if old is not None and hasattr(old, "close"):
            old.close()
        return st

    def load_idf_from_table(self):
        """Scores with the idf values currently stored in term_statistics (COALESCE(ts.idf, 1.0)
        for terms without a row), as the reference's LEFT JOIN does between two runs of the job."""
        ex = self.export
        table = self.store.read_term_statistics()
        idf = np.ones(ex.n_terms, np.float64)
        present = np.zeros(ex.n_terms, np.uint8)
        for t, term in enumerate(ex.terms):
            if term in table:
                idf[t], present[t] = table[term], 1
        with self._lock:
            self.index.set_idf(idf, present)

    # ---- the scoring call: replaces cur.execute(SEARCH_SQL / NEWS_SEARCH_SQL, (terms, limit))
    def _score(self, source: int, terms, limit: int):
        if self.index is None:
            raise RuntimeError("SearchEngine.rebuild() has not run yet")
        with self._lock:
            ex, ix = self.export, self.index
            ids = [ex.term_index[t] for t in terms if t in ex.term_index]    # unknown terms match nothing
            if not ids:
                return ex, np.zeros(0, np.uint32), np.zeros(0), np.zeros(0, np.uint32)
            out = ix.search([ids], int(limit), source=source)
        n = int(out["count"][0])
        return ex, out["unit"][0, :n], out["score"][0, :n], out["matched"][0, :n]

    def score_web(self, terms, limit: int) -> list:
        """[(title, description, url, token_score, matched_terms)] (search_service.py:36)."""
        ex, unit, score, matched = self._score(SRC_WEB, terms, limit)
        ids = [int(ex.doc_ids[u]) for u in unit]
        meta = self.store.web_rows(ids)
        return [meta[i] + (float(s), int(m)) for i, s, m in zip(ids, score, matched) if i in meta]

    def score_web_pairs(self, terms, limit: int) -> list:
        """[(token_score, matched_terms)] ordered by score: the rows of the reference's fallback
        statement (search_service.py:42-66), which carries no text columns."""
        _, _, score, matched = self._score(SRC_WEB, terms, limit)
        return [(float(s), int(m)) for s, m in zip(score, matched)]

    def score_news(self, terms, limit: int) -> list:
        """17-column rows of the news statement (search_service.py:69-88): 15 article / feed
        columns, then token_score and matched_terms."""
        ex, unit, score, matched = self._score(SRC_NEWS, terms, limit)
        urls = [ex.article_urls[u] for u in unit]
        meta = self.store.news_rows(urls)
        return [meta[u] + (float(s), int(m)) for u, s, m in zip(urls, score, matched) if u in meta]


def _native_index(device):
    from .native import Index

    return Index(device)
