"""Host-side glue between the SQL tables and one GPU index: rebuild (the statistics job) and
the scoring call that stands where the reference executes its search SQL.

`SearchEngine.score_web / score_news` return rows in exactly the tuple layout the reference's
statements yield (app/api/search_service.py:36 and :69-88), ordered token_score DESC, url ASC and
cut at the candidate limit, so the ranking code above them is unchanged.

Lifecycle.  The reference runs the statistics job and the search API as separate processes
(docker-compose.yml:50,67) that meet in the database: its search statement reads the live `tokens`
table and the idf values the job last wrote.  Here every process holds its own index, a snapshot of
`tokens`:
  * the job's engine: `rebuild()` = export, build, write `term_statistics`;
  * the API's engine: builds itself on first use (`ensure_ready`), scores with the idf values of the
    table (`idf_source="table"`, i.e. COALESCE(ts.idf, 1.0) as the reference's LEFT JOIN), and
    `refresh_if_changed()` / `start_auto_refresh(interval_s)` rebuild it when the table's signature
    changes, i.e. after each run of the job.  Between two refreshes searches see the snapshot: rows the
    crawler wrote since are not visible (the staleness window is the refresh interval plus one export).
"""
from __future__ import annotations

import itertools
import threading

import numpy as np

from .native import SRC_NEWS, SRC_WEB

# A cap a deployment may put on the distinct indexed terms of one request (None: none).  The reference's
# `t.term = ANY($1)` (search_service.py:33) has no limit and neither has the library; a request beyond a cap that
# was set fails alone, not the batch it was coalesced into.
MAX_QUERY_TERMS = None
from .store import TEXT_DECODE_ERRORS, SqlStore, TokenExport, UrlRanker, _factorize, url_ranks


def _pinned_alloc():
    """Row arrays in page-locked memory when a GPU is present: the library then copies them at full
    PCIe rate without staging."""
    try:
        import torch

        if not torch.cuda.is_available():
            return None
    except Exception:  # noqa: BLE001
        return None
    kinds = {np.dtype(np.uint32): torch.int32, np.dtype(np.uint8): torch.uint8}

    def alloc(n, dtype):
        t = torch.empty(max(int(n), 1), dtype=kinds[np.dtype(dtype)], pin_memory=True)
        return t.numpy().view(dtype)[:n]

    return alloc


class SearchEngine:
    """One index + the dictionaries needed to talk in term strings and urls.

    index_factory() must return an object with the methods of native.Index (build, stats,
    set_idf, set_url_rank, search).  The default creates a native.Index on GPU `device`.
    idf_source: "fresh" scores with the idf of this build; "table" with the values stored in
    term_statistics (1.0 for a term without a row)."""

    def __init__(self, store: SqlStore, device: int = 0, index_factory=None, collation_key=None, idf_source: str = "fresh"):
        if idf_source not in ("fresh", "table"):
            raise ValueError("idf_source must be 'fresh' or 'table'")
        self.store = store
        self._factory = index_factory or (lambda: _native_index(device))
        self._native = index_factory is None
        self._collation_key = collation_key
        self._lock = threading.Lock()        # one index handle serves one call at a time
        self._build_lock = threading.Lock()  # one rebuild at a time
        self._ready_lock = threading.Lock()
        self.idf_source = idf_source
        self.index = None
        self.export: TokenExport | None = None
        self.table_version = None
        self._refresher = None
        self._stop = threading.Event()

    # ---- the statistics job: replaces the SQL of app/batch/bm25_stats.py:9-27
    def rebuild(self, write_back: bool = True) -> dict:
        with self._build_lock:
            ex = self.store.export_tokens(alloc=_pinned_alloc() if self._native else None, collation_key=self._collation_key)
            ix = self._factory()
            ix.build(ex.term_id, ex.unit_id, ex.tf, ex.meta, ex.n_terms, ex.n_docs, ex.n_articles, ex.word_count,
                     ex.doc_flags, unit_major=ex.unit_major)
            st = ix.stats()
            # docs and articles are numbered in url order by the export: a unit's index is its ORDER BY url rank, which
            # is the index's default tie-break key (no rank array, no gather per candidate)
            if ex.n_docs and not ex.url_ordered:
                ix.set_url_rank(SRC_WEB, url_ranks(ex.doc_urls, self._collation_key))
            if write_back:
                self.store.write_term_statistics(ex.terms, st["df"], st["idf"], st["avgdl"], st["term_rows"],
                                                 ex.extra_term_rows)
            version = self.store.term_statistics_version()
            if self.idf_source == "table":
                self._apply_table_idf(ix, ex)
            with self._lock:
                old, self.index, self.export = self.index, ix, ex    # readers see the old index or the new one
                self.table_version = version
            if old is not None and hasattr(old, "close"):
                old.close()
            return st

    # ---- incremental: what changed since the last build, instead of the whole table again
    COMPACT_UPDATE_FRACTION = 0.05  # merge the update segment into the base when it holds this share of the rows
    #                                 (every update sorts the update segment's rows again) ...
    COMPACT_DEAD_FRACTION = 0.25    # ... or when this share of the held rows is dead

    def update(self, doc_ids=(), article_urls=(), write_back: bool = True) -> dict:
        """Brings the index (and term_statistics) to the state a rebuild would give after the crawler replaced,
        removed or added the given documents (documents.id) and news articles (url): their current rows are
        read from the database, everything the index held for them is dropped (app/crawler/worker.py:229-239,
        app/batch/news_fetcher.py:303-311).  Only these units' rows cross to the GPU.

        The export's dictionaries (doc positions, terms, url ranks) are extended before the index takes the rows.  If
        anything fails between the two, export and index would describe different id spaces: the engine then rebuilds
        from the database (the state a rebuild gives is what an update has to reach anyway) and raises the failure.

        Where an update and a rebuild differ: token rows whose document or article row is gone (orphans) are dropped
        here and not counted, while a rebuild counts them (a tokens-only doc id is a doc that is not done and counts
        in df; rows of a vanished article go to the extra term rows).  The reference's writers delete a unit's tokens
        together with the unit, so the two agree on what they write; the next rebuild reconciles the rest."""
        self.ensure_ready()
        self._update_dirty = False
        try:
            return self._update(doc_ids, article_urls, write_back)
        except Exception:
            if self._update_dirty:
                self._pos_of = None
                self.rebuild(write_back=write_back)
            raise

    def update_pending(self, write_back: bool = True):
        """update() for whatever the change-capture triggers listed since the last call (SqlStore.install_change_capture):
        the statistics job without reading the whole `tokens` table and without being told what changed.  -> the
        statistics, or None when nothing was written.  A run that fails lists its units again."""
        docs, arts = self.store.drain_changes()
        if not docs and not arts:
            return None
        try:
            return self.update(docs, arts, write_back=write_back)
        except Exception:
            self.store.note_changes(docs, arts)
            raise

    def _update(self, doc_ids, article_urls, write_back):
        with self._build_lock:
            ex, ix = self.export, self.index
            data = self.store.export_units(ex, doc_ids, article_urls)
            if getattr(self, "_doc_pos", None) is None or self._pos_of is not ex:
                self._doc_pos = {int(d): i for i, d in enumerate(ex.doc_ids.tolist())}
                self._art_pos = {u: i for i, u in enumerate(ex.article_urls)}
                # ranks in url order, kept current unit by unit (a full sort of a million urls per update would
                # cost more than everything else in it)
                self._doc_ranker = UrlRanker(ex.doc_urls, self._collation_key, presorted=ex.url_ordered)
                self._art_ranker = UrlRanker(ex.article_urls, self._collation_key, presorted=self._collation_key is None)
                self._pos_of = ex
            doc_pos, art_pos = self._doc_pos, self._art_pos
            with self._lock:
                # ---- documents: position of each changed id (new ids go to the end), columns as they are now
                self._update_dirty = True
                n_old = len(ex.doc_ids)
                new_ids, new_wc, new_fl, new_urls = [], [], [], []
                set_wc, set_fl, url_changes = {}, {}, {}
                changed_docs = []
                for d in dict.fromkeys(int(x) for x in doc_ids):
                    row = data["docs"].get(d)
                    if d not in doc_pos:
                        if row is None:
                            continue                       # never indexed and not there now
                        doc_pos[d] = n_old + len(new_ids)
                        new_ids.append(d)
                        new_wc.append(0)
                        new_fl.append(0)
                        new_urls.append(None)
                    i = doc_pos[d]
                    changed_docs.append(i)
                    if row is None:                        # the document is gone
                        w_i, f_i, u_i = 0, 0, None
                    else:
                        url, w, status = row
                        w_i, f_i, u_i = int(w or 0), (1 if status == "done" else 0) | (2 if w is None else 0), url
                    if i >= n_old:
                        new_wc[i - n_old], new_fl[i - n_old], new_urls[i - n_old] = w_i, f_i, u_i
                    else:
                        set_wc[i], set_fl[i], url_changes[i] = w_i, f_i, u_i
                if new_ids:
                    ex.doc_ids = np.concatenate([ex.doc_ids, np.asarray(new_ids, np.int64)])
                    ex.word_count = np.concatenate([ex.word_count, np.asarray(new_wc, np.uint32)])
                    ex.doc_flags = np.concatenate([ex.doc_flags, np.asarray(new_fl, np.uint8)])
                if set_wc:
                    at = np.fromiter(set_wc, np.int64, len(set_wc))
                    ex.word_count[at] = np.fromiter(set_wc.values(), np.uint32, len(set_wc))
                    ex.doc_flags[at] = np.fromiter(set_fl.values(), np.uint8, len(set_fl))
                for i, u in url_changes.items():
                    ex.doc_urls[i] = u
                ex.doc_urls.extend(new_urls)
                self._doc_ranker.apply(url_changes, new_urls)
                changed_arts = []
                new_arts = []
                for u in dict.fromkeys(article_urls):
                    if u not in art_pos:
                        if u not in data["articles"]:
                            continue
                        art_pos[u] = len(ex.article_urls)
                        ex.article_urls.append(u)
                        new_arts.append(u)
                    changed_arts.append(art_pos[u])
                self._art_ranker.apply({}, new_arts)
                # ---- rows of these units, terms through the dictionary (new terms go to the end)
                ti = ex.term_index
                parts = []

                def convert(rows, unit_of, source):
                    """(key, term, field, frequency) rows -> columns; the Python work is per distinct key and term."""
                    if not rows:
                        return
                    a = np.array(rows, dtype=object)
                    keys, kinv = _distinct(a[:, 0])
                    pos = np.fromiter((unit_of(x) for x in keys), np.int64, len(keys))
                    unit = pos[kinv]
                    keep = unit >= 0
                    if not keep.all():
                        a, unit = a[keep], unit[keep]
                    if not len(unit):
                        return
                    tcodes, tuniq = _factorize(a[:, 1])
                    ids = np.fromiter((ti.setdefault(t, len(ti)) for t in tuniq), np.int64, len(tuniq))
                    parts.append((ids[tcodes].astype(np.uint32), unit.astype(np.uint32),
                                  np.maximum(a[:, 3].astype(np.int64), 0).astype(np.uint32),
                                  (source | ((a[:, 2].astype(np.int64) & 63) << 2)).astype(np.uint8)))

                convert(data["doc_rows"], lambda d: doc_pos.get(int(d), -1), 1)
                live_arts = data["articles"]
                convert(data["article_rows"], lambda u: art_pos.get(u, -1) if u in live_arts else -1, 2)
                t_l, u_l, f_l, m_l = ([np.concatenate([p[i] for p in parts]) for i in range(4)] if parts else
                                      [np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8)])
                if len(ti) > len(ex.terms):
                    ex.terms.extend(itertools.islice(ti, len(ex.terms), None))   # new terms, in the order they were given ids
                ix.update(np.asarray(t_l, np.uint32), np.asarray(u_l, np.uint32), np.asarray(f_l, np.uint32),
                          np.asarray(m_l, np.uint8), ex.n_terms, ex.n_docs, ex.n_articles, ex.word_count, ex.doc_flags,
                          changed_docs=changed_docs, changed_articles=changed_arts)
                self._update_dirty = False
                # explicit ranks only once a unit's index is no longer its place in url order (docs or articles were
                # appended, a url changed): until then ties break on the index and searches skip the rank gather
                # (while the units of the last rebuild still stand in url order, only the appended ones get a key from memory)
                for source, n_units, ranker in ((SRC_WEB, ex.n_docs, self._doc_ranker), (SRC_NEWS, ex.n_articles, self._art_ranker)):
                    if not n_units or ranker.identity:
                        continue
                    sparse = ranker.appended_keys() if hasattr(ix, "set_url_rank_appended") else None
                    if sparse is not None:
                        ix.set_url_rank_appended(source, *sparse)
                    else:
                        ix.set_url_rank(source, ranker.ranks())
                inf = ix.info() if hasattr(ix, "info") else None
                held = max(inf.n_rows + inf.n_dead_rows, 1) if inf is not None else 1
                if inf is not None and (inf.n_update_rows > self.COMPACT_UPDATE_FRACTION * held or
                                        inf.n_dead_rows > self.COMPACT_DEAD_FRACTION * held):
                    ix.compact()
                st = ix.stats()
            if write_back:
                self.store.write_term_statistics(ex.terms, st["df"], st["idf"], st["avgdl"], st["term_rows"], ex.extra_term_rows)
            self.table_version = self.store.term_statistics_version()
            if self.idf_source == "table":
                with self._lock:
                    self._apply_table_idf(ix, ex)
            return st

    def refresh(self) -> dict:
        """Rebuilds this process's index from the database without touching term_statistics."""
        return self.rebuild(write_back=False)

    def ensure_ready(self):
        """Builds the index on first use (a process that only searches never runs the job)."""
        if self.index is None:
            with self._ready_lock:
                if self.index is None:
                    self.refresh()

    def refresh_if_changed(self) -> bool:
        """Rebuilds when term_statistics differs from what this index was built beside (the job ran)."""
        if self.index is not None and self.store.term_statistics_version() == self.table_version:
            return False
        self.refresh()
        return True

    def start_auto_refresh(self, interval_s: float = 30.0):
        """Polls term_statistics every interval_s seconds in a daemon thread and rebuilds on change."""
        if self._refresher is not None:
            return
        self._stop.clear()

        def loop():
            while not self._stop.wait(interval_s):
                try:
                    self.refresh_if_changed()
                except Exception:  # noqa: BLE001 - keep serving the old snapshot; the next tick tries again
                    pass

        self._refresher = threading.Thread(target=loop, name="sebm25-refresh", daemon=True)
        self._refresher.start()

    def stop_auto_refresh(self):
        self._stop.set()
        if self._refresher is not None:
            self._refresher.join(timeout=5)
            self._refresher = None

    def _apply_table_idf(self, ix, ex):
        table = self.store.read_term_statistics()
        idf = np.ones(ex.n_terms, np.float64)
        present = np.zeros(ex.n_terms, np.uint8)
        for t, term in enumerate(ex.terms):
            v = table.get(term)
            if v is not None:
                idf[t], present[t] = v, 1
        ix.set_idf(idf, present)

    def load_idf_from_table(self):
        """Scores with the idf values currently stored in term_statistics (COALESCE(ts.idf, 1.0)
        for terms without a row), as the reference's LEFT JOIN does between two runs of the job."""
        with self._lock:
            self._apply_table_idf(self.index, self.export)

    # ---- the scoring call: replaces cur.execute(SEARCH_SQL / NEWS_SEARCH_SQL, (terms, limit))
    def _score_many(self, source: int, requests):
        """requests: [(terms, limit)] -> (export, [(unit, score, matched)]) from ONE batched search."""
        self.ensure_ready()
        with self._lock:
            ex, ix = self.export, self.index
            queries = [list(dict.fromkeys(ex.term_index[t] for t in terms if t in ex.term_index)) for terms, _ in requests]
            kmax = max([int(limit) for _, limit in requests] + [1])
            # unknown terms match nothing; a query beyond the deployment's cap fails alone, not the batch it came in
            cap = MAX_QUERY_TERMS if MAX_QUERY_TERMS is not None else float("inf")
            live = [i for i, q in enumerate(queries) if q and len(q) <= cap]
            out = ix.search([queries[i] for i in live], kmax, source=source) if live else None
        res = [(np.zeros(0, np.uint32), np.zeros(0), np.zeros(0, np.uint32))] * len(requests)
        for i, q in enumerate(queries):
            if len(q) > cap:
                res[i] = ValueError(f"query with {len(q)} distinct indexed terms: at most {MAX_QUERY_TERMS} are scored")
        for j, i in enumerate(live):
            # a result list cut at a smaller limit is a prefix of the list cut at a larger one
            n = min(int(out["count"][j]), int(requests[i][1]))
            res[i] = (out["unit"][j, :n], out["score"][j, :n], out["matched"][j, :n])
        return ex, res

    def _score(self, source: int, terms, limit: int):
        ex, res = self._score_many(source, [(terms, limit)])
        if isinstance(res[0], Exception):
            raise res[0]
        return (ex,) + res[0]

    def score_batch(self, requests) -> list:
        """requests: [(kind, terms, limit)], kind 'web' | 'news'.  One batched GPU search per kind and one
        metadata statement per kind for the whole batch.  -> per request ('rows', rows) in the layout of
        score_web / score_news, or ('pairs', [(token_score, matched_terms)]) for web requests when the row
        text cannot be decoded (the reference's fallback, search_service.py:321-338), or the exception of a
        request that cannot be scored (more distinct terms than the library takes)."""
        out = [None] * len(requests)
        for kind, source in (("web", SRC_WEB), ("news", SRC_NEWS)):
            idx = [i for i, r in enumerate(requests) if r[0] == kind]
            if not idx:
                continue
            ex, res = self._score_many(source, [(requests[i][1], requests[i][2]) for i in idx])
            failed = [(i, r) for i, r in zip(idx, res) if isinstance(r, Exception)]
            if failed:      # these requests get their exception, the others their rows
                for i, r in failed:
                    out[i] = r
                keep = [j for j, r in enumerate(res) if not isinstance(r, Exception)]
                idx, res = [idx[j] for j in keep], [res[j] for j in keep]
            if kind == "web":
                ids = [[int(ex.doc_ids[u]) for u in unit] for unit, _, _ in res]
                try:
                    meta = self.store.web_rows([i for row in ids for i in row])
                except TEXT_DECODE_ERRORS:
                    for i, (_, score, matched) in zip(idx, res):
                        out[i] = ("pairs", [(float(s), int(m)) for s, m in zip(score, matched)])
                    continue
                for i, row_ids, (_, score, matched) in zip(idx, ids, res):
                    out[i] = ("rows", [meta[d] + (float(s), int(m)) for d, s, m in zip(row_ids, score, matched) if d in meta])
            else:
                urls = [[ex.article_urls[u] for u in unit] for unit, _, _ in res]
                meta = self.store.news_rows([u for row in urls for u in row])
                for i, row_urls, (_, score, matched) in zip(idx, urls, res):
                    out[i] = ("rows", [meta[u] + (float(s), int(m)) for u, s, m in zip(row_urls, score, matched) if u in meta])
        return out

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


def _distinct(values):
    """-> (distinct values in order of first appearance, index of each value among them)."""
    codes, uniq = _factorize(values)
    return uniq, codes


def _native_index(device):
    from .native import Index

    return Index(device)
