"""Host-side mirror of the reference interface: tokenizer, query context, intent re-rank, the
`run()` job and the `perform_*_search` entry points, end to end over a sqlite database that has
the reference's tables.  On CPU the per-index engine is the oracle behind native.Index's
interface (test stand-in); the gpu-marked test runs the same flow on the real GPU index."""
import asyncio
import math
import sqlite3

import numpy as np
import pytest

from oracle import oracle as O
from search_engine_amd.api import search_service as ss
from search_engine_amd.batch import bm25_stats
from search_engine_amd.engine import SearchEngine
from search_engine_amd.store import SqlStore, url_ranks
from search_engine_amd.tokenizer import porter_stem, tokenize

DDL = """
CREATE TABLE documents (id INTEGER PRIMARY KEY, url TEXT UNIQUE NOT NULL, title TEXT, description TEXT, content TEXT,
                        word_count INT, status TEXT DEFAULT 'done');
CREATE TABLE news_feeds (feed_url TEXT PRIMARY KEY, home_url TEXT, name TEXT, link TEXT, image TEXT, discovered_by_url TEXT,
                         last_published TEXT, last_fetched TEXT, next_fetch_at TEXT, publish_rate_per_hour REAL);
CREATE TABLE news_articles (url TEXT PRIMARY KEY, feed_url TEXT NOT NULL, title TEXT, description TEXT, image TEXT,
                            content TEXT, author TEXT, published_at TEXT);
CREATE TABLE tokens (doc_id INTEGER, term TEXT NOT NULL, field INTEGER NOT NULL, frequency INT NOT NULL,
                     article_url TEXT, source_type INTEGER DEFAULT 1);
CREATE TABLE term_statistics (term TEXT PRIMARY KEY, doc_frequency BIGINT NOT NULL, idf DOUBLE PRECISION NOT NULL,
                              avg_doc_len DOUBLE PRECISION NOT NULL);
"""

PAGES = [
    (11, "https://gpu.example/radix-sort", "Radix sort on GPUs", "Sorting keys with radix passes", "radix sort gpu keys passes histogram scatter radix"),
    (12, "https://gpu.example/bm25", "BM25 ranking", "Ranking documents with BM25", "ranking documents bm25 idf term frequency ranking"),
    (15, "https://db.example/postgres-sort", "Sorting in PostgreSQL", None, "postgres sort order by limit sorting rows"),
    (20, "https://news.example/archive", "Archive", "Old stories about sorting", "archive stories"),
    (21, "https://gpu.example/draft", "Radix draft", "unfinished", "radix radix radix"),
]
ARTICLES = [
    ("https://news.example/a1", "https://news.example/feed", "GPU radix sort record", "A new record", "radix sort record gpu", "Ann", "2026-02-01T10:00:00"),
    ("https://news.example/a2", "https://news.example/feed", "Ranking functions", "", "bm25 ranking functions", None, None),
]


def make_db(path):
    db = sqlite3.connect(path)
    db.executescript(DDL)
    db.execute("INSERT INTO news_feeds VALUES ('https://news.example/feed','https://news.example','Example News',NULL,NULL,NULL,"
               "'2026-02-01T10:00:00',NULL,NULL,0.5)")
    for doc_id, url, title, desc, content in PAGES:
        status = "processing" if doc_id == 21 else "done"
        db.execute("INSERT INTO documents VALUES (?,?,?,?,?,?,?)", (doc_id, url, title, desc, content, len(content.split()), status))
        for field, text in ((1, title), (2, desc), (4, content)):          # worker.py:192-194, :230-239
            for term, freq in tokenize(text or "").items():
                db.execute("INSERT INTO tokens VALUES (?,?,?,?,NULL,1)", (doc_id, term, field, freq))
    for url, feed, title, desc, content, author, pub in ARTICLES:
        db.execute("INSERT INTO news_articles VALUES (?,?,?,?,NULL,?,?,?)", (url, feed, title, desc, content, author, pub))
        for term, freq in tokenize(" ".join([title, desc, content])).items():   # news_fetcher.py:298-311
            db.execute("INSERT INTO tokens VALUES (NULL,?,4,?,?,2)", (term, freq, url))
    db.commit()
    db.close()


class OracleBackedIndex:
    """native.Index look-alike on the CPU oracle: lets the host logic run without a GPU."""

    def build(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major=True):
        self.rows = [np.array(term_id, np.uint32), np.array(unit_id, np.uint32), np.array(tf, np.uint32), np.array(meta, np.uint8)]
        order = np.lexsort((self.rows[1], self.rows[3] & 3))        # the oracle takes unit-major rows
        self.rows = [a[order] for a in self.rows]
        self.orc = O.OracleIndex(*self.rows, n_terms, n_docs, n_articles, word_count, doc_flags)
        self.rank = {1: None, 2: None}

    def stats(self):
        o = self.orc
        return dict(df=o.df, idf=o.idf, term_rows=o.term_rows, n_docs=o.n_docs_stat, avgdl=o.avgdl)

    def set_url_rank(self, source, rank):
        self.rank[source] = rank

    def set_url_rank_appended(self, source, n_base, shift, keys):
        self.rank[source] = np.concatenate([(np.arange(n_base, dtype=np.uint64) + 1 << shift).astype(np.uint32),
                                            np.asarray(keys, np.uint32)])

    def set_idf(self, idf, present=None):
        self.orc.set_idf(idf, present)

    def search(self, queries, k, source=1):
        nq = len(queries)
        out = dict(unit=np.zeros((nq, k), np.uint32), score=np.zeros((nq, k)), matched=np.zeros((nq, k), np.uint32),
                   count=np.zeros(nq, np.uint32))
        for i, q in enumerate(queries):
            u, s, m = self.orc.search(np.asarray(q, np.uint32), k, source=source, url_rank=self.rank[source], level="csr")
            n = len(u)
            out["unit"][i, :n], out["score"][i, :n], out["matched"][i, :n], out["count"][i] = u, s, m, n
        return out

    def update(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, changed_docs=(),
               changed_articles=(), **kw):
        # the stand-in has no segments: it drops the changed units' rows from its copy of the table and rebuilds
        o = self.rows
        src = o[3] & 3
        dead = ((src == 1) & np.isin(o[1], np.asarray(changed_docs, np.uint32))) | \
               ((src == 2) & np.isin(o[1], np.asarray(changed_articles, np.uint32)))
        rows = [np.concatenate([a[~dead], np.asarray(b, a.dtype)]) for a, b in zip(o, (term_id, unit_id, tf, meta))]
        self.build(*rows, n_terms, n_docs, n_articles, word_count, doc_flags)

    def close(self):
        pass


def sql_reference_rows(path, terms, limit):
    """The reference's web statement on the same database (its term_statistics already written)."""
    db = sqlite3.connect(path)
    ph = ",".join("?" * len(terms))
    rows = db.execute(f"""
        WITH scored AS (SELECT d.id, d.title, d.description, d.url,
            SUM(t.frequency * COALESCE(ts.idf, 1.0) * CASE t.field WHEN 1 THEN 3.2 WHEN 2 THEN 1.7 ELSE 1.0 END) AS token_score,
            COUNT(DISTINCT t.term) AS matched_terms
          FROM tokens t JOIN documents d ON d.id = t.doc_id LEFT JOIN term_statistics ts ON ts.term = t.term
          WHERE d.status = 'done' AND t.term IN ({ph}) GROUP BY d.id, d.title, d.description, d.url)
        SELECT title, COALESCE(description, ''), url, token_score, matched_terms FROM scored
        ORDER BY token_score DESC, url ASC LIMIT ?""", list(terms) + [limit]).fetchall()
    db.close()
    return rows


def run_flow(tmp_path, index_factory):
    path = str(tmp_path / "se.db")
    make_db(path)
    store = SqlStore(lambda: sqlite3.connect(path))
    engine = SearchEngine(store, index_factory=index_factory)
    bm25_stats.configure(engine)
    ss.configure(engine)

    async def flow():
        await asyncio.gather(bm25_stats.run(), asyncio.sleep(0))      # shares a loop with sibling jobs (runner.py:52)
        web = await ss.perform_web_search(q="The radix sorting", limit=2, offset=0)
        news = await ss.perform_news_search(q="radix sort")
        empty = await ss.perform_web_search(q="the of and")
        return web, news, empty

    web, news, empty = asyncio.run(flow())
    # ---- term_statistics was replaced with what the SQL job would have written
    db = sqlite3.connect(path)
    got = {t: (df, idf, al) for t, df, idf, al in db.execute("SELECT * FROM term_statistics")}
    n_done = db.execute("SELECT COUNT(*) FROM documents WHERE status='done'").fetchone()[0]
    avg = db.execute("SELECT AVG(word_count) FROM documents WHERE status='done'").fetchone()[0]
    want = db.execute("SELECT term, COUNT(DISTINCT doc_id) FROM tokens GROUP BY term").fetchall()
    db.close()
    assert set(got) == {t for t, _ in want}
    for term, df in want:
        assert got[term][0] == df
        assert got[term][1] == pytest.approx(math.log((n_done - df + 0.5) / (df + 0.5) + 1), rel=1e-14)
        assert got[term][2] == pytest.approx(avg, rel=1e-15)
    assert got["radix"][0] == 2          # docs 11 and 21 (any status); the news row adds nothing
    assert got["record"][0] == 0         # news-only term
    # ---- web search: candidates equal the reference statement's rows, then the intent re-rank
    svc = ss.search_service
    ctx = svc._search_context("The radix sorting", 2, 0)
    assert ctx["query_terms"] == ["radix", "sort"] and ctx["candidate_limit"] == 202
    rows = engine.score_web(ctx["query_terms"], ctx["candidate_limit"])
    ref = sql_reference_rows(path, ctx["query_terms"], ctx["candidate_limit"])
    assert [r[:3] for r in rows] == [r[:3] for r in ref]
    assert [r[4] for r in rows] == [r[4] for r in ref]
    np.testing.assert_allclose([r[3] for r in rows], [r[3] for r in ref], rtol=1e-12)
    assert "https://gpu.example/draft" not in [r[2] for r in rows]     # status != 'done'
    expect = svc._rank_web_rows(ref, context=ctx)
    assert [i.url for i in web.results] == [i.url for i in expect[:2]]
    np.testing.assert_allclose([i.score for i in web.results], [i.score for i in expect[:2]], rtol=1e-12)
    assert web.count == max(len(expect), 0 + len(web.results)) and web.results[0].url == "https://gpu.example/radix-sort"
    # ---- news: statement order is kept, +8 bonus, feed attached
    assert [i.url for i in news.results] == ["https://news.example/a1"]
    assert news.results[0].feed.name == "Example News" and news.results[0].author == "Ann"
    assert news.results[0].published_at == "2026-02-01T10:00:00"
    assert empty.results == [] and empty.count == 0
    return engine


def test_flow_on_cpu_stand_in(tmp_path):
    run_flow(tmp_path, OracleBackedIndex)


@pytest.mark.gpu
def test_flow_on_gpu(tmp_path):
    engine = run_flow(tmp_path, None)          # default factory: native.Index on cuda:0
    engine.load_idf_from_table()               # COALESCE path: same table, same results
    rows = engine.score_web(["radix", "sort"], 50)
    assert rows[0][2] == "https://gpu.example/radix-sort"
    check_fallback_rows(engine, str(tmp_path / "se.db"))
    check_concurrent_requests(engine)
    engine.index.close()


def check_fallback_rows(engine, path):
    """The fallback statement's rows (search_service.py:42-66): (token_score, matched_terms) by score, and the
    reduced ranking the service applies when the row text cannot be decoded (:321-338)."""
    terms = ["radix", "sort"]
    pairs = engine.score_web_pairs(terms, 50)
    db = sqlite3.connect(path)
    ref = db.execute("""SELECT SUM(t.frequency * COALESCE(ts.idf, 1.0) * CASE t.field WHEN 1 THEN 3.2 WHEN 2 THEN 1.7 ELSE 1.0 END) AS s,
                               COUNT(DISTINCT t.term) FROM tokens t JOIN documents d ON d.id = t.doc_id
                        LEFT JOIN term_statistics ts ON ts.term = t.term
                        WHERE d.status = 'done' AND t.term IN (?, ?) GROUP BY d.id ORDER BY s DESC LIMIT 50""", terms).fetchall()
    db.close()
    assert [m for _, m in pairs] == [m for _, m in ref]
    np.testing.assert_allclose([s for s, _ in pairs], [s for s, _ in ref], rtol=1e-12)

    class Undecodable(type(engine.store)):
        def web_rows(self, doc_ids):
            raise UnicodeDecodeError("utf-8", b"\xff", 0, 1, "invalid start byte")

    good = engine.store
    engine.store = Undecodable(good._connect, good._p)
    try:
        svc = ss.SearchService(engine)
        out = asyncio.run(svc.perform_web_search(q="radix sort", limit=2, offset=0))
    finally:
        engine.store = good
    ctx = svc._search_context("radix sort", 2, 0)
    expect = svc._rank_score_only_rows(pairs, context=ctx)
    assert [i.score for i in out.results] == [i.score for i in expect[:2]]
    assert all(i.url == "" and i.title == "" for i in out.results) and out.count == len(expect)


def check_concurrent_requests(engine, n=96):
    """Many coroutines at once: coalesced into a few batched searches, results identical to serial calls."""
    svc = ss.SearchService(engine)
    queries = ["radix sort", "ranking documents", "postgres sorting rows", "bm25", "archive stories", "gpu keys", "the of"]

    async def serial():
        return [await svc.perform_web_search(q=queries[i % len(queries)], limit=1 + i % 3, offset=i % 2) for i in range(n)] + \
               [await svc.perform_news_search(q=queries[i % len(queries)]) for i in range(8)]

    async def together():
        return await asyncio.gather(*[svc.perform_web_search(q=queries[i % len(queries)], limit=1 + i % 3, offset=i % 2) for i in range(n)],
                                    *[svc.perform_news_search(q=queries[i % len(queries)]) for i in range(8)])

    a = asyncio.run(serial())
    before = svc.batcher.batches
    b = asyncio.run(together())
    assert [r.model_dump() for r in a] == [r.model_dump() for r in b]
    assert svc.batcher.batches - before <= 2          # ~100 requests, one or two engine calls


def test_fallback_rows_and_concurrent_requests_on_cpu_stand_in(tmp_path):
    engine = run_flow(tmp_path, OracleBackedIndex)
    check_fallback_rows(engine, str(tmp_path / "se.db"))
    check_concurrent_requests(engine)


def test_a_query_beyond_the_term_limit_fails_alone(tmp_path, monkeypatch):
    """A request with more distinct indexed terms than the library scores gets an exception of its own; the requests
    that were batched with it get their results (the reference's `= ANY($1)` has no such limit, so this is where the
    drop-in says no instead of failing a whole batch)."""
    import search_engine_amd.engine as engine_module

    engine = run_flow(tmp_path, OracleBackedIndex)
    svc = ss.SearchService(engine)
    alone = asyncio.run(svc.perform_web_search(q="radix sort", limit=5))
    monkeypatch.setattr(engine_module, "MAX_QUERY_TERMS", 2)

    async def together():
        return await asyncio.gather(svc.perform_web_search(q="postgres sorting rows", limit=5),
                                    svc.perform_web_search(q="radix sort", limit=5), return_exceptions=True)

    before = svc.batcher.batches
    long_one, short_one = asyncio.run(together())
    assert svc.batcher.batches - before == 1
    assert isinstance(long_one, ValueError) and "distinct indexed terms" in str(long_one)
    assert short_one.model_dump() == alone.model_dump()
    with pytest.raises(ValueError):
        engine.score_web(list(tokenize("postgres sorting rows").keys()), 10)


def test_job_and_service_in_separate_engines(tmp_path):
    """The statistics job and the search API are separate processes in the reference (docker-compose.yml:50,67):
    here two engines over one database.  The service builds itself on first use, scores with the table's idf
    and follows the job."""
    path = str(tmp_path / "se.db")
    make_db(path)
    store = SqlStore(lambda: sqlite3.connect(path))
    service_engine = SearchEngine(store, index_factory=OracleBackedIndex, idf_source="table")
    svc = ss.SearchService(service_engine)
    # before the job ever ran: no term_statistics rows, every idf is COALESCE(NULL, 1.0)
    out0 = asyncio.run(svc.perform_web_search(q="radix sort", limit=5))
    rows0 = service_engine.score_web(["radix", "sort"], 50)
    assert out0.results and [r[:3] for r in rows0] == [r[:3] for r in sql_reference_rows(path, ["radix", "sort"], 50)]
    np.testing.assert_allclose([r[3] for r in rows0], [r[3] for r in sql_reference_rows(path, ["radix", "sort"], 50)], rtol=1e-12)
    job_engine = SearchEngine(store, index_factory=OracleBackedIndex)
    job_engine.rebuild()                                   # the job process writes the table
    assert service_engine.refresh_if_changed() is True     # the service notices and rebuilds
    assert service_engine.refresh_if_changed() is False
    rows1 = service_engine.score_web(["radix", "sort"], 50)
    ref1 = sql_reference_rows(path, ["radix", "sort"], 50)
    assert [r[:3] for r in rows1] == [r[:3] for r in ref1]
    np.testing.assert_allclose([r[3] for r in rows1], [r[3] for r in ref1], rtol=1e-12)
    # the crawler adds a page: invisible until the job has run and the service has refreshed
    db = sqlite3.connect(path)
    db.execute("INSERT INTO documents VALUES (30,'https://gpu.example/new','Radix radix','', 'radix', 1, 'done')")
    db.execute("INSERT INTO tokens VALUES (30,'radix',1,2,NULL,1)")
    db.commit()
    db.close()
    assert "https://gpu.example/new" not in [r[2] for r in service_engine.score_web(["radix"], 50)]
    job_engine.rebuild()
    service_engine.start_auto_refresh(0.05)
    import time
    for _ in range(100):
        if "https://gpu.example/new" in [r[2] for r in service_engine.score_web(["radix"], 50)]:
            break
        time.sleep(0.05)
    service_engine.stop_auto_refresh()
    assert [r[:3] for r in service_engine.score_web(["radix"], 50)] == [r[:3] for r in sql_reference_rows(path, ["radix"], 50)]


def test_export_matches_row_by_row_conversion_and_copy_text(tmp_path):
    """The chunked column-wise export equals a plain row-by-row conversion, whatever the chunk size, and the
    COPY text path (PostgreSQL) parses to the same arrays.  Rows without a unit keep their term's statistics row."""
    path = str(tmp_path / "se.db")
    make_db(path)
    db = sqlite3.connect(path)
    db.execute("INSERT INTO tokens VALUES (NULL,'ghostterm',4,3,'https://news.example/gone',2)")      # article deleted meanwhile
    db.execute("INSERT INTO tokens VALUES (99,'phantomdoc',4,1,NULL,1)")                              # doc id only tokens knows
    db.commit()
    rows = db.execute("SELECT COALESCE(doc_id,-1), term, field, frequency, COALESCE(article_url,''), COALESCE(source_type,1) FROM tokens").fetchall()
    db.close()
    exports = [SqlStore(lambda: sqlite3.connect(path), chunk_rows=c).export_tokens() for c in (7, 1 << 20)]

    class CopyCursor:
        """Stands in for a psycopg cursor: .copy(sql) yields the rows in COPY text format, in uneven pieces."""
        def __init__(self, cur):
            self._cur = cur
        def execute(self, sql, params=()):
            if not sql.startswith("SET TRANSACTION"):
                self._cur.execute(sql.replace("%s", "?"), params)
        def fetchall(self):
            return self._cur.fetchall()
        def copy(self, sql):
            text = "".join("\t".join(str(x) for x in r) + "\n" for r in rows).encode()
            class Ctx:
                def __enter__(s):
                    return iter([text[:11], text[11:200], text[200:]])
                def __exit__(s, *a):
                    return False
            return Ctx()

    class CopyConn:
        def __init__(self):
            self._c = sqlite3.connect(path)
        def cursor(self):
            return CopyCursor(self._c.cursor())
        def rollback(self):
            self._c.rollback()
        def close(self):
            self._c.close()

    exports.append(SqlStore(CopyConn, "%s").export_tokens())
    a = exports[0]
    assert "ghostterm" in a.terms and "phantomdoc" in a.terms
    assert a.extra_term_rows[a.term_index["ghostterm"]] == 1 and a.extra_term_rows.sum() == 1
    assert 99 in a.doc_ids.tolist() and a.doc_flags[a.doc_ids.tolist().index(99)] == 0
    for b in exports[1:]:
        for name in ("term_id", "unit_id", "tf", "meta", "doc_ids", "word_count", "doc_flags", "extra_term_rows"):
            np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
        assert a.terms == b.terms and a.doc_urls == b.doc_urls and a.article_urls == b.article_urls
    # row-by-row conversion of the same table
    term_index = {t: i for i, t in enumerate(sorted({r[1] for r in rows}))}
    doc_pos = {int(d): i for i, d in enumerate(a.doc_ids.tolist())}
    art_pos = {u: i for i, u in enumerate(a.article_urls)}
    want = []
    for d, t, f, fr, art, st in rows:
        if st == 2:
            if art in art_pos:
                want.append((term_index[t], art_pos[art], fr, 2 | (f << 2)))
        else:
            want.append((term_index[t], doc_pos[d], fr, 1 | (f << 2)))
    got = list(zip(a.term_id.tolist(), a.unit_id.tolist(), a.tf.tolist(), a.meta.tolist()))
    assert got == want
    # end to end: the job writes a row for the orphan's term with df = 0, as GROUP BY t.term over `tokens` does
    store = SqlStore(lambda: sqlite3.connect(path))
    SearchEngine(store, index_factory=OracleBackedIndex).rebuild()
    db = sqlite3.connect(path)
    got = dict(db.execute("SELECT term, doc_frequency FROM term_statistics").fetchall())
    want_df = dict(db.execute("SELECT term, COUNT(DISTINCT doc_id) FROM tokens GROUP BY term").fetchall())
    db.close()
    assert got == want_df and got["ghostterm"] == 0 and got["phantomdoc"] == 1


def test_candidate_limit_formula():
    svc = ss.SearchService()
    f = lambda limit, offset: svc._search_context("radix", limit, offset)["candidate_limit"]   # noqa: E731
    assert f(10, 0) == 210 and f(20, 0) == 220 and f(100, 0) == 1000 and f(100, 1900) == 2000 and f(1, 5000) == 2000
    assert svc._search_context("of the", 10, 0) is None


def test_intent_score_hand_computed():
    svc = ss.SearchService()
    ctx = svc._search_context("radix sort", 10, 0)
    s = svc._intent_score(token_score=math.e - 1, matched_terms=2, total_terms=2, query_phrase=ctx["query_phrase"],
                          query_compact=ctx["query_compact"], query_words=ctx["query_words"],
                          title="Radix sort on GPUs", description="About radix sort.", url="https://x.test/radix-sort")
    # log1p(e-1)*12 + coverage 25 + all terms 40 + phrase in title 140, url 70, description 25, compact url 90
    # + hits: title 2*22, url 2*16, compact url 2*12 + all-words bonuses 80 + 55 + 45
    assert s == pytest.approx(12 + 25 + 40 + 140 + 70 + 25 + 90 + 44 + 32 + 24 + 80 + 55 + 45)
    s2 = svc._intent_score(token_score=-3.0, matched_terms=1, total_terms=2, query_phrase="radix sort", query_compact="radixsort",
                           query_words=["radix", "sort"], title="", description="", url="https://x.test/")
    assert s2 == pytest.approx(12.5)            # negative token scores clamp to 0; half coverage


def test_score_only_ranking_and_news_order():
    svc = ss.SearchService()
    ctx = svc._search_context("radix sort", 10, 0)
    items = svc._rank_score_only_rows([(1.0, 1), (0.5, 2)], context=ctx)
    assert [round(i.score, 6) for i in items] == sorted([round(math.log1p(1.0) * 12 + 12.5, 6), round(math.log1p(0.5) * 12 + 65, 6)], reverse=True)
    row = lambda url, sc: ("t", "", url, None, None, None, None, None, None, None, None, None, None, None, None, sc, 1)   # noqa: E731
    out = svc._rank_news_rows([row("https://b", 1.0), row("https://a", 9.0)], context=ctx)
    assert [i.url for i in out] == ["https://b", "https://a"] and out[0].feed is None     # no re-sort


def test_tokenizer_known_answers():
    assert list(tokenize("The Quick brown foxes; the QUICK dog's 2 x86 cpus!").items()) == [
        ("quick", 2), ("brown", 1), ("fox", 1), ("dog", 1), ("x86", 1), ("cpu", 1)]
    pairs = {"caresses": "caress", "ponies": "poni", "cats": "cat", "feed": "feed", "plastered": "plaster",
             "motoring": "motor", "sing": "sing", "conflated": "conflat", "troubled": "troubl", "sized": "size",
             "hopping": "hop", "falling": "fall", "hissing": "hiss", "failing": "fail", "filing": "file", "happy": "happi",
             "relational": "relat", "conditional": "condit", "digitizer": "digit", "vietnamization": "vietnam",
             "operator": "oper", "feudalism": "feudal", "decisiveness": "decis", "hopefulness": "hope",
             "formaliti": "formal", "sensitiviti": "sensit", "triplicate": "triplic", "electrical": "electr",
             "goodness": "good", "revival": "reviv", "allowance": "allow", "inference": "infer", "adjustable": "adjust",
             "replacement": "replac", "adoption": "adopt", "communism": "commun", "effective": "effect", "rate": "rate",
             "cease": "ceas", "controll": "control", "roll": "roll", "running": "run", "sorting": "sort", "ranking": "rank"}
    for w, s in pairs.items():
        assert porter_stem(w) == s, w
    assert porter_stem("dying") == "die" and porter_stem("skies") == "sky" and porter_stem("ties") == "tie"
    assert porter_stem("by") == "by" and porter_stem("a") == "a"


def test_builtin_stemmer_against_published_porter_and_nltk_outputs():
    """The built-in stemmer against outputs that are in print: the example list of NLTK's PorterStemmer documentation,
    the Porter-stemmed sentence of the NLTK book (chapter 3), the worked examples of Porter's 1980 paper carried through
    all steps, and one pair or more for every departure of NLTK's default mode from the paper (irregular-form pool,
    short words, -ies / -ied after a three-letter stem, the consonant test before a final y, fulli / logi / bli in
    step 2).  nltk itself is not installed here: still "parity unpinned" against the stemmer the reference's term
    strings come from, but every rule the two could differ in has a pair below."""
    nltk_doc = dict(zip("caresses flies dies mules denied died agreed owned humbled sized meeting stating siezing itemization "
                        "sensational traditional reference colonizer plotted".split(),
                        "caress fli die mule deni die agre own humbl size meet state siez item sensat tradit refer colon plot".split()))
    nltk_book = dict(zip("dennis listen strange women lying in ponds distributing swords is no basis for a system of government "
                         "supreme executive power derives from a mandate from the masses not from some farcical aquatic "
                         "ceremony".split(),
                         "denni listen strang women lie in pond distribut sword is no basi for a system of govern suprem execut "
                         "power deriv from a mandat from the mass not from some farcic aquat ceremoni".split()))
    paper = {"generalizations": "gener", "oscillators": "oscil", "caress": "caress", "bled": "bled", "tanned": "tan", "fizzed": "fizz",
             "valenci": "valenc", "hesitanci": "hesit", "conformabli": "conform", "radicalli": "radic", "differentli": "differ",
             "vileli": "vile", "analogousli": "analog", "predication": "predic", "callousness": "callous", "sensibiliti": "sensibl",
             "formative": "form", "formalize": "formal", "electriciti": "electr", "hopeful": "hope", "airliner": "airlin",
             "gyroscopic": "gyroscop", "defensible": "defens", "irritant": "irrit", "adjustment": "adjust", "dependent": "depend",
             "homologou": "homolog", "activate": "activ", "angulariti": "angular", "homologous": "homolog", "bowdlerize": "bowdler",
             "probate": "probat", "rational": "ration"}
    departures = {
        # the pool of irregular forms
        "sky": "sky", "skies": "sky", "dying": "die", "lying": "lie", "tying": "tie", "news": "news", "innings": "inning",
        "inning": "inning", "outings": "outing", "outing": "outing", "cannings": "canning", "canning": "canning", "howe": "howe",
        "proceed": "proceed", "exceed": "exceed", "succeed": "succeed",
        # words of one or two letters are left alone
        "by": "by", "is": "is", "as": "as", "a": "a",
        # -ies / -ied: "ie" after a one-letter... three-letter word, "i" otherwise
        "ties": "tie", "lies": "lie", "pies": "pie", "ponies": "poni", "cries": "cri",
        "tied": "tie", "lied": "lie", "tried": "tri", "spied": "spi", "applied": "appli",
        # a final y turns into i only behind a consonant (the paper: behind any stem with a vowel: enjoy -> enjoi)
        "enjoy": "enjoy", "enjoyment": "enjoy", "say": "say", "saying": "say", "fly": "fli", "cry": "cri", "happy": "happi",
        # step 2: fulli -> ful, logi -> log, bli -> ble
        "hopefully": "hope", "apology": "apolog", "possibly": "possibl",
    }
    n = 0
    for table in (nltk_doc, nltk_book, paper, departures):
        for w, stem in table.items():
            assert porter_stem(w) == stem, (w, stem, porter_stem(w))
            n += 1
    assert n >= 120


def test_url_ranks():
    r = url_ranks(["https://b", "https://a", "https://c"])
    assert r.tolist() == [1, 0, 2]
    r = url_ranks(["B", "a", "C"], key=str.lower)
    assert r.tolist() == [1, 0, 2]


def check_engine_update(tmp_path, index_factory, via_capture=False):
    """Crawler-style changes applied through SearchEngine.update: term_statistics and search rows equal what the
    reference's statements give on the changed database.  via_capture: nobody tells the job what changed; the
    change-capture triggers do."""
    path = str(tmp_path / "se.db")
    make_db(path)
    store = SqlStore(lambda: sqlite3.connect(path))
    engine = SearchEngine(store, index_factory=index_factory)
    engine.rebuild()
    if via_capture:
        store.install_change_capture()
        store.install_change_capture()      # idempotent
        assert engine.update_pending() is None
    db = sqlite3.connect(path)
    # doc 12 re-crawled with new text, doc 15 removed, doc 21 finished, a new doc 40, a new article, article a2 removed
    def put_doc(doc_id, url, title, desc, content, status="done"):
        db.execute("DELETE FROM tokens WHERE doc_id = ?", (doc_id,))                     # worker.py:229
        db.execute("INSERT OR REPLACE INTO documents VALUES (?,?,?,?,?,?,?)", (doc_id, url, title, desc, content, len(content.split()), status))
        for field, text in ((1, title), (2, desc), (4, content)):
            for term, freq in tokenize(text or "").items():
                db.execute("INSERT INTO tokens VALUES (?,?,?,?,NULL,1)", (doc_id, term, field, freq))
    put_doc(12, "https://gpu.example/bm25", "BM25 and radix", "Ranking with radix sorted postings", "ranking bm25 radix sort postings quantum")
    db.execute("DELETE FROM tokens WHERE doc_id = 15")
    db.execute("DELETE FROM documents WHERE id = 15")
    put_doc(21, "https://gpu.example/draft", "Radix draft", "finished now", "radix radix radix sort")
    put_doc(40, "https://aaa.example/first", "Sorting radix keys", "", "radix sort keys quantum")
    db.execute("INSERT INTO news_articles VALUES ('https://news.example/a0','https://news.example/feed','Quantum radix','',NULL,'quantum radix sort','Bo',NULL)")
    for term, freq in tokenize("Quantum radix quantum radix sort").items():
        db.execute("INSERT INTO tokens VALUES (NULL,?,4,?,'https://news.example/a0',2)", (term, freq))
    db.execute("DELETE FROM tokens WHERE article_url = 'https://news.example/a2'")
    db.execute("DELETE FROM news_articles WHERE url = 'https://news.example/a2'")
    db.commit()
    db.close()
    if via_capture:
        from search_engine_amd.batch import bm25_stats
        import asyncio

        bm25_stats.configure(engine)
        asyncio.run(bm25_stats.run_incremental())
        assert store.drain_changes() == ([], [])
    else:
        engine.update(doc_ids=[12, 15, 21, 40, 999], article_urls=["https://news.example/a0", "https://news.example/a2"])
    db = sqlite3.connect(path)
    got = {t: (df, idf) for t, df, idf, _ in db.execute("SELECT * FROM term_statistics")}
    n_done = db.execute("SELECT COUNT(*) FROM documents WHERE status='done'").fetchone()[0]
    want = db.execute("SELECT term, COUNT(DISTINCT doc_id) FROM tokens GROUP BY term").fetchall()
    db.close()
    assert set(got) == {t for t, _ in want} and "quantum" in got
    for term, df in want:
        assert got[term][0] == df, term
        assert got[term][1] == pytest.approx(math.log((n_done - df + 0.5) / (df + 0.5) + 1), rel=1e-14)
    for terms in (["radix", "sort"], ["quantum"], ["rank", "bm25", "post"]):
        rows, ref = engine.score_web(terms, 50), sql_reference_rows(path, terms, 50)
        assert [r[:3] for r in rows] == [r[:3] for r in ref] and [r[4] for r in rows] == [r[4] for r in ref], terms
        np.testing.assert_allclose([r[3] for r in rows], [r[3] for r in ref], rtol=1e-12)
    news = engine.score_news(["radix", "quantum"], 50)
    assert [r[2] for r in news] == ["https://news.example/a0", "https://news.example/a1"]
    return engine


def test_engine_update_on_cpu_stand_in(tmp_path):
    check_engine_update(tmp_path, OracleBackedIndex)


def test_engine_update_from_change_capture_on_cpu_stand_in(tmp_path):
    check_engine_update(tmp_path, OracleBackedIndex, via_capture=True)


def test_engine_update_that_fails_in_the_index_rebuilds_from_the_database(tmp_path):
    """The export's dictionaries are extended before the index takes the rows; when the index refuses them the engine
    must not go on with an export and an index that describe different id spaces: it rebuilds and raises."""
    path = str(tmp_path / "se.db")
    make_db(path)
    store = SqlStore(lambda: sqlite3.connect(path))

    class FailingOnce(OracleBackedIndex):
        fail = True

        def update(self, *a, **kw):
            if FailingOnce.fail:
                FailingOnce.fail = False
                raise RuntimeError("index refused the update")
            return super().update(*a, **kw)

    engine = SearchEngine(store, index_factory=FailingOnce)
    engine.rebuild()
    first_export = engine.export
    db = sqlite3.connect(path)
    db.execute("INSERT INTO documents VALUES (40,'https://aaa.example/first','Sorting radix keys','','radix sort keys quantum',4,'done')")
    for term, freq in tokenize("radix sort keys quantum").items():
        db.execute("INSERT INTO tokens VALUES (40,?,4,?,NULL,1)", (term, freq))
    db.commit()
    db.close()
    with pytest.raises(RuntimeError, match="refused"):
        engine.update(doc_ids=[40])
    assert engine.export is not first_export            # rebuilt: export and index are one snapshot again
    for terms in (["radix", "sort"], ["quantum"]):
        rows, ref = engine.score_web(terms, 50), sql_reference_rows(path, terms, 50)
        assert [r[:3] for r in rows] == [r[:3] for r in ref], terms
    engine.update(doc_ids=[40])                          # and updates work again
    rows, ref = engine.score_web(["quantum"], 50), sql_reference_rows(path, ["quantum"], 50)
    assert [r[:3] for r in rows] == [r[:3] for r in ref]
    # the same through change capture: a run that fails lists its units again
    store.install_change_capture()
    db = sqlite3.connect(path)
    db.execute("UPDATE documents SET status = 'pending' WHERE id = 40")
    db.commit()
    db.close()
    FailingOnce.fail = True
    with pytest.raises(RuntimeError, match="refused"):
        engine.update_pending()
    assert store.drain_changes() == ([40], [])
    store.note_changes([40])
    assert engine.update_pending() is not None
    assert engine.score_web(["quantum"], 50) == sql_reference_rows(path, ["quantum"], 50) == []


@pytest.mark.gpu
def test_engine_update_on_gpu(tmp_path):
    engine = check_engine_update(tmp_path, None)
    assert engine.index.info().n_segments in (1, 2)      # a tiny index crosses the dead-row threshold and is compacted at once
    engine.index.close()


def test_copy_text_parsers_agree_on_escapes_and_empty_fields():
    """COPY text escapes backslash, tab and newline inside a value; both parsers (pyarrow when importable, pandas)
    must hand the assembler the same columns."""
    from search_engine_amd import store as S

    urls = ["https://a.example/x", "https://a.example/tab\there", "https://a.example/back\\slash", "", "https://a.example/nl\nx"]
    terms = ["plain", "back\\slash", "tab\tbed", "plain", "ünï"]

    def esc(v):
        return v.replace("\\", "\\\\").replace("\t", "\\t").replace("\n", "\\n")

    lines = "".join(f"{i - 1}\t{esc(t)}\t{i % 3 + 1}\t{i + 1}\t{esc(u)}\t{1 + (i % 2)}\n" for i, (t, u) in enumerate(zip(terms, urls)))
    got = [S._parse_copy_text(lines.encode()), S._parse_copy_text_pandas(lines.encode())]
    for cols in got:
        d, t, f, q, a, s = cols
        assert d.tolist() == [-1, 0, 1, 2, 3] and f.tolist() == [1, 2, 3, 1, 2] and q.tolist() == [1, 2, 3, 4, 5]
        assert s.tolist() == [1, 2, 1, 2, 1]
        assert [t.uniq[c] for c in t.codes.tolist()] == terms
        assert [a.uniq[c] for c in a.codes.tolist()] == urls


def test_term_statistics_write_back_as_copy_text():
    """PostgreSQL leg of the write-back without a server: TRUNCATE + COPY ... FROM STDIN in one transaction, a
    statistics row for every term that has token rows (also those whose rows have no unit), floats that parse back
    to the same bits, text escaped the way COPY reads it."""
    log = []

    class Copy:
        def __enter__(self):
            return self
        def __exit__(self, *a):
            return False
        def write(self, data):
            log.append(("data", bytes(data)))

    class Cur:
        def execute(self, sql, params=()):
            log.append(("sql", sql))
        def copy(self, sql):
            log.append(("copy", sql))
            return Copy()

    class Conn:
        def cursor(self):
            return Cur()
        def commit(self):
            log.append(("commit",))
        def rollback(self):
            log.append(("rollback",))
        def close(self):
            pass

    terms = ["alpha", "tab\there", "back\\slash", "norows", "orphan"]
    df = np.array([3, 1, 2, 0, 0]); idf = np.array([0.1, 1.0 / 3.0, 2.718281828459045, 9.0, 7.5])
    n = SqlStore(Conn, "%s").write_term_statistics(terms, df, idf, 123.456, np.array([5, 1, 2, 0, 0]), extra_term_rows=np.array([0, 0, 0, 0, 4]))
    assert n == 4                                     # "norows" has no token rows at all
    kinds = [e[0] for e in log]
    assert kinds[0] == "sql" and log[0][1].startswith("TRUNCATE") and kinds[1] == "copy" and kinds[-1] == "commit"
    text = b"".join(e[1] for e in log if e[0] == "data").decode()
    rows = [line.split("\t") for line in text.splitlines()]
    unesc = lambda v: v.replace("\\t", "\t").replace("\\\\", "\\")  # noqa: E731
    assert [unesc(r[0]) for r in rows] == ["alpha", "tab\there", "back\\slash", "orphan"]
    assert [int(r[1]) for r in rows] == [3, 1, 2, 0]
    assert [float(r[2]) for r in rows] == [0.1, 1.0 / 3.0, 2.718281828459045, 7.5] and {float(r[3]) for r in rows} == {123.456}


def test_intent_score_with_cached_text_features_equals_the_formula_built_from_the_helpers():
    """The per-document text features are cached; the score must be what the reference's formula gives when it is
    put together from the plain helpers (search_service.py:150-222), also for None / empty fields and odd text."""
    import math
    import random

    from search_engine_amd.api import search_service as S

    svc = S.SearchService()
    rnd = random.Random(3)
    words = ["radix", "sort", "gpu", "ünï", "c++", "rust-lang", "2024", "a.b", "Search", "ENGINE", ""]
    def text():
        return rnd.choice([None, "", " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 6))) + rnd.choice(["", "!", " - x"])])
    for _ in range(300):
        title, desc = text(), text()
        url = "https://" + "/".join((rnd.choice(words) or "x") for _ in range(rnd.randint(1, 4)))
        q = " ".join(rnd.choice(words) for _ in range(rnd.randint(1, 3))) or "gpu"
        ctx = svc._search_context(q, 10, 0)
        if not ctx:
            continue
        ts, m = rnd.random() * 30, rnd.randint(0, ctx["total_terms"])
        got = svc._score_row(ctx, ts, m, title or "", desc or "", url)
        want = math.log1p(max(ts, 0.0)) * S._W_LOG
        want += (m / ctx["total_terms"]) * S._W_COVERAGE
        if m == ctx["total_terms"]:
            want += S._W_ALL_TERMS
        compact_url = S._NON_ALNUM.sub("", url.lower())
        norm = {"title": svc._normalize_text(title or ""), "url": svc._normalize_text(url), "description": svc._normalize_text(desc or "")}
        if ctx["query_phrase"]:
            for field, bonus in S._PHRASE_BONUS:
                if ctx["query_phrase"] in norm[field]:
                    want += bonus
        if ctx["query_compact"] and ctx["query_compact"] in compact_url:
            want += S._COMPACT_URL_PHRASE
        hits = {"title": svc._count_hits(title or "", ctx["query_words"]), "url": svc._count_hits(url, ctx["query_words"]),
                "compact_url": svc._compact_word_hits(compact_url, ctx["query_words"])}
        for field, per_hit, _ in S._HIT_WEIGHT:
            want += hits[field] * per_hit
        if ctx["query_words"]:
            for field, _, all_bonus in S._HIT_WEIGHT:
                if hits[field] == len(ctx["query_words"]):
                    want += all_bonus
        assert got == want, (q, title, desc, url)


def test_url_ranker_keys_for_appended_units_order_like_full_ranks():
    """Keys that need no array for the units of the last rebuild ((unit + 1) << shift) and one key per appended unit:
    the same order as the ranks of a full sort, through appends, re-crawls, dropped units; a base unit that moves
    ends it."""
    from search_engine_amd.store import UrlRanker

    rng = np.random.default_rng(5)
    urls = sorted(f"https://h{int(x):05d}.example/p" for x in rng.choice(90000, 400, replace=False))
    r = UrlRanker(urls, presorted=True)
    nb0, shift0, keys0 = r.appended_keys()
    assert (nb0, shift0, len(keys0)) == (400, 8, 0)
    for round_ in range(6):
        new = [f"https://h{int(x):05d}.example/q{round_}" for x in rng.integers(0, 90000, 30)] + ["https://a.example/first"] * (round_ == 0)
        dropped = {int(i): None for i in rng.choice(len(r.urls), 5, replace=False)}
        r.apply(dropped, new)
        nb, shift, keys = r.appended_keys()
        assert nb == 400 and shift == 8 and len(keys) == len(r.urls) - 400
        full = np.concatenate([(np.arange(nb, dtype=np.uint64) + 1 << shift).astype(np.uint32), keys])
        live = np.array([u is not None for u in r.urls])
        ranks = r.ranks()
        order_by_key = np.argsort(full[live], kind="stable")
        order_by_rank = np.argsort(ranks[live], kind="stable")
        np.testing.assert_array_equal(order_by_key, order_by_rank)
        assert len(set(full[live].tolist())) == int(live.sum())          # keys are distinct
    # 300 units into one gap: more than 255 keys
    r2 = UrlRanker(urls, presorted=True)
    r2.apply({}, [f"{urls[10]}/{i:04d}" for i in range(300)])
    assert r2.appended_keys() is None and not r2.identity
    # a base unit whose url changes moves: every unit needs its rank again
    r3 = UrlRanker(urls, presorted=True)
    r3.apply({3: "https://zzz.example/"}, ["https://h00000.example/x"])
    assert r3.appended_keys() is None


def test_url_ranker_keeps_the_ranks_of_a_full_sort():
    """Units appended, urls changed or dropped: the ranks kept unit by unit equal those of sorting all urls again,
    and `identity` holds exactly while every unit's index is its rank."""
    import random

    from search_engine_amd.store import UrlRanker, url_ranks

    rnd = random.Random(1)
    for trial in range(120):
        n = rnd.randint(0, 40)
        urls = list(dict.fromkeys("u%04d" % rnd.randint(0, 9999) for _ in range(n)))
        urls = [None if rnd.random() < 0.1 else u for u in urls]
        presorted = rnd.random() < 0.5
        if presorted:
            urls = [urls[i] for i in np.argsort(url_ranks(urls), kind="stable")]
        key = (lambda u: u[::-1]) if trial % 5 == 4 and not presorted else None
        r = UrlRanker(urls, key, presorted=presorted)
        np.testing.assert_array_equal(r.ranks(), url_ranks(urls, key))
        for _ in range(5):
            changed = {rnd.randrange(len(urls)): (None if rnd.random() < 0.3 else "v%05d" % rnd.randint(0, 99999))
                       for _ in range(rnd.randint(0, 4)) if urls}
            appended = [None if rnd.random() < 0.1 else "w%05d" % rnd.randint(0, 99999) for _ in range(rnd.randint(0, 3))]
            for i, u in changed.items():
                urls[i] = u
            urls += appended
            r.apply(changed, appended)
            want = url_ranks(urls, key)
            np.testing.assert_array_equal(r.ranks(), want)
            if r.identity:
                np.testing.assert_array_equal(want, np.arange(len(urls)))
