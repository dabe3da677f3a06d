"""Cross-check of the CPU oracle against an SQL engine: stdlib sqlite3 executes statements
equivalent to the reference's (app/batch/bm25_stats.py:9-27, app/api/search_service.py:13-101)
over the reference's table shapes (alembic/versions/0001_initial_schema.py:33-51,65-74,110-115,
0007_unify_news_tokens.py:19-27).  sqlite is a stand-in for PostgreSQL, which is not available
here: it checks GROUP BY / COUNT(DISTINCT) / NULL / LEFT JOIN + COALESCE / ORDER BY .. LIMIT
semantics, not PostgreSQL's NUMERIC arithmetic.  No GPU."""
import sqlite3

import numpy as np
import pytest

from oracle import oracle as O

from helpers import assert_topk_valid, small_corpus

DDL = """
CREATE TABLE documents (id INTEGER PRIMARY KEY, url TEXT, title TEXT, description TEXT, word_count INTEGER, status TEXT);
CREATE TABLE news_articles (url TEXT PRIMARY KEY, title TEXT, description TEXT);
CREATE TABLE tokens (doc_id INTEGER, term TEXT, field INTEGER, frequency INTEGER, article_url TEXT, source_type INTEGER DEFAULT 1);
CREATE TABLE term_statistics (term TEXT PRIMARY KEY, doc_frequency INTEGER, idf REAL, avg_doc_len REAL);
CREATE INDEX idx_tokens_term ON tokens(term);
"""

STATS_SQL = """
INSERT INTO term_statistics(term, doc_frequency, idf, avg_doc_len)
SELECT t.term, COUNT(DISTINCT t.doc_id),
       LN((? - COUNT(DISTINCT t.doc_id) + 0.5) / (COUNT(DISTINCT t.doc_id) + 0.5) + 1), ?
FROM tokens t GROUP BY t.term
"""

WEB_SQL = """
WITH scored AS (
  SELECT d.id AS id, d.url AS url,
         SUM(t.frequency * COALESCE(ts.idf, 1.0) * CASE t.field WHEN 1 THEN 3.2 WHEN 2 THEN 1.7 ELSE 1.0 END) AS token_score,
         COUNT(DISTINCT t.term) AS matched_terms
  FROM tokens t JOIN documents d ON d.id = t.doc_id
  LEFT JOIN term_statistics ts ON ts.term = t.term
  WHERE d.status = 'done' AND t.term IN ({ph})
  GROUP BY d.id, d.title, d.description, d.url)
SELECT id, token_score, matched_terms FROM scored ORDER BY token_score DESC, url ASC LIMIT ?
"""

NEWS_SQL = """
SELECT na.url, SUM(t.frequency * COALESCE(ts.idf, 1.0)) AS token_score, COUNT(DISTINCT t.term) AS matched_terms
FROM tokens t JOIN news_articles na ON na.url = t.article_url
LEFT JOIN term_statistics ts ON ts.term = t.term
WHERE t.source_type = 2 AND t.term IN ({ph})
GROUP BY na.title, na.description, na.url
ORDER BY token_score DESC, na.url ASC LIMIT ?
"""


def term_text(t):
    return "t%07d" % t


def doc_url(d):
    return "https://s.test/%012d" % d      # byte order == doc id order


def art_url(a):
    return "https://n.test/%012d" % a


def load_db(c, doc_flags=None, stale_terms=()):
    db = sqlite3.connect(":memory:")
    db.executescript(DDL)
    nd = c["n_docs"]
    for d in range(nd):
        fl = 1 if doc_flags is None else int(doc_flags[d])
        wc = None if fl & 2 else int(c["word_count"][d])
        db.execute("INSERT INTO documents VALUES (?,?,?,?,?,?)",
                   (d, doc_url(d), "", "", wc, "done" if fl & 1 else "processing"))
    for a in range(c["n_articles"]):
        db.execute("INSERT INTO news_articles VALUES (?,?,?)", (art_url(a), "", ""))
    rows = []
    for t, u, f, m in zip(c["term_id"].tolist(), c["unit_id"].tolist(), c["tf"].tolist(), c["meta"].tolist()):
        st, field = m & 3, m >> 2
        if st == 1:
            rows.append((u, term_text(t), field, f, None, 1))
        else:
            rows.append((None, term_text(t), field, f, art_url(u), 2))      # news rows carry no doc id
    db.executemany("INSERT INTO tokens VALUES (?,?,?,?,?,?)", rows)
    # the statistics job, statement for statement
    avg = db.execute("SELECT AVG(word_count) FROM documents WHERE status='done'").fetchone()[0] or 0.0
    total = db.execute("SELECT COUNT(*) FROM documents WHERE status='done'").fetchone()[0] or 1
    db.execute("DELETE FROM term_statistics")
    db.execute(STATS_SQL, (total, avg))
    for t in stale_terms:
        db.execute("DELETE FROM term_statistics WHERE term = ?", (term_text(t),))
    db.commit()
    return db, total, avg


@pytest.fixture(scope="module")
def world():
    c = small_corpus(n_docs=250, n_terms=3000, n_articles=60, body_min=30, body_max=90)
    rng = np.random.default_rng(5)
    flags = np.where(rng.random(250) < 0.85, 1, 0).astype(np.uint8)
    flags |= np.where(rng.random(250) < 0.1, 2, 0).astype(np.uint8)
    stale = [3, 17, 40]
    db, total, avg = load_db(c, flags, stale)
    orc = O.OracleIndex(c["term_id"], c["unit_id"], c["tf"], c["meta"], c["n_terms"], c["n_docs"], c["n_articles"],
                        c["word_count"], flags)
    return c, db, total, avg, orc, stale


def test_statistics_job_matches_sql(world):
    c, db, total, avg, orc, stale = world
    assert orc.n_docs_stat == total
    assert orc.avgdl == pytest.approx(avg, rel=1e-15)
    got = {term: (df, idf, al) for term, df, idf, al in db.execute("SELECT * FROM term_statistics")}
    df, idf, tr = orc.df, orc.idf, orc.term_rows
    present = {term_text(t) for t in range(c["n_terms"]) if tr[t] > 0} - {term_text(t) for t in stale}
    assert set(got) == present                       # one row per term that has token rows, df = 0 rows included
    assert any(v[0] == 0 for v in got.values())      # a news-only term exists in this corpus
    for t in range(c["n_terms"]):
        if term_text(t) in got:
            g = got[term_text(t)]
            assert g[0] == df[t]
            assert g[1] == pytest.approx(idf[t], rel=1e-13)
            assert g[2] == pytest.approx(orc.avgdl, rel=1e-15)


@pytest.mark.parametrize("k", [5, 40, 500])
def test_scoring_sql_matches_oracle(world, k):
    c, db, total, avg, orc, stale = world
    present = np.ones(c["n_terms"], np.uint8)
    present[stale] = 0
    idf_tab = np.zeros(c["n_terms"])
    for term, df, idf, al in db.execute("SELECT * FROM term_statistics"):
        idf_tab[int(term[1:])] = idf
    orc.set_idf(idf_tab, present)                    # COALESCE(ts.idf, 1.0) for the deleted rows
    qs = O.gen_queries(9, c["n_terms"], O.zipf_cdf(c["n_terms"]), 1, 30, 3).tolist() + [[3, 17, 5], [40], [2999, 3]]
    for q in qs:
        ph = ",".join("?" * len(q))
        web = db.execute(WEB_SQL.format(ph=ph), [term_text(t) for t in q] + [100000]).fetchall()
        u, s, m = orc.search(np.array(q, np.uint32), k, level="rows")
        assert_topk_valid(u, s, m, [r[0] for r in web], [r[1] for r in web], [r[2] for r in web], k, rtol=1e-12)
        news = db.execute(NEWS_SQL.format(ph=ph), [term_text(t) for t in q] + [100000]).fetchall()
        u, s, m = orc.search(np.array(q, np.uint32), k, source=O.SRC_NEWS, level="rows")
        assert_topk_valid(u, s, m, [int(r[0][-12:]) for r in news], [r[1] for r in news], [r[2] for r in news], k,
                          rtol=1e-12)
    orc.set_idf(orc.idf)
