"""SQL-engine stand-in for the reference's PostgreSQL: stdlib sqlite3 executing statements equivalent
to the reference's (app/batch/bm25_stats.py:9-27, app/api/search_service.py:13-101) over the
reference's table shapes (alembic/versions/0001_initial_schema.py:33-51,65-74,110-115,
0007_unify_news_tokens.py:19-27).

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): used by tests/test_oracle_sqlite.py to
cross-check the C oracle's GROUP BY / COUNT(DISTINCT) / NULL / LEFT JOIN + COALESCE / ORDER BY ..
LIMIT semantics, and by the cpu_baseline leg of bench.py as the "sqlite stand-in, not Postgres"
timing tier.  It does not reproduce PostgreSQL's NUMERIC arithmetic or its planner."""
import sqlite3
import time

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



def run_stats_job(db):
    """The statements of the statistics job, in the reference's order."""
    avg = db.execute("SELECT AVG(word_count) FROM documents WHERE status='done'").fetchone()[0] or 0.0
    total = db.execute("SELECT COUNT(*) FROM documents WHERE status='done'").fetchone()[0] or 1
    db.execute("DELETE FROM term_statistics")
    db.execute(STATS_SQL, (total, avg))
    db.commit()
    return total, avg


def time_standin(c, queries, ks=(10, 210)):
    """Loads corpus `c` (oracle.gen_corpus) and times the statistics job and the web scoring
    statement for `queries` at every k of `ks` -> dict of seconds (one sqlite thread = one core)."""
    db, _, _ = load_db(c)
    n_rows = len(c["term_id"])
    t0 = time.perf_counter()
    run_stats_job(db)
    t_build = time.perf_counter() - t0
    out = {"rows": n_rows, "build_s": t_build, "build_rows_per_s": n_rows / t_build, "queries": len(queries)}
    for k in ks:
        t0 = time.perf_counter()
        for q in queries:
            q = [int(t) for t in q]
            db.execute(WEB_SQL.format(ph=",".join("?" * len(q))), [term_text(t) for t in q] + [k]).fetchall()
        dt = time.perf_counter() - t0
        out[f"top{k}_queries_per_s"] = len(queries) / dt
    db.close()
    return out
