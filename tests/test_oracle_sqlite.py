"""Cross-check of the CPU oracle against an SQL engine: stdlib sqlite3 executes statements
equivalent to the reference's (app/batch/bm25_stats.py:9-27, app/api/search_service.py:13-101)
over the reference's table shapes (alembic/versions/0001_initial_schema.py:33-51,65-74,110-115,
0007_unify_news_tokens.py:19-27).  sqlite is a stand-in for PostgreSQL, which is not available
here: it checks GROUP BY / COUNT(DISTINCT) / NULL / LEFT JOIN + COALESCE / ORDER BY .. LIMIT
semantics, not PostgreSQL's NUMERIC arithmetic.  No GPU."""
import numpy as np
import pytest

from oracle import oracle as O

from helpers import assert_topk_valid, small_corpus

from oracle.sqlite_standin import NEWS_SQL, WEB_SQL, load_db, term_text


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
