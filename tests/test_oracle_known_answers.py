"""Hand-worked cases pinning the CPU oracle to the reference's SQL semantics
(app/batch/bm25_stats.py:9-27, app/api/search_service.py:13-101).  No GPU."""
import math

import numpy as np
import pytest

from oracle import oracle as O

W, N_ = O.SRC_WEB, O.SRC_NEWS


def rows(spec):
    """spec: list of (term, unit, tf, source_type, field)."""
    a = np.array(spec, dtype=np.int64).reshape(-1, 5)
    return (a[:, 0].astype(np.uint32), a[:, 1].astype(np.uint32), a[:, 2].astype(np.uint32),
            (a[:, 3] | (a[:, 4] << 2)).astype(np.uint8))


def build(spec, n_terms, n_docs, n_articles=0, word_count=None, doc_flags=None):
    t, u, f, m = rows(spec)
    return O.OracleIndex(t, u, f, m, n_terms, n_docs, n_articles, word_count, doc_flags)


def test_df_counts_distinct_web_docs_only():
    # term 0: doc 0 in title + description + body (3 rows), doc 2 body, one news row -> df = 2
    # term 1: news only -> a term_statistics row with df = 0
    spec = [(0, 0, 2, W, 1), (0, 0, 1, W, 2), (0, 0, 5, W, 4), (0, 2, 1, W, 4), (0, 0, 7, N_, 4), (1, 1, 3, N_, 4)]
    ix = build(spec, n_terms=3, n_docs=3, n_articles=2, word_count=np.array([10, 20, 30], np.uint32))
    assert ix.df.tolist() == [2, 0, 0]
    assert ix.term_rows.tolist() == [5, 1, 0]          # the reference inserts rows only for terms 0 and 1
    assert ix.n_docs_stat == 3 and ix.avgdl == 20.0
    n = 3
    assert ix.idf[0] == pytest.approx(math.log((n - 2 + 0.5) / (2 + 0.5) + 1), rel=1e-15)
    assert ix.idf[1] == pytest.approx(math.log(2 * n + 2), rel=1e-15)   # df = 0: ln((N + .5)/.5 + 1) = ln(2N + 2)
    # CSR: web postings then news postings per term, one posting per (term, unit)
    assert ix.row_ptr.tolist() == [0, 3, 4, 4] and ix.news_ptr.tolist() == [2, 3, 4]
    assert ix.post_unit.tolist() == [0, 2, 0, 1]
    # weights in tenths, exact: 3.2 / 1.7 / 1.0 -> 32 / 17 / 10; news: no field weight
    assert ix.post_weight.tolist() == [2 * 32 + 1 * 17 + 5 * 10, 10, 70, 30]
    assert ix.doc_len.tolist() == [5, 0, 1]            # body rows only


def test_score_formula_and_matched_terms():
    spec = [(0, 0, 2, W, 1), (0, 0, 1, W, 2), (0, 0, 5, W, 4), (1, 0, 4, W, 4), (1, 1, 1, W, 9), (0, 1, 1, W, 4)]
    ix = build(spec, n_terms=2, n_docs=2)
    idf = ix.idf
    u, s, m = ix.search([0, 1], 10)
    # doc 0: term 0 in three fields + term 1; doc 1: both terms once (field 9 falls in ELSE 1.0)
    s0 = ((2 * idf[0]) * 3.2 + (1 * idf[0]) * 1.7 + (5 * idf[0]) * 1.0) + (4 * idf[1]) * 1.0
    s1 = (1 * idf[0]) * 1.0 + (1 * idf[1]) * 1.0
    assert u.tolist() == [0, 1] and m.tolist() == [2, 2]
    assert s[0] == pytest.approx(s0, rel=1e-15) and s[1] == pytest.approx(s1, rel=1e-15)
    # the merged-posting arithmetic (exact integer weights) agrees with the row formula to rounding
    u2, s2, m2 = ix.search([0, 1], 10, level="csr")
    assert u2.tolist() == [0, 1] and np.allclose(s2, s, rtol=1e-14, atol=0)
    # ANY($1) with a repeated / unknown term
    u3, s3, m3 = ix.search([1, 1, 99], 10)
    assert u3.tolist() == [0, 1] and m3.tolist() == [1, 1]


def test_no_done_docs_gives_n_one_and_avgdl_zero():
    spec = [(0, 0, 1, W, 4)]
    ix = build(spec, n_terms=1, n_docs=1, word_count=np.array([9], np.uint32), doc_flags=np.array([0], np.uint8))
    assert ix.n_docs_stat == 1 and ix.avgdl == 0.0     # `or 1`, `or 0.0`
    assert ix.df.tolist() == [1]                       # tokens of a not-done doc still count in df
    assert ix.totals.n_postings == 0                   # but the doc cannot be returned (d.status = 'done')
    u, s, m = ix.search([0], 5)
    assert len(u) == 0


def test_null_word_count_is_skipped_by_avg():
    spec = [(0, d, 1, W, 4) for d in range(4)]
    flags = np.array([1, 1 | 2, 1, 0], np.uint8)       # doc 1: word_count NULL, doc 3: not done
    ix = build(spec, n_terms=1, n_docs=4, word_count=np.array([10, 1000, 30, 500], np.uint32), doc_flags=flags)
    assert ix.n_docs_stat == 3                         # COUNT(*) counts the NULL row
    assert ix.avgdl == 20.0                            # AVG skips it


def test_missing_idf_defaults_to_one():
    spec = [(0, 0, 3, W, 1), (1, 1, 2, W, 4)]
    ix = build(spec, n_terms=2, n_docs=2)
    ix.set_idf(np.array([7.0, 5.0]), np.array([0, 1], np.uint8))   # term 0 absent from term_statistics
    u, s, m = ix.search([0, 1], 10)
    assert dict(zip(u.tolist(), s.tolist())) == {1: 2 * 5.0 * 1.0, 0: (3 * 1.0) * 3.2}


def test_ties_break_on_url_order_and_limit():
    spec = [(0, d, 1, W, 4) for d in range(6)]
    ix = build(spec, n_terms=1, n_docs=6)
    u, s, m = ix.search([0], 4)
    assert u.tolist() == [0, 1, 2, 3] and len(set(s.tolist())) == 1       # default: url order = doc id order
    rank = np.array([5, 4, 3, 2, 1, 0], np.uint32)
    u, s, m = ix.search([0], 4, url_rank=rank)
    assert u.tolist() == [5, 4, 3, 2]


def test_news_scoring_ignores_field_and_uses_web_idf():
    spec = [(0, 0, 2, N_, 1), (0, 1, 1, N_, 4), (0, 0, 1, W, 4), (0, 1, 1, W, 4), (0, 2, 1, W, 4)]
    ix = build(spec, n_terms=1, n_docs=4, n_articles=2)
    idf = math.log((4 - 3 + 0.5) / (3 + 0.5) + 1)
    assert ix.idf[0] == pytest.approx(idf, rel=1e-15)
    u, s, m = ix.search([0], 10, source=N_)
    assert u.tolist() == [0, 1]
    assert s[0] == pytest.approx(2 * idf, rel=1e-15) and s[1] == pytest.approx(idf, rel=1e-15)   # field 1 not weighted


def test_idf_is_accurate_when_df_is_close_to_n():
    # PostgreSQL evaluates LN in NUMERIC; a naive double log(x + 1) loses digits when df ~ N
    from decimal import Decimal, getcontext

    getcontext().prec = 60
    n = 1_000_000
    spec = [(0, d, 1, W, 4) for d in range(n - 3)]
    ix = build(spec, n_terms=1, n_docs=n)
    df = n - 3
    exact = ((Decimal(n) - df + Decimal("0.5")) / (df + Decimal("0.5")) + 1).ln()
    assert abs(Decimal(float(ix.idf[0])) - exact) / exact < Decimal("4e-16")


def test_out_of_range_rows_are_rejected():
    with pytest.raises(ValueError):
        build([(5, 0, 1, W, 4)], n_terms=5, n_docs=1)
    with pytest.raises(ValueError):
        build([(0, 1, 1, W, 4)], n_terms=5, n_docs=1)
    with pytest.raises(ValueError):
        build([(0, 0, 1, 3, 4)], n_terms=5, n_docs=1)
