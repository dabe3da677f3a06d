"""Independent numpy restatement of the integer outputs (np.unique over (term, source, unit))
checked against the C oracle, including unsorted input.  No GPU."""
import numpy as np

from oracle import oracle as O

from helpers import oracle_index, small_corpus


def numpy_build(c, doc_flags=None):
    term, unit, tf, meta = (c[k].astype(np.int64) for k in ("term_id", "unit_id", "tf", "meta"))
    st, field = meta & 3, meta >> 2
    V, nd = c["n_terms"], c["n_docs"]
    key = (term << 40) | ((st - 1) << 39) | unit
    uniq, first = np.unique(key, return_index=True)
    u_term, u_news, u_unit = uniq >> 40, (uniq >> 39) & 1, uniq & ((1 << 39) - 1)
    df = np.bincount(u_term[u_news == 0], minlength=V)
    done = np.ones(nd, bool) if doc_flags is None else (doc_flags & 1).astype(bool)
    kept = (u_news == 1) | done[np.minimum(u_unit, max(nd - 1, 0))]
    row_ptr = np.concatenate([[0], np.cumsum(np.bincount(u_term[kept], minlength=V))])
    web_kept = np.bincount(u_term[kept & (u_news == 0)], minlength=V)
    doc_len = np.bincount(unit[(st == 1) & (field == 4)], weights=tf[(st == 1) & (field == 4)], minlength=nd)
    # weights in tenths per (term, source, unit): exact integers
    w10 = np.where(st == 1, tf * np.select([field == 1, field == 2], [32, 17], 10), tf * 10)
    sums = np.bincount(np.searchsorted(uniq, key), weights=w10, minlength=len(uniq)).astype(np.int64)
    return dict(df=df, row_ptr=row_ptr, news_ptr=row_ptr[:-1] + web_kept, post_unit=u_unit[kept],
                post_weight=sums[kept], doc_len=doc_len.astype(np.int64),
                term_rows=np.bincount(term, minlength=V))


def check(c, doc_flags=None):
    orc = oracle_index(c, doc_flags=doc_flags)
    ref = numpy_build(c, doc_flags)
    np.testing.assert_array_equal(orc.df, ref["df"])
    np.testing.assert_array_equal(orc.term_rows, ref["term_rows"])
    np.testing.assert_array_equal(orc.row_ptr, ref["row_ptr"])
    np.testing.assert_array_equal(orc.news_ptr, ref["news_ptr"])
    np.testing.assert_array_equal(orc.post_unit, ref["post_unit"])
    np.testing.assert_array_equal(orc.post_weight, ref["post_weight"])
    np.testing.assert_array_equal(orc.doc_len, ref["doc_len"])
    assert int(orc.totals.n_postings) == len(ref["post_unit"])


def test_c_oracle_matches_numpy():
    check(small_corpus(n_docs=200, n_terms=700, n_articles=40, body_min=20, body_max=80))


def test_c_oracle_matches_numpy_with_status_and_shuffle():
    c = small_corpus(n_docs=150, n_terms=400, n_articles=30, body_min=20, body_max=60)
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(c["term_id"]))
    for k in ("term_id", "unit_id", "tf", "meta"):
        c[k] = np.ascontiguousarray(c[k][perm])
    flags = np.where(rng.random(150) < 0.7, 1, 0).astype(np.uint8)
    check(c, flags)


def test_generator_contract():
    c = small_corpus(n_docs=120, n_terms=5000, n_articles=15)
    meta, unit, term, tf = c["meta"], c["unit_id"], c["term_id"], c["tf"]
    st, field = meta & 3, meta >> 2
    key = (st.astype(np.int64) << 32) | unit
    assert (np.diff(key) >= 0).all()                                   # unit-major: web docs, then articles
    assert set(np.unique(field[st == 1])) <= {1, 2, 4} and set(np.unique(field[st == 2])) == {4}
    # rows are unique per (unit, field, term), as the crawler's Counter rows are (worker.py:230-239)
    full = (key << 24 | field.astype(np.int64) << 20).astype(object) * (1 << 20) + term.astype(object)
    assert len(set(full.tolist())) == len(term)
    body = np.bincount(unit[(st == 1) & (field == 4)], weights=tf[(st == 1) & (field == 4)], minlength=120)
    np.testing.assert_array_equal(body.astype(np.uint32), c["word_count"])   # word_count = body draws
    assert 100 <= c["word_count"].min() and c["word_count"].max() <= 500
    title = np.bincount(unit[(st == 1) & (field == 1)], weights=tf[(st == 1) & (field == 1)], minlength=120)
    assert (title == 8).all()
    # shards regenerate the same documents
    part = O.gen_corpus(O.make_spec(n_terms=5000, n_docs=40, doc_begin=60))
    sel = (st == 1) & (unit >= 60) & (unit < 100)
    np.testing.assert_array_equal(part["term_id"], term[sel])
    np.testing.assert_array_equal(part["unit_id"] + 60, unit[sel])
    np.testing.assert_array_equal(part["tf"], tf[sel])
