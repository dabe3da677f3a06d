"""GPU parity: batched scoring + top-k through the C ABI against the CPU oracle."""
import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import assert_topk_valid, gpu_build, oracle_index, small_corpus

pytestmark = pytest.mark.gpu


def _queries(n_terms, n_queries, tpq=3, min_rank=1, seed=20260228):
    return O.gen_queries(seed, n_terms, O.zipf_cdf(n_terms), min_rank, n_queries, tpq)


@pytest.fixture(scope="module")
def built(gpu_index_factory):
    c = small_corpus(n_docs=3000, n_terms=6000, n_articles=400)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    ix.set_idf(orc.idf)  # same idf bits on both sides: scores from the integer postings are then bit-exact
    return c, orc, ix


@pytest.mark.parametrize("k", [1, 10, 48, 49, 64, 210, 700, 2000])
@pytest.mark.parametrize("source", [se.SRC_WEB, se.SRC_NEWS])
def test_topk_matches_reference_formula(built, k, source):
    c, orc, ix = built
    qs = _queries(c["n_terms"], 40, 3, min_rank=1 if k >= 64 else 20)
    out = ix.search(qs, k, source=source)
    for qi, q in enumerate(qs):
        n = int(out["count"][qi])
        ru, rs, rm = orc.search(q, max(k, 4096), source=source, level="rows")  # the SQL formula, row by row
        assert_topk_valid(out["unit"][qi, :n], out["score"][qi, :n], out["matched"][qi, :n], ru, rs, rm, k)
        assert (out["unit"][qi, n:] == se.PAD_UNIT).all()


@pytest.mark.parametrize("k", [10, 210])
def test_topk_bit_exact_against_posting_level_oracle(built, k):
    c, orc, ix = built
    qs = _queries(c["n_terms"], 60, 3, min_rank=5)
    out = ix.search(qs, k)
    for qi, q in enumerate(qs):
        ru, rs, rm = orc.search(q, k, level="csr")
        n = int(out["count"][qi])
        assert n == len(ru)
        np.testing.assert_array_equal(out["unit"][qi, :n], ru)
        np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
        np.testing.assert_array_equal(out["matched"][qi, :n], rm)


def test_query_edge_cases(built):
    c, orc, ix = built
    V = c["n_terms"]
    qs = [[], [V + 5], [3, 3, 3], [7, V + 1, 7, 11], list(range(40, 70)), list(range(0, 80)), list(range(100, 330))]
    out = ix.search(qs, 10)
    assert out["count"][0] == 0 and out["count"][1] == 0
    for qi in (2, 3, 4, 5, 6):      # the last two: more lists than one wavefront has lanes
        ru, rs, rm = orc.search(np.array(qs[qi], np.uint32), 10, level="csr")
        n = int(out["count"][qi])
        np.testing.assert_array_equal(out["unit"][qi, :n], ru)
        np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
        np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    assert out["matched"][2, 0] == 1  # a repeated term counts once
    with pytest.raises(se.Sebm25Error):
        ix.search([[1]], 0)
    with pytest.raises(se.Sebm25Error):
        ix.search([[1]], 5000)


@pytest.mark.parametrize("n_q_terms", [256, 257, 300, 600, 1500, 6000])
def test_queries_with_hundreds_of_terms(built, n_q_terms):
    """`t.term = ANY($1)` takes any number of terms (search_service.py:33): a pasted paragraph is a query.  Beyond 256
    lists the scoring kernel keeps its cursors in global memory; results stay bit for bit those of the oracle."""
    c, orc, ix = built
    rng = np.random.default_rng(n_q_terms)
    qs = [rng.choice(c["n_terms"], size=n_q_terms, replace=False).astype(np.uint32) for _ in range(3)]
    qs.append(np.arange(n_q_terms, dtype=np.uint32))          # the most frequent terms: long lists
    rank = rng.permutation(c["n_docs"]).astype(np.uint32)
    for source, k, with_rank in ((se.SRC_WEB, 10, False), (se.SRC_WEB, 210, True), (se.SRC_NEWS, 10, False), (se.SRC_WEB, 2000, False)):
        ix.set_url_rank(se.SRC_WEB, rank if with_rank else None)
        out = ix.search(qs, k, source=source)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, source=source, url_rank=rank if with_rank and source == se.SRC_WEB else None, level="csr")
            n = int(out["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    ix.set_url_rank(se.SRC_WEB, None)


def test_missing_idf_is_one_and_url_rank_breaks_ties(built):
    c, orc, ix = built
    V = c["n_terms"]
    present = np.ones(V, np.uint8)
    present[::3] = 0
    idf = orc.idf.copy()
    try:
        ix.set_idf(idf, present)
        orc.set_idf(idf, present)
        rng = np.random.default_rng(11)
        rank = rng.permutation(c["n_docs"]).astype(np.uint32)
        ix.set_url_rank(se.SRC_WEB, rank)
        qs = _queries(V, 30, 2, min_rank=300)  # rare terms with tf 1: many exact score ties
        out = ix.search(qs, 20)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, 20, url_rank=rank, level="csr")
            n = int(out["count"][qi])
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
    finally:
        ix.set_url_rank(se.SRC_WEB, None)
        ix.set_idf(orc.idf)
        orc.set_idf(orc.idf)


@pytest.mark.parametrize("tpq", [1, 3, 4, 6, 10])
def test_url_ranks_with_every_size_of_candidate_storage(built, tpq):
    """An explicit rank array (ties break on it, not on the unit id) through every kernel variant: candidates in
    registers, in the LDS buffers whose tie-break keys are read when the buffer is cut, and the general kernel;
    queries of rare terms (many exact score ties at the cut) and of frequent ones."""
    c, orc, ix = built
    V = c["n_terms"]
    rng = np.random.default_rng(100 + tpq)
    rank = rng.permutation(c["n_docs"]).astype(np.uint32)
    try:
        ix.set_url_rank(se.SRC_WEB, rank)
        qs = _queries(V, 12, tpq, min_rank=300, seed=5 + tpq) + _queries(V, 12, tpq, min_rank=2, seed=50 + tpq)
        for k in (10, 48, 60, 128, 210, 300, 700, 2000):
            out = ix.search(qs, k)
            for qi, q in enumerate(qs):
                ru, rs, rm = orc.search(q, k, url_rank=rank, level="csr")
                n = int(out["count"][qi])
                assert n == len(ru)
                np.testing.assert_array_equal(out["unit"][qi, :n], ru, err_msg=f"k {k} query {qi}")
                np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
                np.testing.assert_array_equal(out["matched"][qi, :n], rm)
        # the news segment with a rank array of its own
        rank_n = rng.permutation(c["n_articles"]).astype(np.uint32)
        ix.set_url_rank(se.SRC_NEWS, rank_n)
        for k in (10, 210):
            out = ix.search(qs, k, source=se.SRC_NEWS)
            for qi, q in enumerate(qs):
                ru, rs, rm = orc.search(q, k, source=se.SRC_NEWS, url_rank=rank_n, level="csr")
                n = int(out["count"][qi])
                assert n == len(ru)
                np.testing.assert_array_equal(out["unit"][qi, :n], ru, err_msg=f"news k {k} query {qi}")
                np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
    finally:
        ix.set_url_rank(se.SRC_WEB, None)
        ix.set_url_rank(se.SRC_NEWS, None)


def test_keys_without_an_array_for_the_base_units_order_like_full_ranks(gpu_index_factory):
    """sebm25_set_url_rank_appended: (unit + 1) << shift for the first n_base units, an array only for the rest.  Same
    results as the full rank array that orders the units the same way, for every kernel form and candidate storage."""
    c = small_corpus(n_docs=3000, n_terms=600, n_articles=0, body_min=5, body_max=12)   # few terms, small weights: many equal scores
    ix = gpu_index_factory()
    gpu_build(ix, c)
    n, nb, shift = 3000, 2600, 6
    rng = np.random.default_rng(11)
    # the appended units fall into random gaps between the base units (several into some gaps, a few before the first)
    gaps = np.sort(rng.integers(0, nb + 1, n - nb))
    gaps[:3] = 0
    keys = np.zeros(n - nb, np.uint32)
    seen = {}
    for i, g in enumerate(gaps.tolist()):
        seen[g] = seen.get(g, 0) + 1
        keys[i] = (g << shift) + seen[g]
    keys = keys[rng.permutation(n - nb)]
    full = np.concatenate([(np.arange(nb, dtype=np.uint64) + 1 << shift).astype(np.uint32), keys])
    ranks = np.empty(n, np.uint32)
    ranks[np.argsort(full, kind="stable")] = np.arange(n, dtype=np.uint32)
    qs = [np.array(q, np.uint32) for q in ([0], [1, 2], [0, 3, 5], [2, 4, 6, 9], [1, 2, 3, 4, 5, 6], list(range(12)))]
    for k in (7, 60, 200, 600):
        ix.set_url_rank(se.SRC_WEB, ranks)
        ref = ix.search(qs, k)
        ix.set_url_rank_appended(se.SRC_WEB, nb, shift, keys)
        got = ix.search(qs, k)
        for key in ("count", "unit", "matched"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"k {k} {key}")
        np.testing.assert_array_equal(got["score"].view(np.uint64), ref["score"].view(np.uint64))
    with pytest.raises(se.Sebm25Error):
        ix.set_url_rank_appended(se.SRC_WEB, nb, shift, keys[:-1])       # must cover every unit
    with pytest.raises(se.Sebm25Error):
        ix.set_url_rank_appended(se.SRC_WEB, nb, 20, keys)               # shift too large
    ix.set_url_rank(se.SRC_WEB, None)


def test_sliced_long_queries_equal_unsliced(gpu_index_factory):
    # head terms over a larger corpus force the unit-range split + merge path
    c = small_corpus(n_docs=40_000, n_terms=3000, body_min=20, body_max=60, title_len=2, desc_len=4)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    ix.set_idf(orc.idf)
    # one to five lists: every stepwise kernel form with work items that start inside their lists
    qs = [np.array(q, np.uint32) for q in ([0, 1, 2], [3, 9, 700], [5, 6, 2999], [0, 1], [4, 2500], [0, 1, 2, 3], [8, 7, 1500, 2998],
                                           [2], [0, 3, 5, 7, 11])]
    for k in (10, 300, 700):
        out = ix.search(qs, k)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, level="csr")
            n = int(out["count"][qi])
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)


def test_sharded_build_and_merge_equals_unsharded(gpu_index_factory):
    n_docs, n_art, V = 2400, 300, 5000
    full = small_corpus(n_docs=n_docs, n_terms=V, n_articles=n_art)
    orc = oracle_index(full)
    G = 3
    shards, dfs, tots = [], [], []
    for g in range(G):
        d0, d1 = n_docs * g // G, n_docs * (g + 1) // G
        a0, a1 = n_art * g // G, n_art * (g + 1) // G
        spec = O.make_spec(n_terms=V, n_docs=d1 - d0, n_articles=a1 - a0, doc_begin=d0, article_begin=a0)
        c = O.gen_corpus(spec)
        ix = gpu_index_factory()
        gpu_build(ix, c, local=True)
        ix.set_unit_base(d0, a0)
        df, t = ix.local_stats()
        shards.append(ix)
        dfs.append(df.astype(np.uint64))
        tots.append(t)
    df_global = sum(dfs).astype(np.uint32)  # what the all-reduce computes
    tg = se.DocTotals(sum(t.n_done for t in tots), sum(t.n_wc for t in tots), sum(t.sum_wc for t in tots),
                      sum(t.n_postings for t in tots))
    np.testing.assert_array_equal(df_global, orc.df)
    assert tg.n_postings == orc.totals.n_postings
    for ix in shards:
        ix.finalize(df_global, tg)
        st = ix.stats()
        np.testing.assert_array_equal(st["df"], orc.df)
        assert st["n_docs"] == orc.n_docs_stat and st["avgdl"] == orc.avgdl
        ix.set_idf(orc.idf)
    qs = O.gen_queries(5, V, O.zipf_cdf(V), 10, 25, 3)
    for source in (se.SRC_WEB, se.SRC_NEWS):
        k = 15
        parts = [ix.search(qs, k, source=source, want_tie=True) for ix in shards]
        cat = {key: np.ascontiguousarray(np.stack([p[key] for p in parts])) for key in ("unit", "score", "matched", "tie", "count")}
        merged = shards[0].merge_topk(G, len(qs), k, cat["unit"], cat["score"], cat["matched"], cat["tie"], cat["count"])
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, source=source, level="csr")
            n = int(merged["count"][qi])
            np.testing.assert_array_equal(merged["unit"][qi, :n], ru)
            np.testing.assert_array_equal(merged["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(merged["matched"][qi, :n], rm)

@pytest.mark.parametrize("tpq", [5, 6, 7, 8, 9, 12, 40])
def test_queries_with_five_to_forty_terms(built, tpq):
    """Five to eight lists: the stepwise kernels with two words of accumulator fields per table slot and one chunk
    per list in registers; more than eight: the general kernel.  Same results, every size of candidate storage."""
    c, orc, ix = built
    qs = _queries(c["n_terms"], 30, tpq, min_rank=1, seed=77 + tpq)
    for k in (10, 100, 300):
        out = ix.search(qs, k)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, level="csr")
            n = int(out["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)


@pytest.mark.parametrize("tpq", [1, 2, 4])
@pytest.mark.parametrize("k", [10, 100, 210])
def test_one_two_and_four_list_queries_bit_exact(built, tpq, k):
    """Every list count of the stepwise kernels (the longest list is filtered, with four lists two middle lists
    are filtered against the longer ones), both sources."""
    c, orc, ix = built
    qs = _queries(c["n_terms"], 50, tpq, min_rank=3, seed=77 + tpq)
    for source in (se.SRC_WEB, se.SRC_NEWS):
        out = ix.search(qs, k, source=source)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, source=source, level="csr")
            n = int(out["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)


def test_zero_and_negative_idf_switch_the_filters_off(built):
    """The bounds behind the list filters assume that every term adds to a score.  With idf values of zero or
    below (df above N when docs are not 'done', or an idf table that says so) the results must still be exact."""
    c, orc, ix = built
    V = c["n_terms"]
    rng = np.random.default_rng(5)
    idf = orc.idf.copy()
    idf[rng.random(V) < 0.3] *= -1.0
    idf[rng.random(V) < 0.1] = 0.0
    idf[:40] = -np.abs(idf[:40])                 # the head terms (the longest lists) among them
    try:
        ix.set_idf(idf)
        orc.set_idf(idf)
        for tpq in (1, 2, 3, 4, 6, 8):
            qs = _queries(V, 40, tpq, min_rank=1, seed=900 + tpq)
            for k in (10, 210):
                out = ix.search(qs, k)
                for qi, q in enumerate(qs):
                    ru, rs, rm = orc.search(q, k, level="csr")
                    n = int(out["count"][qi])
                    assert n == len(ru)
                    np.testing.assert_array_equal(out["unit"][qi, :n], ru)
                    np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
                    np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    finally:
        ix.set_idf(orc.idf)
        orc.set_idf(orc.idf)


def test_one_long_list_and_rare_terms(gpu_index_factory):
    """A head term beside terms that occur a handful of times: nearly every posting of the long list is dropped
    by the filter; k below and above the number of documents that hold a rare term."""
    c = small_corpus(n_docs=30_000, n_terms=8000, body_min=20, body_max=60, title_len=3, desc_len=6)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    ix.set_idf(orc.idf)
    have = np.flatnonzero(orc.df >= 1)
    rare = have[np.argsort(orc.df[have], kind="stable")[:12]].astype(np.uint32)     # the rarest terms of the corpus
    qs = [np.array([0, r], np.uint32) for r in rare[:6]] + [np.array([1, rare[6], rare[7]], np.uint32),
                                                            np.array([rare[8], 2, 0, rare[9]], np.uint32), np.array([0], np.uint32)]
    for k in (1, 5, 10, 64, 65, 300):
        out = ix.search(qs, k)
        for qi, q in enumerate(qs):
            ru, rs, rm = orc.search(q, k, level="csr")
            n = int(out["count"][qi])
            assert n == len(ru)
            np.testing.assert_array_equal(out["unit"][qi, :n], ru)
            np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
            np.testing.assert_array_equal(out["matched"][qi, :n], rm)
