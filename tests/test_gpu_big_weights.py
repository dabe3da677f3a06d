"""GPU parity of searches over an index that holds posting weights above 16 bits.

`tokens.frequency` is an INT and the scoring statement puts no bound on it (search_service.py:19-27).  A
weight of 0xFFFF tenths or more (tf >= 2048 in a title row, >= 6554 in a body or news row) does not fit the
16-bit accumulator fields of the stepwise scoring kernels, so an index that holds one sends every query to
the general kernel.  These tests search such an index (1-, 3- and 9-term queries, both sources, k = 10 and
210, with and without url ranks), an index that gets such a row through an update, and the same after
compaction, all bit for bit against the CPU oracle's posting-level scores."""
import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import assert_build_matches, gpu_build, oracle_index, small_corpus

pytestmark = pytest.mark.gpu

N_DOCS, N_TERMS, N_ARTICLES = 2500, 5000, 300


def _with_big_tf(c, seed, n_big=40):
    """Copies of the row arrays in which a few rows of frequent terms carry huge frequencies; -> (corpus, terms hit)."""
    rng = np.random.default_rng(seed)
    tf = c["tf"].copy()
    field = c["meta"] >> 2
    news = (c["meta"] & 3) == se.SRC_NEWS
    frequent = np.flatnonzero(c["term_id"] < 60)
    hit = []
    for kind, rows in (("title", frequent[(field[frequent] == 1) & ~news[frequent]]),
                       ("body", frequent[(field[frequent] == 4) & ~news[frequent]]),
                       ("news", frequent[news[frequent]])):
        pick = rng.choice(rows, size=min(n_big, len(rows)), replace=False)
        lo = 2048 if kind == "title" else 6554
        tf[pick] = rng.integers(lo, 40 * lo, len(pick)).astype(np.uint32)
        tf[pick[0]] = lo            # exactly the first weight that does not fit (0xFFFF tenths would be lo - 1 ... lo)
        tf[pick[1]] = (1 << 22) + 3
        hit.extend(c["term_id"][pick].tolist())
    return dict(c, tf=tf), sorted(set(hit))


def _queries(rng, hit, n_terms, tpq, n):
    qs = []
    for _ in range(n):
        some = rng.choice(hit, size=min(len(hit), max(1, tpq // 2)), replace=False)
        rest = rng.integers(0, n_terms, tpq)
        q = np.unique(np.concatenate([some, rest]))[:tpq] if tpq > 1 else some[:1]
        qs.append(rng.permutation(q).astype(np.uint32))
    return qs


def _check_searches(ix, orc, hit, n_terms, n_docs, n_articles, seed):
    rng = np.random.default_rng(seed)
    ix.set_idf(orc.idf)   # the same idf bits on both sides
    ranks = {se.SRC_WEB: rng.permutation(n_docs).astype(np.uint32), se.SRC_NEWS: rng.permutation(n_articles).astype(np.uint32)}
    for with_rank in (False, True):
        for source in (se.SRC_WEB, se.SRC_NEWS):
            ix.set_url_rank(source, ranks[source] if with_rank else None)
            for tpq in (1, 3, 9):
                qs = _queries(rng, hit, n_terms, tpq, 12)
                for k in (10, 210):
                    out = ix.search(qs, k, source=source)
                    for qi, q in enumerate(qs):
                        ru, rs, rm = orc.search(q, k, source=source, url_rank=ranks[source] if with_rank else None, level="csr")
                        n = int(out["count"][qi])
                        assert n == len(ru), (tpq, k, source, with_rank, qi)
                        np.testing.assert_array_equal(out["unit"][qi, :n], ru)
                        np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
                        np.testing.assert_array_equal(out["matched"][qi, :n], rm)
    ix.set_url_rank(se.SRC_WEB, None)
    ix.set_url_rank(se.SRC_NEWS, None)


def test_search_over_an_index_with_weights_above_16_bits(gpu_index_factory):
    c, hit = _with_big_tf(small_corpus(n_docs=N_DOCS, n_terms=N_TERMS, n_articles=N_ARTICLES), seed=1)
    orc = oracle_index(c, word_count=None)
    assert int(orc.post_weight.max()) >= 0xFFFF
    ix = gpu_index_factory()
    gpu_build(ix, c, word_count=None)
    assert_build_matches(ix, orc)
    _check_searches(ix, orc, hit, N_TERMS, N_DOCS, N_ARTICLES, seed=2)
    orc.close()


def test_update_that_brings_a_big_weight_and_compaction(gpu_index_factory):
    """The base segment holds small weights only; the update segment gets the big ones."""
    base = small_corpus(n_docs=N_DOCS, n_terms=N_TERMS, n_articles=N_ARTICLES)
    assert int(oracle_index(base).post_weight.max()) < 0xFFFF
    ix = gpu_index_factory()
    gpu_build(ix, base)
    # docs 40..99 and articles 10..29 are written again, some of their rows with huge frequencies
    changed_docs = np.arange(40, 100, dtype=np.uint32)
    changed_arts = np.arange(10, 30, dtype=np.uint32)
    src_news = (base["meta"] & 3) == se.SRC_NEWS
    is_changed = (~src_news & np.isin(base["unit_id"], changed_docs)) | (src_news & np.isin(base["unit_id"], changed_arts))
    upd = {k: base[k][is_changed].copy() for k in ("term_id", "unit_id", "tf", "meta")}
    big, hit = _with_big_tf(dict(base, **upd), seed=3, n_big=15)
    upd["tf"] = big["tf"]
    final = dict(base, tf=base["tf"].copy())
    final["tf"][is_changed] = upd["tf"]
    orc = oracle_index(final)
    assert int(orc.post_weight.max()) >= 0xFFFF
    ix.update(upd["term_id"], upd["unit_id"], upd["tf"], upd["meta"], N_TERMS, N_DOCS, N_ARTICLES, base["word_count"], None,
              changed_docs=changed_docs, changed_articles=changed_arts, unit_major=True)
    assert ix.info().n_segments == 2
    assert_build_matches(ix, orc)
    _check_searches(ix, orc, hit, N_TERMS, N_DOCS, N_ARTICLES, seed=4)
    ix.compact()
    assert ix.info().n_segments == 1
    assert_build_matches(ix, orc)
    _check_searches(ix, orc, hit, N_TERMS, N_DOCS, N_ARTICLES, seed=5)
    orc.close()
