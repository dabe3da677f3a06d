"""GPU parity of the build whose last sort pass writes the posting slots itself (unit-major rows, two radix
passes): runs cut by a tile boundary, the cases the pass hands to the separate posting pass, the sorted rows
an update asks for afterwards, and equality with the separate pass on the same input."""
import os

import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import assert_build_matches, gpu_build, oracle_index, small_corpus

pytestmark = pytest.mark.gpu

TILE = 4096


def _dense_runs(n_docs, n_terms, low_bits, per_low, fields=(1, 2, 4), n_articles=0, seed=1):
    """Unit-major rows in which every doc holds `per_low` terms of each of three low-digit classes, each term in
    all `fields` (field-major inside the doc): in the input of the last sort pass the field rows of one
    (term, doc) are separated by the rows of the doc's other terms of the same class, and nearly every tile
    boundary cuts such a group."""
    rng = np.random.default_rng(seed)
    n_low = 1 << low_bits
    n_high = n_terms >> low_bits
    term, unit, tf, meta = [], [], [], []
    for src, n_units in ((se.SRC_WEB, n_docs), (se.SRC_NEWS, n_articles)):
        for u in range(n_units):
            terms = [(int(h) << low_bits) | int(lo) for lo in rng.choice(n_low, size=3, replace=False)
                     for h in rng.choice(n_high, size=min(per_low, n_high), replace=False)]
            for f in (fields if src == se.SRC_WEB else (4,)):
                for t in terms:
                    term.append(t)
                    unit.append(u)
                    tf.append(int(rng.integers(1, 9)))
                    meta.append(se.meta(src, f))
    return dict(term_id=np.array(term, np.uint32), unit_id=np.array(unit, np.uint32), tf=np.array(tf, np.uint32),
                meta=np.array(meta, np.uint8), n_terms=n_terms, n_docs=n_docs, n_articles=n_articles, word_count=None)


def test_synthetic_corpus_takes_the_fused_pass(gpu_index_factory):
    c = small_corpus(n_docs=3000, n_terms=20_000, n_articles=200)
    orc = oracle_index(c)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    info = ix.info()
    assert info.sort_passes == 2 and info.fused_postings == 1
    assert_build_matches(ix, orc)


@pytest.mark.parametrize("per_low,n_articles", [(4, 0), (12, 300), (20, 0)])
def test_runs_cut_by_tile_boundaries(gpu_index_factory, per_low, n_articles):
    # 2048 terms: 11 term bits, passes of 6 and 5 bits; three rows per (term, doc), up to 3 * per_low rows of one
    # doc share the low digit, so the rows a boundary group spans stay below the 64 the pass looks at
    c = _dense_runs(n_docs=900, n_terms=2048, low_bits=6, per_low=per_low, n_articles=n_articles, seed=per_low)
    assert len(c["term_id"]) > 6 * TILE
    orc = oracle_index(c, word_count=None)
    ix = gpu_index_factory()
    gpu_build(ix, c, word_count=None)
    assert ix.info().fused_postings == 1
    assert_build_matches(ix, orc)
    orc.close()


def test_boundary_groups_too_long_fall_back(gpu_index_factory):
    # 30 terms x 3 fields of one doc in one low-digit class: 90 rows, more than the pass looks at beside a tile
    c = _dense_runs(n_docs=400, n_terms=2048, low_bits=6, per_low=30, seed=9)
    orc = oracle_index(c, word_count=None)
    ix = gpu_index_factory()
    gpu_build(ix, c, word_count=None)
    assert ix.info().fused_postings == 0
    assert_build_matches(ix, orc)
    orc.close()


def test_rows_not_done_and_news_at_the_boundaries(gpu_index_factory):
    c = _dense_runs(n_docs=700, n_terms=4096, low_bits=6, per_low=10, n_articles=400, seed=4)
    rng = np.random.default_rng(11)
    flags = np.where(rng.random(700) < 0.7, se.DOC_DONE, 0).astype(np.uint8)
    orc = oracle_index(c, word_count=None, doc_flags=flags)
    ix = gpu_index_factory()
    gpu_build(ix, c, word_count=None, doc_flags=flags)
    assert ix.info().fused_postings == 1
    assert_build_matches(ix, orc)
    orc.close()


def test_same_index_as_the_separate_posting_pass(gpu_index_factory):
    """The two build paths on one input: every array the searches read, and the searches themselves."""
    c = small_corpus(n_docs=20_000, n_terms=60_000, n_articles=1500)
    rng = np.random.default_rng(2)
    flags = np.where(rng.random(20_000) < 0.9, se.DOC_DONE, 0).astype(np.uint8)
    a = gpu_index_factory()
    gpu_build(a, c, doc_flags=flags)
    assert a.info().fused_postings == 1
    os.environ["SEBM25_FUSED_POSTINGS"] = "0"
    try:
        b = gpu_index_factory()
    finally:
        del os.environ["SEBM25_FUSED_POSTINGS"]
    gpu_build(b, c, doc_flags=flags)
    assert b.info().fused_postings == 0
    sa, sb = a.stats(), b.stats()
    for key in ("df", "term_rows"):
        np.testing.assert_array_equal(sa[key], sb[key])
    np.testing.assert_array_equal(sa["idf"].view(np.uint64), sb["idf"].view(np.uint64))
    for x, y in zip(a.postings(), b.postings()):
        np.testing.assert_array_equal(x, y)
    ia, ib = a.info(), b.info()
    assert (ia.n_postings, ia.n_web_postings, ia.n_rows) == (ib.n_postings, ib.n_web_postings, ib.n_rows)
    qs = se.gen_queries(7, 60_000, 300, 3, 1)
    for src in (se.SRC_WEB, se.SRC_NEWS):
        for k in (10, 210):
            ra, rb = a.search(qs, k, source=src), b.search(qs, k, source=src)
            np.testing.assert_array_equal(ra["count"], rb["count"])
            np.testing.assert_array_equal(ra["unit"], rb["unit"])
            np.testing.assert_array_equal(ra["score"].view(np.uint64), rb["score"].view(np.uint64))
            np.testing.assert_array_equal(ra["matched"], rb["matched"])


def test_update_and_compaction_after_a_fused_build(gpu_index_factory):
    """The sorted rows an update reads are made when it first asks for them."""
    c = small_corpus(n_docs=4000, n_terms=30_000, n_articles=300)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    assert ix.info().fused_postings == 1
    # docs 100..139 are crawled again with new text
    spec = O.make_spec(seed=991, n_terms=30_000, n_docs=40)
    u = O.gen_corpus(spec)
    keep = ~((c["unit_id"] >= 100) & (c["unit_id"] < 140) & ((c["meta"] & 3) == se.SRC_WEB))
    new_unit = u["unit_id"] + np.uint32(100)
    wc = c["word_count"].copy()
    wc[100:140] = u["word_count"]
    web_end = int(np.searchsorted(((c["meta"] & 3) == se.SRC_NEWS).astype(np.int8), 1))
    pos = int(np.searchsorted(c["unit_id"][:web_end], 140))
    full = {}
    for key, new in (("term_id", u["term_id"]), ("unit_id", new_unit), ("tf", u["tf"]), ("meta", u["meta"])):
        old = c[key][keep]
        cut = pos - int((~keep[:pos]).sum())
        full[key] = np.concatenate([old[:cut], new, old[cut:]])
    c2 = dict(c, word_count=wc, **full)
    orc = oracle_index(c2)
    ix.update(u["term_id"], new_unit, u["tf"], u["meta"], 30_000, 4000, 300, wc, None,
              changed_docs=np.arange(100, 140, dtype=np.uint32), unit_major=True)
    assert_build_matches(ix, orc)
    ix.compact()
    assert ix.info().n_segments == 1
    assert_build_matches(ix, orc)
    orc.close()
