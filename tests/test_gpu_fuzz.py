"""Seeded random sweep over what the fixed cases do not pin down one by one: corpus shape (dense and sparse lists,
few and many docs, with and without articles), list count 1..12, k on and beside every boundary between the kinds
of candidate storage, both sources, with and without an explicit rank array, zero / negative idf values — every
result bit for bit against the oracle's posting-level ranking."""
import os

import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import assert_build_matches, gpu_build, oracle_index, small_corpus

pytestmark = pytest.mark.gpu

K_EDGES = [1, 2, 9, 47, 48, 49, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048]


@pytest.mark.parametrize("case", range(int(os.environ.get("SEBM25_FUZZ_CASES", "12"))))   # more for a longer sweep
def test_random_configurations_against_the_oracle(gpu_index_factory, case):
    rng = np.random.default_rng(9000 + case)
    n_docs = int(rng.choice([200, 1500, 6000, 20000]))
    n_terms = int(rng.choice([60, 500, 4000]))
    n_articles = int(rng.choice([0, 50, 900]))
    body = int(rng.choice([5, 40, 120]))
    c = small_corpus(n_docs=n_docs, n_terms=n_terms, n_articles=n_articles, seed=31 + case, body_min=body, body_max=2 * body,
                     title_len=int(rng.integers(1, 6)), desc_len=int(rng.integers(1, 10)))
    flags = np.ones(n_docs, np.uint8)
    flags[rng.random(n_docs) < 0.1] = 0                       # some docs are not 'done'
    orc = oracle_index(c, doc_flags=flags)
    ix = gpu_index_factory()
    gpu_build(ix, c, doc_flags=flags)
    assert_build_matches(ix, orc)                              # df, lengths, CSR, weights: bit for bit; idf to 1e-14
    idf = orc.idf.copy()
    if case % 3 == 2:                                          # an idf table with zeros and negative values
        idf[rng.random(n_terms) < 0.2] *= -1.0
        idf[rng.random(n_terms) < 0.1] = 0.0
    ix.set_idf(idf)
    orc.set_idf(idf)
    ranks = {se.SRC_WEB: None, se.SRC_NEWS: None}
    if case % 2:
        ranks[se.SRC_WEB] = rng.permutation(n_docs).astype(np.uint32)
        ix.set_url_rank(se.SRC_WEB, ranks[se.SRC_WEB])
        if n_articles:
            ranks[se.SRC_NEWS] = rng.permutation(n_articles).astype(np.uint32)
            ix.set_url_rank(se.SRC_NEWS, ranks[se.SRC_NEWS])
    cdf = O.zipf_cdf(n_terms)
    for tpq in (1, 2, 3, 4, 5, 7, 8, 9, 12):
        if tpq > n_terms // 4:
            continue
        qs = O.gen_queries(100 * case + tpq, n_terms, cdf, 1, 8, tpq)
        qs = [q for q in qs] + [np.array([0], np.uint32), np.array([n_terms - 1, 0], np.uint32)]
        for k in rng.choice(K_EDGES, 5, replace=False).tolist():
            for source in (se.SRC_WEB, se.SRC_NEWS):
                if source == se.SRC_NEWS and not n_articles:
                    continue
                out = ix.search(qs, k, source=source)
                for qi, q in enumerate(qs):
                    ru, rs, rm = orc.search(q, k, source=source, url_rank=ranks[source], level="csr")
                    n = int(out["count"][qi])
                    assert n == len(ru), (case, tpq, k, source, qi)
                    np.testing.assert_array_equal(out["unit"][qi, :n], ru, err_msg=f"case {case} tpq {tpq} k {k} source {source} query {qi}")
                    np.testing.assert_array_equal(out["score"][qi, :n].view(np.uint64), rs.view(np.uint64))
                    np.testing.assert_array_equal(out["matched"][qi, :n], rm)
