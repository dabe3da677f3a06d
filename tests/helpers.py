"""Shared checks for the parity tests."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O

REL_TOL = 1e-6  # scores: within 1e-6 relative of the reference formula evaluated in double


def small_corpus(n_docs=300, n_terms=2000, n_articles=0, seed=20260227, **kw):
    spec = O.make_spec(seed=seed, n_terms=n_terms, n_docs=n_docs, n_articles=n_articles, **kw)
    return O.gen_corpus(spec)


def oracle_index(c, word_count="corpus", doc_flags=None):
    wc = c["word_count"] if word_count == "corpus" else word_count
    return O.OracleIndex(c["term_id"], c["unit_id"], c["tf"], c["meta"], c["n_terms"], c["n_docs"], c["n_articles"],
                         wc, doc_flags)


def gpu_build(ix, c, word_count="corpus", doc_flags=None, unit_major=True, local=False):
    wc = c["word_count"] if word_count == "corpus" else word_count
    fn = ix.build_local if local else ix.build
    fn(c["term_id"], c["unit_id"], c["tf"], c["meta"], c["n_terms"], c["n_docs"], c["n_articles"], wc, doc_flags,
       unit_major=unit_major)


def assert_build_matches(ix, orc, idf_rtol=1e-14):
    """df, doc lengths, posting counts, CSR: bit-exact; idf: a few ulp (device log1p vs libm)."""
    st = ix.stats()
    np.testing.assert_array_equal(st["df"], orc.df)
    np.testing.assert_array_equal(st["term_rows"], orc.term_rows)
    assert st["n_docs"] == orc.n_docs_stat
    assert st["avgdl"] == orc.avgdl
    np.testing.assert_allclose(st["idf"], orc.idf, rtol=idf_rtol, atol=0)
    np.testing.assert_array_equal(ix.doc_len(), orc.doc_len)
    rp, nps, pu, pw = ix.postings()
    np.testing.assert_array_equal(rp, orc.row_ptr)
    np.testing.assert_array_equal(nps, orc.news_ptr)
    np.testing.assert_array_equal(pu, orc.post_unit)
    np.testing.assert_array_equal(pw, orc.post_weight)      # exact integer weights (tenths)


def assert_topk_valid(got_unit, got_score, got_matched, ref_unit, ref_score, ref_matched, k, tie_of=lambda u: u,
                      rtol=REL_TOL):
    """`got` (from the GPU) is a correct top-k given the reference's full-precision ranking
    `ref` (top-k' with k' >= k, ideally all matches), identical modulo ties inside rtol."""
    ref = {int(u): (float(s), int(m)) for u, s, m in zip(ref_unit, ref_score, ref_matched)}
    n = len(got_unit)
    assert n == min(k, len(ref_unit)) or (len(ref_unit) >= k and n == k), (n, k, len(ref_unit))
    for i in range(n):
        u = int(got_unit[i])
        if u in ref:
            s, m = ref[u]
            assert abs(got_score[i] - s) <= rtol * abs(s) + 1e-300, (u, got_score[i], s)
            assert int(got_matched[i]) == m
        else:
            # not in the reference top-k': only legitimate when it ties with the cut-off
            assert len(ref_unit) >= k
            s_cut = float(ref_score[k - 1])
            assert abs(got_score[i] - s_cut) <= rtol * abs(s_cut), (u, got_score[i], s_cut)
        if i:
            a, b = float(got_score[i - 1]), float(got_score[i])
            assert a > b or (a == b and tie_of(int(got_unit[i - 1])) < tie_of(u)), (i, a, b)
    # same ordering as the reference wherever scores are separated by more than the tolerance
    for i in range(min(n, len(ref_unit))):
        if int(got_unit[i]) != int(ref_unit[i]):
            s = float(ref_score[i])
            assert abs(got_score[i] - s) <= rtol * abs(s), (i, got_unit[i], ref_unit[i], got_score[i], s)
