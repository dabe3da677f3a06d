"""Golden fixture tests/golden/bm25_small.npz (made by tests/golden/make_golden.py from the CPU
oracle): the oracle must keep reproducing it (CPU), and the GPU path must match it (gpu)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bm25_small.npz"))


def _oracle():
    return O.OracleIndex(G["term_id"], G["unit_id"], G["tf"], G["meta"], int(G["n_terms"]), int(G["n_docs"]),
                         int(G["n_articles"]), G["word_count"], G["doc_flags"])


def test_oracle_reproduces_golden():
    orc = _oracle()
    for key in ("df", "term_rows", "doc_len", "row_ptr", "news_ptr", "post_unit"):
        np.testing.assert_array_equal(getattr(orc, key), G[key], err_msg=key)
    np.testing.assert_array_equal(orc.post_weight, G["post_weight"])
    np.testing.assert_allclose(orc.idf, G["idf"], rtol=4e-16, atol=0)    # libm log1p may differ by an ulp across hosts
    assert orc.n_docs_stat == int(G["n_docs_stat"]) and orc.avgdl == float(G["avgdl"])
    t = orc.totals
    assert [t.n_done, t.n_wc, t.sum_wc, t.n_postings] == G["totals"].tolist()
    orc.set_idf(G["idf"])
    k = int(G["k"])
    for name, source, level in (("web_rows", 1, "rows"), ("web_csr", 1, "csr"), ("news_rows", 2, "rows"), ("news_csr", 2, "csr")):
        for i, q in enumerate(G["queries"]):
            u, s, m = orc.search(q, k, source=source, level=level)
            n = int(G[f"{name}_count"][i])
            assert len(u) == n
            np.testing.assert_array_equal(u, G[f"{name}_unit"][i, :n])
            np.testing.assert_array_equal(s, G[f"{name}_score"][i, :n])
            np.testing.assert_array_equal(m, G[f"{name}_matched"][i, :n])


def test_golden_is_the_reference_formula_within_tolerance():
    # merged integer postings vs the row-by-row SQL formula: far inside the 1e-6 relative bar
    # (this fixture has docs that are not 'done', so df can exceed N and idf can be negative:
    # scores then mix signs, which is where a float posting weight would lose the bar)
    for src in ("web", "news"):
        a, b = G[f"{src}_rows_score"], G[f"{src}_csr_score"]
        assert np.all(np.abs(a - b) <= 1e-12 * np.abs(a))
        # the two rankings differ only where neighbouring scores tie inside that rounding
        for q, i in np.argwhere(G[f"{src}_rows_unit"] != G[f"{src}_csr_unit"]):
            j = i + 1 if i + 1 < a.shape[1] and G[f"{src}_rows_unit"][q, i] == G[f"{src}_csr_unit"][q, i + 1] else i - 1
            assert abs(a[q, i] - a[q, j]) <= 1e-12 * abs(a[q, i])


@pytest.mark.gpu
def test_gpu_matches_golden(gpu_index_factory):
    import search_engine_amd as se

    ix = gpu_index_factory()
    ix.build(G["term_id"], G["unit_id"], G["tf"], G["meta"], int(G["n_terms"]), int(G["n_docs"]), int(G["n_articles"]),
             G["word_count"], G["doc_flags"])
    st = ix.stats()
    np.testing.assert_array_equal(st["df"], G["df"])
    np.testing.assert_array_equal(st["term_rows"], G["term_rows"])
    np.testing.assert_allclose(st["idf"], G["idf"], rtol=1e-14, atol=0)
    assert st["n_docs"] == int(G["n_docs_stat"]) and st["avgdl"] == float(G["avgdl"])
    np.testing.assert_array_equal(ix.doc_len(), G["doc_len"])
    rp, nps, pu, pw = ix.postings()
    np.testing.assert_array_equal(rp, G["row_ptr"])
    np.testing.assert_array_equal(nps, G["news_ptr"])
    np.testing.assert_array_equal(pu, G["post_unit"])
    np.testing.assert_array_equal(pw, G["post_weight"])
    ix.set_idf(G["idf"])
    k = int(G["k"])
    for src, source in (("web", se.SRC_WEB), ("news", se.SRC_NEWS)):
        out = ix.search(G["queries"], k, source=source)
        np.testing.assert_array_equal(out["count"], G[f"{src}_csr_count"])
        np.testing.assert_array_equal(out["unit"], G[f"{src}_csr_unit"])
        np.testing.assert_array_equal(out["score"], G[f"{src}_csr_score"])
        np.testing.assert_array_equal(out["matched"], G[f"{src}_csr_matched"])
        # and within 1e-6 relative of the row-by-row reference formula
        ref = G[f"{src}_rows_score"]
        assert np.all(np.abs(out["score"] - ref) <= 1e-6 * np.abs(ref))
