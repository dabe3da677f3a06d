#!/usr/bin/env python3
"""Writes tests/golden/bm25_small.npz: a small seeded corpus (token rows + document columns),
the statistics and CSR the CPU oracle derives from it, and top-k results for a query set.

The vectors come from this repository's own CPU restatement (oracle/bm25_oracle.c) — the
reference holds no fixtures for this path — and freeze its behaviour so that later changes to
the oracle or the kernels are caught.  Run from the repository root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    spec = O.make_spec(seed=20260227, n_terms=1500, n_docs=120, n_articles=24, body_min=30, body_max=80)
    c = O.gen_corpus(spec)
    rng = np.random.default_rng(20260227)
    flags = np.ones(120, np.uint8)
    flags[rng.choice(120, 9, replace=False)] = 0          # not 'done'
    flags[rng.choice(120, 5, replace=False)] |= 2         # word_count IS NULL
    orc = O.OracleIndex(c["term_id"], c["unit_id"], c["tf"], c["meta"], c["n_terms"], c["n_docs"], c["n_articles"],
                        c["word_count"], flags)
    qs = O.gen_queries(20260228, 1500, O.zipf_cdf(1500), 1, 48, 3)
    k = 12
    res = {}
    for name, source, level in (("web_rows", 1, "rows"), ("web_csr", 1, "csr"), ("news_rows", 2, "rows"), ("news_csr", 2, "csr")):
        U = np.full((len(qs), k), 0xFFFFFFFF, np.uint32)
        S = np.zeros((len(qs), k))
        M = np.zeros((len(qs), k), np.uint32)
        N = np.zeros(len(qs), np.uint32)
        for i, q in enumerate(qs):
            u, s, m = orc.search(q, k, source=source, level=level)
            N[i] = len(u)
            U[i, :len(u)], S[i, :len(u)], M[i, :len(u)] = u, s, m
        res.update({f"{name}_unit": U, f"{name}_score": S, f"{name}_matched": M, f"{name}_count": N})
    t = orc.totals
    out = os.path.join(ROOT, "tests", "golden", "bm25_small.npz")
    np.savez_compressed(
        out, term_id=c["term_id"], unit_id=c["unit_id"], tf=c["tf"], meta=c["meta"], word_count=c["word_count"],
        doc_flags=flags, n_terms=1500, n_docs=120, n_articles=24, df=orc.df, idf=orc.idf, term_rows=orc.term_rows,
        doc_len=orc.doc_len, row_ptr=orc.row_ptr, news_ptr=orc.news_ptr, post_unit=orc.post_unit,
        post_weight=orc.post_weight, totals=np.array([t.n_done, t.n_wc, t.sum_wc, t.n_postings], np.uint64),
        n_docs_stat=orc.n_docs_stat, avgdl=orc.avgdl, queries=qs, k=k, **res)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
