"""Scratch: update + compaction cycles on the 1M-doc index against a full build of the final table (device arrays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import search_engine_amd as se
docs = int(os.environ.get("DOCS", 1_000_000)); V = 500_000; UPD = int(os.environ.get("UPD", 20_000))
dev = torch.device("cuda", 0)
def gen(ix, seed, n_docs, doc_begin):
    spec = se.make_spec(seed=seed, n_terms=V, n_docs=n_docs, doc_begin=doc_begin)
    n = ix.gen_corpus_rows(spec)
    c = dict(term=torch.empty(n, dtype=torch.int32, device=dev), unit=torch.empty(n, dtype=torch.int32, device=dev),
             tf=torch.empty(n, dtype=torch.int32, device=dev), meta=torch.empty(n, dtype=torch.uint8, device=dev),
             wc=torch.empty(n_docs, dtype=torch.int32, device=dev))
    ix.gen_corpus(spec, c["term"], c["unit"], c["tf"], c["meta"], c["wc"])
    return c
ix = se.Index(0)
c = gen(ix, 20260227, docs, 0)
wc = c["wc"].clone(); flags = torch.full((docs,), se.DOC_DONE, dtype=torch.uint8, device=dev)
ix.build(c["term"], c["unit"], c["tf"], c["meta"], V, docs, 0, wc, flags)
keep = torch.ones(c["term"].numel(), dtype=torch.bool, device=dev)
new_parts = []
for cyc in range(2):
    for i in range(3):
        a = ((cyc * 3 + i) * 2 + 1) * docs // 13
        u = gen(ix, 900 + cyc * 3 + i, UPD, a)
        u["unit"] += a
        wc[a:a + UPD] = u["wc"]
        changed = np.arange(a, a + UPD, dtype=np.uint32)
        ix.update(u["term"], u["unit"], u["tf"], u["meta"], V, docs, 0, wc, flags, changed_docs=changed, unit_major=True)
        keep &= ~((c["unit"] >= a) & (c["unit"] < a + UPD))
        new_parts = [p for p in new_parts if not (p["a"] == a)] + [dict(a=a, **u)]
    t0 = time.perf_counter(); ix.compact(); torch.cuda.synchronize(); print("compact ms", (time.perf_counter() - t0) * 1e3, ix.info().n_rows, flush=True)
ref = se.Index(0)
cat = lambda k: torch.cat([c[k][keep]] + [p[k] for p in new_parts])
ref.build(cat("term"), cat("unit"), cat("tf"), cat("meta"), V, docs, 0, wc, flags, unit_major=False)
sa, sb = ix.stats(), ref.stats()
for key in ("df", "term_rows", "idf"):
    assert np.array_equal(sa[key], sb[key]), key
assert sa["n_docs"] == sb["n_docs"] and sa["avgdl"] == sb["avgdl"]
ia, ib = ix.info(), ref.info()
assert (ia.n_rows, ia.n_postings, ia.n_web_postings) == (ib.n_rows, ib.n_postings, ib.n_web_postings), (ia.n_rows, ib.n_rows)
qs = se.gen_queries(5, V, 2000, 3, 100)
for k in (10, 210):
    oa, ob = ix.search(qs, k), ref.search(qs, k)
    for key in ("count", "unit", "score", "matched"):
        assert np.array_equal(oa[key], ob[key]), (key, k)
print("large compaction check ok:", ia.n_rows, "rows")
