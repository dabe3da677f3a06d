"""Scratch: cost of searching an index that holds update segments (10k x 3-term batch, 1M-doc base)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import search_engine_amd as se
docs = 1_000_000; V = 500_000; UPD = 1000
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
qs = se.gen_queries(20260228, V, 10_000, 3, 100)
def timeit(k):
    out = {kk: torch.empty((10_000, k), dtype=dt, device=dev) for kk, dt in (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
    out["count"] = torch.empty(10_000, dtype=torch.int32, device=dev)
    ix.search(qs, k, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ix.search(qs, k, out=out)
    torch.cuda.synchronize(); return round((time.perf_counter() - t0) / 5 * 1e3, 2)
res = {}
for nseg in range(0, 9):
    if nseg:
        a = (nseg * 2 + 1) * docs // 19
        u = gen(ix, 900 + nseg, UPD, a); u["unit"] += a; wc[a:a + UPD] = u["wc"]
        ix.update(u["term"], u["unit"], u["tf"], u["meta"], V, docs, 0, wc, flags, changed_docs=np.arange(a, a + UPD, dtype=np.uint32), unit_major=True)
    res[f"segments_{nseg + 1}"] = {"k10_ms": timeit(10), "k210_ms": timeit(210)}
print(json.dumps(res))
