"""Scratch: latency of small batches (1, 8, 64 queries) on a synthetic index."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import search_engine_amd as se
docs = int(os.environ.get("DOCS", 10_000_000)); V = 500_000
dev = torch.device("cuda", 0)
ix = se.Index(0)
spec = se.make_spec(n_terms=V, n_docs=docs)
n = ix.gen_corpus_rows(spec)
term = torch.empty(n, dtype=torch.int32, device=dev); unit = torch.empty_like(term); tf = torch.empty_like(term)
meta = torch.empty(n, dtype=torch.uint8, device=dev); wc = torch.empty(docs, dtype=torch.int32, device=dev)
ix.gen_corpus(spec, term, unit, tf, meta, wc)
ix.build(term, unit, tf, meta, V, docs, 0, wc)
qs = se.gen_queries(20260228, V, 4096, 3, 100)
res = {}
for nq in (1, 8, 64, 512):
    for k in (10, 210):
        ts = []
        for i in range(0, 2048, max(nq, 16)):
            q = qs[i:i + nq]
            t0 = time.perf_counter(); ix.search(q, k); ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.array(ts[2:]); res[f"q{nq}_k{k}"] = {"median_ms": round(float(np.median(ts)), 3), "p99_ms": round(float(np.percentile(ts, 99)), 3)}
print(json.dumps(res))
