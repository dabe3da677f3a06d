"""Scratch: ms per 10k-query batch and algorithmic GB/s by number of terms per query."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import search_engine_amd as se
docs = int(os.environ.get("DOCS", 10_000_000)); V = 500_000; Q = 10_000
dev = torch.device("cuda", 0)
ix = se.Index(0)
spec = se.make_spec(n_terms=V, n_docs=docs)
n = ix.gen_corpus_rows(spec)
term = torch.empty(n, dtype=torch.int32, device=dev); unit = torch.empty_like(term); tf = torch.empty_like(term)
meta = torch.empty(n, dtype=torch.uint8, device=dev); wc = torch.empty(docs, dtype=torch.int32, device=dev)
ix.gen_corpus(spec, term, unit, tf, meta, wc)
ix.build(term, unit, tf, meta, V, docs, 0, wc)
st = ix.stats()
df = np.asarray(st["df"], dtype=np.int64)
for nt in [int(x) for x in os.environ.get("NTS", "1,2,3,4").split(",")]:
    qs = se.gen_queries(20260228, V, Q, nt, 100)
    post = sum(int(df[list(set(q))].sum()) for q in qs)
    for k in (10, 210):
        out = {kk: torch.empty((Q, k), dtype=dt, device=dev) for kk, dt in (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
        out["count"] = torch.empty(Q, dtype=torch.int32, device=dev)
        ix.search(qs, k, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): ix.search(qs, k, out=out)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"T{nt} k{k}: {ms:.2f} ms/batch, postings {post:.3e}, {8 * post / ms / 1e9:.2f} TB/s algorithmic", flush=True)
