"""Scratch: time the 10k x 3-term query batch on a synthetic index with the library named by SEBM25_LIB;
prints per-batch ms and (stats builds) the step counters."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import search_engine_amd as se
from search_engine_amd import native

docs = int(os.environ.get("DOCS", 1_000_000)); V = 500_000; Q = int(os.environ.get("Q", 10_000))
ks = [int(x) for x in os.environ.get("KS", "10,220").split(",")]
dev = torch.device("cuda", 0)
ix = se.Index(0)
spec = se.make_spec(n_terms=V, n_docs=docs)
n = ix.gen_corpus_rows(spec)
term = torch.empty(n, dtype=torch.int32, device=dev); unit = torch.empty_like(term); tf = torch.empty_like(term)
meta = torch.empty(n, dtype=torch.uint8, device=dev); wc = torch.empty(docs, dtype=torch.int32, device=dev)
ix.gen_corpus(spec, term, unit, tf, meta, wc)
ix.build(term, unit, tf, meta, V, docs, 0, wc)
if os.environ.get("RANK") == "1":
    ix.set_url_rank(se.SRC_WEB, torch.randperm(docs, device=dev).to(torch.int32))
qs = se.gen_queries(20260228, V, Q, 3, 100)
L = native.lib()
res = {}
for k in ks:
    out = {kk: torch.empty((Q, k), dtype=dt, device=dev) for kk, dt in (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
    out["count"] = torch.empty(Q, dtype=torch.int32, device=dev)
    ix.search(qs, k, out=out)
    if hasattr(L, "sebm25_exp_stats"):
        buf = (C.c_ulonglong * 8)(); L.sebm25_exp_stats(buf, 1)
    ix.enable_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ix.search(qs, k, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
    ph = dict(ix.timings()); ix.enable_timing(False)
    r = {"ms_per_batch": round(dt, 3), "score_ms": round(ph.get("score_topk", 0), 3), "merge_ms": round(ph.get("merge_slices", 0), 3)}
    if hasattr(L, "sebm25_exp_stats"):
        buf = (C.c_ulonglong * 8)(); L.sebm25_exp_stats(buf, 1)
        s = [int(x) / 5 for x in buf]
        r["stats"] = dict(steps=s[0], insert_passes=s[1], inserted=s[2], harvest_rounds=s[3], offers=s[4], items=s[5])
    res[k] = r
print(json.dumps({"lib": os.path.basename(native.LIB_PATH), "docs": docs, "rank": os.environ.get("RANK"), "res": res}))
