"""Scratch: 10k x 3-term batch with explicit url ranks: a full rank array against keys that need no array for the
units of the last rebuild (the last 1 % of the units count as appended)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
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
qs = se.gen_queries(20260228, V, Q, 3, 100)
nb, shift = docs - docs // 100, 8
rng = np.random.default_rng(1)
gaps = np.sort(rng.integers(0, nb + 1, docs - nb)); keys = np.zeros(docs - nb, np.uint32); seen = {}
for i, g in enumerate(gaps.tolist()):
    seen[g] = seen.get(g, 0) + 1; keys[i] = (g << shift) + seen[g]
full = np.concatenate([(np.arange(nb, dtype=np.uint64) + 1 << shift).astype(np.uint32), keys])
ranks = np.empty(docs, np.uint32); ranks[np.argsort(full, kind="stable")] = np.arange(docs, dtype=np.uint32)
res = {}
for mode in ("none", "full", "appended"):
    if mode == "none": ix.set_url_rank(se.SRC_WEB, None)
    elif mode == "full": ix.set_url_rank(se.SRC_WEB, torch.from_numpy(ranks.view(np.int32)).to(dev))
    else: ix.set_url_rank_appended(se.SRC_WEB, nb, shift, torch.from_numpy(keys.view(np.int32)).to(dev))
    for k in (10, 210):
        out = {kk: torch.empty((Q, k), dtype=dt, device=dev) for kk, dt in (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
        out["count"] = torch.empty(Q, dtype=torch.int32, device=dev)
        ix.search(qs, k, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): ix.search(qs, k, out=out)
        torch.cuda.synchronize(); res[f"{mode}_k{k}"] = round((time.perf_counter() - t0) / 4 * 1e3, 2)
print(json.dumps(res))
