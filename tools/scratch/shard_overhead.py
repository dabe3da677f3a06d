"""Scratch: what the split build (build_local + local_stats + finalize) costs beside the fused sebm25_build."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import search_engine_amd as se
from search_engine_amd.sharded import ShardedIndex
docs = int(os.environ.get("DOCS", 1_250_000)); V = 500_000
dev = torch.device("cuda", 0)
ix = se.Index(0)
ix.set_stream(torch.cuda.current_stream().cuda_stream)
spec = se.make_spec(n_terms=V, n_docs=docs)
n = ix.gen_corpus_rows(spec)
term = torch.empty(n, dtype=torch.int32, device=dev); unit = torch.empty_like(term); tf = torch.empty_like(term)
meta = torch.empty(n, dtype=torch.uint8, device=dev); wc = torch.empty(docs, dtype=torch.int32, device=dev)
fl = torch.full((docs,), se.DOC_DONE, dtype=torch.uint8, device=dev)
ix.gen_corpus(spec, term, unit, tf, meta, wc)
sh = ShardedIndex(ix, dev)
def fused(): ix.build(term, unit, tf, meta, V, docs, 0, wc, fl)
def split(): sh.build(term, unit, tf, meta, V, docs, 0, wc, fl)
def split_limbs():
    # what world > 1 adds on the host besides the collective itself
    sh.build(term, unit, tf, meta, V, docs, 0, wc, fl)
    t = sh.totals
    limbs = [(int(v) >> (16 * j)) & 0xFFFF for v in (t.n_done, t.n_wc, t.sum_wc, t.n_postings) for j in range(4)]
    sh._buf[V:].copy_(torch.tensor(limbs, dtype=torch.int32))
    sh._buf[V:].tolist()
for name, f in (("fused", fused), ("split", split), ("split+limbs", split_limbs)):
    for _ in range(5): f()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "median ms", round(statistics.median(ts), 3), flush=True)
