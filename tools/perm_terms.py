"""Build time when term ids are uncorrelated with term frequency (alphabetical ids in a real
corpus): the synthetic corpus with its term ids pushed through a random permutation."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import search_engine_amd as se
DOCS, VOCAB = 1_000_000, 500_000
ix = se.Index(0)
spec = se.make_spec(n_terms=VOCAB, n_docs=DOCS)
n = ix.gen_corpus_rows(spec)
dev = torch.device("cuda", 0)
term = torch.empty(n, dtype=torch.int32, device=dev); unit = torch.empty_like(term); tf = torch.empty_like(term)
meta = torch.empty(n, dtype=torch.uint8, device=dev); wc = torch.empty(DOCS, dtype=torch.int32, device=dev)
ix.gen_corpus(spec, term, unit, tf, meta, wc)
g = torch.Generator(device="cpu"); g.manual_seed(1)
perm = torch.randperm(VOCAB, generator=g).to(dev).int()
term_p = perm[term.long()].contiguous()
for name, t in (("rank ids", term), ("permuted ids", term_p)):
    for _ in range(3): ix.build(t, unit, tf, meta, VOCAB, DOCS, 0, wc)
    ix.enable_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ix.build(t, unit, tf, meta, VOCAB, DOCS, 0, wc)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(name, "%.3f ms" % ms, {k: round(v, 3) for k, v in ix.timings()})
    ix.enable_timing(False)
    qs = se.gen_queries(20260228, VOCAB, 10000, 3, 100)
    if name == "permuted ids": qs = perm.cpu().numpy().astype(np.uint32)[qs]
    ix.search(qs, 10); t0 = time.perf_counter()
    for _ in range(3): ix.search(qs, 10)
    print("   queries: %.3f ms per 10k" % ((time.perf_counter() - t0) / 3 * 1e3))
