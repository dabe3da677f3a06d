"""Build time when the caller hands host (numpy) arrays: staging over PCIe included."""
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
h = [t.cpu().numpy().view(np.uint32) if t.dtype != torch.uint8 else t.cpu().numpy() for t in (term, unit, tf, meta, wc)]
del term, unit, tf, meta
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.build(h[0], h[1], h[2], h[3], VOCAB, DOCS, 0, h[4])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    gb = (n * 13 + DOCS * 4) / 1e9
    print("host arrays: %.1f ms per build, %.1f GB/s over PCIe incl. device work" % (ms, gb / ms * 1e3))
