import numpy as np, sys, os
sys.path.insert(0, "/root/repo")
import search_engine_amd as se
G = np.load("tests/golden/bm25_small.npz")
ix = se.Index(0)
ix.build(G["term_id"], G["unit_id"], G["tf"], G["meta"], int(G["n_terms"]), int(G["n_docs"]), int(G["n_articles"]), G["word_count"], G["doc_flags"])
st = ix.stats()
bad = np.nonzero(st["term_rows"] != G["term_rows"])[0]
print("bad terms", bad, st["term_rows"][bad], G["term_rows"][bad])
lo = max(bad.min() - 3, 0); hi = min(bad.max() + 4, len(G["term_rows"]))

print("R", len(G["term_id"]), "sum gpu", st["term_rows"].sum(), "sum exp", G["term_rows"].sum())
