"""A small numpy generator of token rows with the reference's shapes for the measurement tools (Zipf term ranks,
a few title / description rows and a body per doc).  It is not the benchmark corpus and no checker."""
import numpy as np


def zipf_cdf(n_terms: int) -> np.ndarray:
    w = 1.0 / np.arange(1, n_terms + 1, dtype=np.float64)
    return np.cumsum(w / w.sum())


def token_rows(n_docs: int, n_terms: int, seed: int = 1, body_min: int = 100, body_max: int = 500, title_len: int = 8, desc_len: int = 24):
    """-> dict(term_id, unit_id, tf, meta (source 1 | field << 2), word_count): rows grouped by doc, one row per
    (doc, field, term) with its frequency."""
    rng = np.random.default_rng(seed)
    cdf = zipf_cdf(n_terms)
    body = rng.integers(body_min, body_max + 1, n_docs)
    parts = []
    for field, lens in ((1, np.full(n_docs, title_len)), (2, np.full(n_docs, desc_len)), (4, body)):
        doc = np.repeat(np.arange(n_docs, dtype=np.int64), lens)
        term = np.minimum(np.searchsorted(cdf, rng.random(len(doc))), n_terms - 1).astype(np.int64)
        key, tf = np.unique(doc * n_terms + term, return_counts=True)       # draws of one term in one field of one doc
        parts.append((key // n_terms, key % n_terms, tf, np.full(len(key), 1 | (field << 2))))
    doc = np.concatenate([p[0] for p in parts]); term = np.concatenate([p[1] for p in parts])
    tf = np.concatenate([p[2] for p in parts]); meta = np.concatenate([p[3] for p in parts])
    order = np.argsort(doc, kind="stable")
    return dict(term_id=term[order].astype(np.uint32), unit_id=doc[order].astype(np.uint32), tf=tf[order].astype(np.uint32),
                meta=meta[order].astype(np.uint8), word_count=body.astype(np.uint32))


def queries(n_terms: int, n_queries: int, terms_per_query: int, min_rank: int, seed: int = 2) -> np.ndarray:
    rng = np.random.default_rng(seed)
    cdf = zipf_cdf(n_terms)
    lo = cdf[min_rank - 1] if min_rank > 0 else 0.0
    out = np.empty((n_queries, terms_per_query), np.uint32)
    for q in range(n_queries):
        seen: list = []
        while len(seen) < terms_per_query:
            t = int(min(np.searchsorted(cdf, lo + rng.random() * (1.0 - lo)), n_terms - 1))
            if t not in seen:
                seen.append(t)
        out[q] = seen
    return out
