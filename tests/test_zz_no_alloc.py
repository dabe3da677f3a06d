"""The entry point that allocates nothing (sebm25_search_noalloc) against sebm25_search: same
results bit for bit, on two calls in a row (the second with a smaller input, so anything the first
call left in the shared fixed device array would show), through the C ABI."""
import numpy as np
import pytest

import search_engine_amd as se
from oracle import oracle as O

from helpers import gpu_build, small_corpus

pytestmark = pytest.mark.gpu


def _dev_out(nq, k):
    import torch

    dev = torch.device("cuda", 0)
    out = {key: torch.full((nq, k), -7, dtype=dt, device=dev)
           for key, dt in (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
    out["count"] = torch.full((nq,), -7, dtype=torch.int32, device=dev)
    return out


def _noalloc_views(out, nq, k):
    """Views into preallocated outputs, contiguous and large enough for nq x k results (no allocation)."""
    v = {key: t.view(-1)[:nq * k].view(nq, k) for key, t in out.items() if key != "count"}
    v["count"] = out["count"][:nq]
    return v


def _same(a, b):
    for key in ("unit", "matched", "tie", "count"):
        assert np.array_equal(a[key].cpu().numpy(), b[key].cpu().numpy()), key
    assert np.array_equal(a["score"].cpu().numpy().view(np.uint64), b["score"].cpu().numpy().view(np.uint64))


def test_noalloc_matches_search_on_two_calls_in_a_row(gpu_index_factory):
    # head terms over 40k short docs: long lists that the planner splits by unit range (temporary lists + merge)
    c = small_corpus(n_docs=40_000, n_terms=3000, n_articles=500, body_min=20, body_max=60, title_len=2, desc_len=4)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    rng = np.random.default_rng(5)
    ix.set_url_rank(se.SRC_WEB, rng.permutation(c["n_docs"]).astype(np.uint32))
    big = [[0, 1, 2], [3, 9, 700], [5, 6, 2999], [], [4000], [7, 7, 8]] + \
          [list(map(int, q)) for q in O.gen_queries(3, c["n_terms"], O.zipf_cdf(c["n_terms"]), 1, 300, 3)] + \
          [list(range(10, 16))]
    small = big[:5]
    for k in (10, 220):
        for source in (se.SRC_WEB, se.SRC_NEWS):
            for qs in (big, small):           # the second call is the smaller one
                ref, got = _dev_out(len(qs), k), _dev_out(len(qs), k)
                ix.search(qs, k, source=source, out=ref)
                ix.search_noalloc(qs, k, got, source=source)
                _same(ref, got)
    # no device memory is taken or given back by the calls: free memory is the same before and after a series of them
    # (the outputs are allocated before the first reading)
    import torch

    got = _dev_out(len(big), 220)
    ix.search_noalloc(big, 220, got)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for qs, k in ((big, 220), (small, 220), (big[:100], 10), (big, 220)):
        ix.search_noalloc(qs, k, _noalloc_views(got, len(qs), k))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    # without the optional tie output
    ref, got = _dev_out(len(small), 10), _dev_out(len(small), 10)
    del got["tie"]
    ix.search(small, 10, out=ref)
    ix.search_noalloc(small, 10, got)
    assert np.array_equal(ref["unit"].cpu().numpy(), got["unit"].cpu().numpy())


def test_noalloc_limits_are_reported(gpu_index_factory):
    c = small_corpus(n_docs=300, n_terms=500)
    ix = gpu_index_factory()
    gpu_build(ix, c)
    out = _dev_out(2, 10)
    with pytest.raises(se.Sebm25Error, match="k must be"):
        ix.search_noalloc([[1], [2]], 300, _dev_out(2, 300))
    with pytest.raises(se.Sebm25Error, match="1024 queries"):
        ix.search_noalloc([[1]] * 1025, 10, _dev_out(1025, 10))
    with pytest.raises(se.Sebm25Error, match="device memory"):
        ix.search_noalloc([[1], [2]], 10, dict(unit=np.empty((2, 10), np.uint32), score=np.empty((2, 10)),
                                               matched=np.empty((2, 10), np.uint32), count=np.empty(2, np.uint32)))
    with pytest.raises(se.Sebm25Error, match="too many query terms"):
        ix.search_noalloc([list(range(t, t + 5)) for t in range(0, 45)] * 22, 10, _dev_out(990, 10))   # 4950 head terms
    ix.search_noalloc([[1], [2]], 10, out)   # the handle still works afterwards
