"""Two ranks, two processes, the real GPU engine in each (both on cuda:0), collectives over gloo:
the product's multi-rank path (sharded.py over native.Index) end to end, compared with one
unsharded index built in the parent.  The driver's 8-GPU run uses the same code with backend nccl
and one GPU per rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

N_DOCS, N_ART, V, K = 6000, 500, 20000, 12


def _worker(rank, world, port, path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import search_engine_amd as se
    from search_engine_amd.sharded import ShardedIndex, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ix = se.Index(0)
    d0, d1 = shard_range(N_DOCS, rank, world)
    a0, a1 = shard_range(N_ART, rank, world)
    spec = se.make_spec(n_terms=V, n_docs=d1 - d0, n_articles=a1 - a0, doc_begin=d0, article_begin=a0)
    n = ix.gen_corpus_rows(spec)
    term = torch.empty(n, dtype=torch.int32, device=dev)
    unit = torch.empty_like(term)
    tf = torch.empty_like(term)
    meta = torch.empty(n, dtype=torch.uint8, device=dev)
    wc = torch.empty(d1 - d0, dtype=torch.int32, device=dev)
    ix.gen_corpus(spec, term, unit, tf, meta, wc)
    sh = ShardedIndex(ix, dev)
    for _ in range(2):  # twice: the second build reuses every buffer
        t = sh.build(term, unit, tf, meta, V, d1 - d0, a1 - a0, wc, doc_base=d0, article_base=a0)
    st = ix.stats()
    qs = se.gen_queries(5, V, 40, 3, 5)
    web = sh.search(qs, K, source=se.SRC_WEB)
    news = sh.search(qs, K, source=se.SRC_NEWS)
    out = {}
    # the query-partitioned merge (large batches) against the all-gather merge; 37 queries: not a multiple of the ranks
    sh.partition_min_entries = 1
    for src, ref in ((se.SRC_WEB, web), (se.SRC_NEWS, news)):
        for nq in (40, 37):
            part = sh.search(qs[:nq], K, source=src)
            for key in ("unit", "score", "matched", "count"):
                np.testing.assert_array_equal(part[key], ref[key][:nq])
    # url ranks (a permutation of the global doc ids, each rank its shard's slice), the library on a stream of its own:
    # both merges again, now with tie-break keys in the exchange and the event hand-over between the streams
    perm = np.random.default_rng(3).permutation(N_DOCS).astype(np.uint32)
    own = torch.cuda.Stream(device=dev)
    ix.set_stream(own.cuda_stream)
    sh.lib_stream = own
    sh.set_url_rank(se.SRC_WEB, np.ascontiguousarray(perm[d0:d1]))
    for name, pme in (("ranked_part", 1), ("ranked_all", 16384)):
        sh.partition_min_entries = pme
        r = sh.search(qs, K, source=se.SRC_WEB)
        out.update({f"{name}_{k}": v for k, v in r.items()})
    sh.set_url_rank(se.SRC_WEB, None)
    ix.set_stream(None)
    sh.lib_stream = None
    np.savez(path % rank, df=st["df"], idf=st["idf"], n=st["n_docs"], avgdl=st["avgdl"],
             totals=np.array([t.n_done, t.n_wc, t.sum_wc, t.n_postings]), **out,
             **{f"web_{k}": v for k, v in web.items()}, **{f"news_{k}": v for k, v in news.items()})
    dist.barrier()
    dist.destroy_process_group()
    ix.close()


def test_two_processes_on_one_gpu_equal_one_unsharded_index(tmp_path, gpu_index_factory):
    import torch

    import search_engine_amd as se

    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "rank%d.npz")
    mp.start_processes(_worker, args=(world, port, path), nprocs=world, join=True, start_method="spawn")

    dev = torch.device("cuda", 0)
    ix = gpu_index_factory()
    spec = se.make_spec(n_terms=V, n_docs=N_DOCS, n_articles=N_ART)
    n = ix.gen_corpus_rows(spec)
    term = torch.empty(n, dtype=torch.int32, device=dev)
    unit = torch.empty_like(term)
    tf = torch.empty_like(term)
    meta = torch.empty(n, dtype=torch.uint8, device=dev)
    wc = torch.empty(N_DOCS, dtype=torch.int32, device=dev)
    ix.gen_corpus(spec, term, unit, tf, meta, wc)
    ix.build(term, unit, tf, meta, V, N_DOCS, N_ART, wc)
    st = ix.stats()
    qs = se.gen_queries(5, V, 40, 3, 5)
    ref = {se.SRC_WEB: ix.search(qs, K, source=se.SRC_WEB), se.SRC_NEWS: ix.search(qs, K, source=se.SRC_NEWS)}
    ix.set_url_rank(se.SRC_WEB, np.random.default_rng(3).permutation(N_DOCS).astype(np.uint32))
    ranked = ix.search(qs, K, source=se.SRC_WEB)
    ix.set_url_rank(se.SRC_WEB, None)
    for r in range(world):
        x = np.load(path % r)
        for name in ("ranked_part", "ranked_all"):
            np.testing.assert_array_equal(x[f"{name}_count"], ranked["count"])
            np.testing.assert_array_equal(x[f"{name}_unit"], ranked["unit"])
            np.testing.assert_array_equal(x[f"{name}_score"].view(np.uint64), ranked["score"].view(np.uint64))
        np.testing.assert_array_equal(x["df"], st["df"])                                  # all-reduced df
        np.testing.assert_array_equal(x["idf"].view(np.uint64), st["idf"].view(np.uint64))  # same idf bits on every rank
        assert int(x["n"]) == st["n_docs"] and float(x["avgdl"]) == st["avgdl"]
        assert int(x["totals"][3]) == ix.info().n_postings
        for src, name in ((se.SRC_WEB, "web"), (se.SRC_NEWS, "news")):
            np.testing.assert_array_equal(x[f"{name}_count"], ref[src]["count"])
            np.testing.assert_array_equal(x[f"{name}_unit"], ref[src]["unit"])
            np.testing.assert_array_equal(x[f"{name}_matched"], ref[src]["matched"])
            np.testing.assert_array_equal(x[f"{name}_score"].view(np.uint64), ref[src]["score"].view(np.uint64))


def _worker_row_limit(rank, world, port, path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    if rank == 1:
        os.environ["SEBM25_TEST_MAX_ROWS"] = "1000"     # this rank's shard is "past 2^32 - 4096 rows"
    import search_engine_amd as se
    from search_engine_amd.sharded import ShardedIndex, ShardError

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    ix = se.Index(0)
    spec = se.make_spec(n_terms=3000, n_docs=200, doc_begin=200 * rank)
    n = ix.gen_corpus_rows(spec)
    term = torch.empty(n, dtype=torch.int32, device=dev)
    unit = torch.empty_like(term)
    tf = torch.empty_like(term)
    meta = torch.empty(n, dtype=torch.uint8, device=dev)
    wc = torch.empty(200, dtype=torch.int32, device=dev)
    ix.gen_corpus(spec, term, unit, tf, meta, wc)
    sh = ShardedIndex(ix, dev)
    try:
        sh.build(term, unit, tf, meta, 3000, 200, 0, wc, doc_base=200 * rank)
        got = "no error"
    except ShardError as e:
        got = f"ShardError n_failed={e.n_failed} code={e.code}"
    with open(path % rank, "w") as f:
        f.write(got)
    dist.barrier()
    dist.destroy_process_group()
    ix.close()


def test_a_rank_past_the_row_limit_fails_the_build_on_every_rank(tmp_path):
    """A shard of more than 2^32 - 4096 rows is refused (SEBM25_E_RANGE) by its rank; every other rank must learn of it
    instead of waiting in the all-reduce.  The limit is lowered for one rank through SEBM25_TEST_MAX_ROWS."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "lim%d.txt")
    mp.start_processes(_worker_row_limit, args=(world, port, path), nprocs=world, join=True, start_method="spawn")
    assert open(path % 0).read() == "ShardError n_failed=1 code=None"
    assert open(path % 1).read() == "ShardError n_failed=1 code=2"
