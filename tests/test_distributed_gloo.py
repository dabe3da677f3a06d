"""The multi-rank path (shard -> all-reduce df/totals -> idf; per-shard top-k -> all-gather ->
merge) run with world_size 2 over gloo on CPU.  The per-shard engine is the CPU oracle behind
the interface of native.Index, so this exercises the product's collective logic
(search-engine_amd/sharded.py) without a GPU; the GPU engine itself is covered by the gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """native.Index look-alike backed by the CPU oracle (test stand-in for one GPU shard)."""

    def __init__(self):
        self.orc = None
        self.base = (0, 0)

    def build_local(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, unit_major=True):
        from oracle import oracle as O
        self.orc = O.OracleIndex(term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags)

        self.rows = [np.array(x) for x in (term_id, unit_id, tf, meta)]

    def update(self, term_id, unit_id, tf, meta, n_terms, n_docs, n_articles, word_count, doc_flags, changed_docs=None,
               changed_articles=None, unit_major=False, local=True):
        # the stand-in has no segments: it drops the changed units' rows from its copy of the shard and rebuilds
        o = self.rows
        src = o[3] & 3
        dead = ((src == 1) & np.isin(o[1], np.asarray(changed_docs if changed_docs is not None else [], np.uint32))) | \
               ((src == 2) & np.isin(o[1], np.asarray(changed_articles if changed_articles is not None else [], np.uint32)))
        rows = [np.concatenate([a[~dead], np.asarray(b, a.dtype)]) for a, b in zip(o, (term_id, unit_id, tf, meta))]
        order = np.lexsort((rows[1], rows[3] & 3))
        self.build_local(*[a[order] for a in rows], n_terms, n_docs, n_articles, word_count, doc_flags)

    def set_unit_base(self, web, news):
        self.base = (web, news)

    def local_stats(self, df_out):
        df_out.copy_(torch.from_numpy(self.orc.df.astype(np.int32)))
        return df_out, self.orc.totals

    def finalize(self, df, totals):
        from oracle import oracle as O
        self.orc.finalize(df.numpy().astype(np.uint32), O.Totals(totals.n_done, totals.n_wc, totals.sum_wc, totals.n_postings))

    def search(self, queries, k, source, out):
        base = self.base[0] if source == 1 else self.base[1]
        for key in ("unit", "tie"):
            if key in out:
                out[key].fill_(-1)
        out["score"].zero_(); out["matched"].zero_()
        for i, q in enumerate(queries):
            u, s, m = self.orc.search(np.asarray(q, np.uint32), k, source=source, unit_base=base, level="csr")
            n = len(u)
            out["count"][i] = n
            out["unit"][i, :n] = torch.from_numpy(u.astype(np.int32))
            if "tie" in out:
                out["tie"][i, :n] = torch.from_numpy(u.astype(np.int32))     # default tie-break key: global unit id
            out["score"][i, :n] = torch.from_numpy(s)
            out["matched"][i, :n] = torch.from_numpy(m.astype(np.int32))

    def merge_topk(self, n_lists, nq, k, unit, score, matched, tie, count, out):
        out["unit"].fill_(-1); out["score"].zero_(); out["matched"].zero_(); out["tie"].fill_(-1)   # padding as the library writes it
        for q in range(nq):
            c = []
            for l in range(n_lists):
                for i in range(int(count[l, q])):
                    c.append((-float(score[l, q, i]), int(tie[l, q, i]) & 0xFFFFFFFF, int(unit[l, q, i]), int(matched[l, q, i])))
            c.sort()
            c = c[:k]
            out["count"][q] = len(c)
            for i, (ns, t, u, m) in enumerate(c):
                out["unit"][q, i] = u; out["score"][q, i] = -ns; out["matched"][q, i] = m; out["tie"][q, i] = t


def _worker(rank, world, port, n_docs, n_art, V, result_path, update=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from search_engine_amd.sharded import ShardedIndex, shard_range

    d0, d1 = shard_range(n_docs, rank, world)
    a0, a1 = shard_range(n_art, rank, world)
    c = O.gen_corpus(O.make_spec(n_terms=V, n_docs=d1 - d0, n_articles=a1 - a0, doc_begin=d0, article_begin=a0,
                                 body_min=30, body_max=90))
    sh = ShardedIndex(OracleEngine(), "cpu")
    t = sh.build(c["term_id"], c["unit_id"], c["tf"], c["meta"], V, d1 - d0, a1 - a0, c["word_count"],
                 doc_base=d0, article_base=a0)
    qs = O.gen_queries(3, V, O.zipf_cdf(V), 5, 20, 3)
    web = sh.search(qs, 12, source=1)
    news = sh.search(qs, 12, source=2)
    # the query-partitioned merge (what large batches take) gives the same rows as the all-gather merge; 19 queries: a
    # count the ranks do not divide
    small_bytes = sh.last_exchange_bytes
    sh.partition_min_entries = 1
    for src, ref in ((1, web), (2, news)):
        for n in (20, 19):
            part = sh.search(qs[:n], 12, source=src)
            for key in ("unit", "score", "matched", "count"):
                np.testing.assert_array_equal(part[key], ref[key][:n])
    assert sh.last_exchange_bytes > 0 and small_bytes > 0
    sh.partition_min_entries = 16384
    if update:
        # the first five docs of every shard are crawled again (rows of a corpus with another seed)
        c2 = O.gen_corpus(O.make_spec(seed=99, n_terms=V, n_docs=d1 - d0, n_articles=0, doc_begin=d0, body_min=30, body_max=90))
        sel = c2["unit_id"] < 5
        wc = c["word_count"].copy()
        wc[:5] = c2["word_count"][:5]
        t = sh.update(c2["term_id"][sel], c2["unit_id"][sel], c2["tf"][sel], c2["meta"][sel], V, d1 - d0, a1 - a0, wc,
                      changed_docs=np.arange(5, dtype=np.uint32))
        web = sh.search(qs, 12, source=1)
        news = sh.search(qs, 12, source=2)
    np.savez(result_path % rank, df=sh.engine.orc.df, idf=sh.engine.orc.idf, df_global=sh._df.numpy(),
             totals=np.array([t.n_done, t.n_wc, t.sum_wc, t.n_postings]), n=sh.engine.orc.n_docs_stat,
             avgdl=sh.engine.orc.avgdl, **{f"web_{k}": v for k, v in web.items()}, **{f"news_{k}": v for k, v in news.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_build_and_search_equal_unsharded(tmp_path):
    from oracle import oracle as O

    n_docs, n_art, V, world = 400, 50, 2000, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "rank%d.npz")
    mp.start_processes(_worker, args=(world, port, n_docs, n_art, V, path), nprocs=world, join=True, start_method="spawn")
    full = O.gen_corpus(O.make_spec(n_terms=V, n_docs=n_docs, n_articles=n_art, body_min=30, body_max=90))
    orc = O.OracleIndex(full["term_id"], full["unit_id"], full["tf"], full["meta"], V, n_docs, n_art, full["word_count"])
    qs = O.gen_queries(3, V, O.zipf_cdf(V), 5, 20, 3)
    r = [np.load(path % i) for i in range(world)]
    np.testing.assert_array_equal(r[0]["df"] + r[1]["df"], orc.df)          # local df sums to the global df
    for x in r:
        np.testing.assert_array_equal(x["df_global"].view(np.uint32), orc.df)   # all-reduced, bit-exact
        np.testing.assert_array_equal(x["idf"], orc.idf)                    # identical idf on every rank
        assert int(x["n"]) == orc.n_docs_stat and float(x["avgdl"]) == orc.avgdl
        t = orc.totals
        assert x["totals"].tolist() == [t.n_done, t.n_wc, t.sum_wc, t.n_postings]
        for src, name in ((1, "web"), (2, "news")):
            for i, q in enumerate(qs):
                u, sc, m = orc.search(q, 12, source=src, level="csr")
                n = int(x[f"{name}_count"][i])
                assert n == len(u)
                np.testing.assert_array_equal(x[f"{name}_unit"][i, :n], u)
                np.testing.assert_array_equal(x[f"{name}_score"][i, :n], sc)
                np.testing.assert_array_equal(x[f"{name}_matched"][i, :n], m)


def test_two_rank_sharded_update_equals_unsharded_rebuild(tmp_path):
    """Every shard replaces a few of its docs (sebm25_update_local's contract), df and totals are summed again:
    statistics and both searches equal an unsharded index built from the final table."""
    from oracle import oracle as O
    from search_engine_amd.sharded import shard_range

    n_docs, n_art, V, world = 400, 50, 2000, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "rank%d.npz")
    mp.start_processes(_worker, args=(world, port, n_docs, n_art, V, path, True), nprocs=world, join=True, start_method="spawn")
    a = O.gen_corpus(O.make_spec(n_terms=V, n_docs=n_docs, n_articles=n_art, body_min=30, body_max=90))
    b = O.gen_corpus(O.make_spec(seed=99, n_terms=V, n_docs=n_docs, n_articles=0, body_min=30, body_max=90))
    changed = np.concatenate([np.arange(shard_range(n_docs, r, world)[0], shard_range(n_docs, r, world)[0] + 5) for r in range(world)])
    keep = ~(((a["meta"] & 3) == 1) & np.isin(a["unit_id"], changed))
    take = np.isin(b["unit_id"], changed)
    rows = [np.concatenate([a[k][keep], b[k][take]]) for k in ("term_id", "unit_id", "tf", "meta")]
    order = np.lexsort((rows[1], rows[3] & 3))
    wc = a["word_count"].copy()
    wc[changed] = b["word_count"][changed]
    orc = O.OracleIndex(*[x[order] for x in rows], V, n_docs, n_art, wc)
    qs = O.gen_queries(3, V, O.zipf_cdf(V), 5, 20, 3)
    for i in range(world):
        x = np.load(path % i)
        np.testing.assert_array_equal(x["df_global"].view(np.uint32), orc.df)
        np.testing.assert_array_equal(x["idf"], orc.idf)
        assert int(x["n"]) == orc.n_docs_stat and float(x["avgdl"]) == orc.avgdl
        for src, name in ((1, "web"), (2, "news")):
            for qi, q in enumerate(qs):
                u, sc, m = orc.search(q, 12, source=src, level="csr")
                n = int(x[f"{name}_count"][qi])
                assert n == len(u)
                np.testing.assert_array_equal(x[f"{name}_unit"][qi, :n], u)
                np.testing.assert_array_equal(x[f"{name}_score"][qi, :n], sc)


def test_shard_ranges_cover_everything():
    from search_engine_amd.sharded import shard_range

    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[g][1] == r[g + 1][0] for g in range(world - 1))


def _worker_growth_and_failure(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from search_engine_amd.native import Sebm25Error
    from search_engine_amd.sharded import ShardedIndex, ShardError

    V, per = 1500, 100
    c = O.gen_corpus(O.make_spec(n_terms=V, n_docs=per, doc_begin=rank * per, body_min=20, body_max=60))
    sh = ShardedIndex(OracleEngine(), "cpu")
    out = {}
    # (1) bases with headroom: rank r owns global ids [200 r, 200 r + 100)
    sh.build(c["term_id"], c["unit_id"], c["tf"], c["meta"], V, per, 0, c["word_count"], doc_base=200 * rank)
    # (2) rank 0 gains 150 docs: its range would reach into rank 1's -> ValueError on every rank, nothing changed
    grow = 150 if rank == 0 else 0
    z32, z8 = np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    wc = np.concatenate([c["word_count"], np.zeros(grow, np.uint32)])
    try:
        sh.update(z32, z32, z32, z8, V, per + grow, 0, wc)
        out["overlap"] = "no error"
    except ValueError as e:
        out["overlap"] = str(e)
    assert sh.n_docs == per
    # (3) rank 0 gains 50 docs (inside its headroom), with rows: fine, and the new docs are found under their global ids
    grow = 50 if rank == 0 else 0
    n2 = per + grow
    c2 = O.gen_corpus(O.make_spec(seed=5, n_terms=V, n_docs=grow, body_min=20, body_max=60)) if grow else None
    wc = np.concatenate([c["word_count"], c2["word_count"]]) if grow else c["word_count"]
    if grow:
        sh.update(c2["term_id"], c2["unit_id"] + np.uint32(per), c2["tf"], c2["meta"], V, n2, 0, wc,
                  changed_docs=np.arange(per, n2, dtype=np.uint32), unit_major=True)
    else:
        sh.update(z32, z32, z32, z8, V, n2, 0, wc)
    qs = O.gen_queries(3, V, O.zipf_cdf(V), 5, 15, 3)
    res = sh.search(qs, 10, source=1)
    out.update({f"grown_{k}": v for k, v in res.items()})
    # (4) the local step fails on rank 1 only: both ranks get ShardError, nobody waits in the all-reduce
    class Failing(OracleEngine):
        def build_local(self, *a, **kw):
            raise Sebm25Error(2, "build: more than 2^32 - 4096 rows in one shard")
    sh2 = ShardedIndex(Failing() if rank == 1 else OracleEngine(), "cpu")
    try:
        sh2.build(c["term_id"], c["unit_id"], c["tf"], c["meta"], V, per, 0, c["word_count"], doc_base=200 * rank)
        out["failure"] = "no error"
    except ShardError as e:
        out["failure"] = f"ShardError n_failed={e.n_failed} code={e.code}"
    np.savez(result_path % rank, **out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_growth_needs_headroom_and_a_failing_rank_fails_all(tmp_path):
    from oracle import oracle as O

    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "g%d.npz")
    mp.start_processes(_worker_growth_and_failure, args=(world, port, path), nprocs=world, join=True, start_method="spawn")
    r = [np.load(path % i) for i in range(world)]
    for x in r:
        assert "overlap" in str(x["overlap"]) and "document id ranges" in str(x["overlap"])
    assert str(r[0]["failure"]) == "ShardError n_failed=1 code=None"      # rank 0's own step was fine
    assert str(r[1]["failure"]) == "ShardError n_failed=1 code=2"
    # the grown index equals an unsharded one over the same global ids: docs 0..149 (rank 0) and 200..299 (rank 1)
    V, per = 1500, 100
    a = O.gen_corpus(O.make_spec(n_terms=V, n_docs=per, doc_begin=0, body_min=20, body_max=60))
    b = O.gen_corpus(O.make_spec(n_terms=V, n_docs=per, doc_begin=per, body_min=20, body_max=60))
    g = O.gen_corpus(O.make_spec(seed=5, n_terms=V, n_docs=50, body_min=20, body_max=60))
    rows = {k: np.concatenate([a[k], g[k] if k != "unit_id" else g[k] + np.uint32(per), b[k] if k != "unit_id" else b[k] + np.uint32(200)])
            for k in ("term_id", "unit_id", "tf", "meta")}
    wc = np.zeros(300, np.uint32)
    wc[:100], wc[100:150], wc[200:300] = a["word_count"], g["word_count"], b["word_count"]
    flags = np.zeros(300, np.uint8)
    flags[:150] = 1
    flags[200:] = 1
    orc = O.OracleIndex(rows["term_id"], rows["unit_id"], rows["tf"], rows["meta"], V, 300, 0, wc, flags)
    qs = O.gen_queries(3, V, O.zipf_cdf(V), 5, 15, 3)
    for x in r:
        for qi, q in enumerate(qs):
            u, sc, m = orc.search(q, 10, source=1, level="csr")
            n = int(x["grown_count"][qi])
            assert n == len(u)
            np.testing.assert_array_equal(x["grown_unit"][qi, :n], u)
            np.testing.assert_array_equal(x["grown_score"][qi, :n], sc)
