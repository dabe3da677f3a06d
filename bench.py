#!/usr/bin/env python3
"""Benchmark of the BM25 index-build path and the batched top-k scoring path on MI355X.

One "step" = one full statistics job over this rank's shard of the synthetic Zipfian corpus, rows
already resident in HBM: radix sort whose last pass writes the posting slots -> df / document totals ->
(N > 1: one all-reduce of df[V] and the document totals over RCCL) -> idf.  The headline runs the
variant the drop-in runs: `doc_flags` supplied (every doc 'done'); scoring breaks ties on the unit id, as an index
does whose docs the export numbered in url order (explicit rank arrays are timed as a variant).

Workloads
  N = 1 : build = benchmark configuration 2 (1M docs, 500k-term vocabulary, ~300 draws per doc);
          scoring = configuration 4's one-GPU leg (10k queries x 3 terms over a 10M-doc index).
  N > 1 : every rank builds a 1.25M-doc shard (at N = 8 this is configuration 3: 10M docs over the
          node); scoring goes through ShardedIndex.search (local top-k, one all-gather, merge), which at
          N = 8 is configuration 4's eight-GPU leg.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--docs D] [--query-docs D] ...

Rank 0 prints one JSON line.  `value` = token rows of all ranks x K / the max-over-ranks wall time of
exactly K steps (timing events off).  Per-kernel times come from a separate pass with HIP events on
the library's stream.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
PROFILE_TAG = "r3"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=0, help="web docs per GPU (0: 1M at N = 1, 1.25M at N > 1)")
    ap.add_argument("--articles", type=int, default=0, help="news articles per GPU")
    ap.add_argument("--vocab", type=int, default=500_000)
    ap.add_argument("--query-docs", type=int, default=-1,
                    help="N = 1: docs of the index the query batch runs over (-1: 10M; 0: the build leg's index)")
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--terms-per-query", type=int, default=3)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--k2", type=int, default=210, help="second scoring run at this k (0 = skip): the candidate "
                    "limit of the reference's top-10 call")
    ap.add_argument("--query-steps", type=int, default=20)
    ap.add_argument("--cpu-docs", type=int, default=1_000_000, help="docs in the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-query", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the flag-less / rank-less comparison runs")
    ap.add_argument("--no-update", action="store_true", help="skip the incremental-update leg")
    ap.add_argument("--update-docs", type=int, default=1000, help="docs re-crawled per incremental update")
    return ap.parse_args()


# ------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, rows):
    """The CPU oracle (a C restatement of the reference's SQL) and the sqlite stand-in, timed on
    host cores.  Checker code, never the product path.  Tiers (never "the reference": PostgreSQL
    is not on this box):
      T3/1   the single-threaded oracle on the first `cpu_docs` docs of the benchmark corpus
      T3/P   the same oracle doc-sharded over P threads (df summed, top-k lists merged)
      T2     stdlib sqlite3 executing the equivalent statements on the 10k-doc configuration"""
    import concurrent.futures as cf
    import heapq

    import numpy as np

    from oracle import oracle as O
    from oracle import sqlite_standin as S

    term, unit, tf, meta, wc, n_docs = rows
    n = len(term)
    V = args.vocab
    ks = [args.k] + ([args.k2] if args.k2 else [])
    nq = 500
    qs = O.gen_queries(20260228, V, O.zipf_cdf(V), 100, nq, args.terms_per_query)

    # ---- T3, one core
    t0 = time.perf_counter()
    orc = O.OracleIndex(term, unit, tf, meta, V, n_docs, 0, wc)
    t_build = time.perf_counter() - t0
    q1 = {}
    for k in ks:
        t0 = time.perf_counter()
        for q in qs:
            orc.search(q, k, level="rows")
        q1[k] = nq / (time.perf_counter() - t0)
    df_ref = orc.df
    ref_top = [orc.search(q, args.k, level="rows") for q in qs[:20]]
    orc.close()

    # ---- T3, P threads over doc shards (ctypes releases the GIL inside the oracle)
    P = max(1, min(16, os.cpu_count() or 1))
    bounds = [n_docs * p // P for p in range(P + 1)]
    cuts = np.searchsorted(unit, np.asarray(bounds[:-1], np.uint32), side="left").tolist() + [n]   # web rows are doc-major

    def build_shard(p):
        a, b = cuts[p], cuts[p + 1]
        return O.OracleIndex(term[a:b], unit[a:b] - np.uint32(bounds[p]), tf[a:b], meta[a:b], V, bounds[p + 1] - bounds[p], 0,
                             wc[bounds[p]:bounds[p + 1]])

    with cf.ThreadPoolExecutor(P) as ex:
        t0 = time.perf_counter()
        shards = list(ex.map(build_shard, range(P)))
        df_sum = np.zeros(V, np.uint64)
        tot = O.Totals(0, 0, 0, 0)
        for s in shards:
            df_sum += s.df
            t = s.totals
            tot = O.Totals(tot.n_done + t.n_done, tot.n_wc + t.n_wc, tot.sum_wc + t.sum_wc, tot.n_postings + t.n_postings)
        df_sum = df_sum.astype(np.uint32)
        list(ex.map(lambda s: s.finalize(df_sum, tot), shards))
        t_build_p = time.perf_counter() - t0
        assert np.array_equal(df_sum, df_ref), "sharded CPU df differs from the unsharded one"
        qp = {}
        for k in ks:
            t0 = time.perf_counter()
            parts = list(ex.map(lambda p: [shards[p].search(q, k, unit_base=bounds[p], level="rows") for q in qs], range(P)))
            merged = []
            for qi in range(nq):
                cand = [(-s, int(u), int(m)) for p in range(P) for u, s, m in zip(*parts[p][qi])]
                merged.append(heapq.nsmallest(k, cand))
            qp[k] = nq / (time.perf_counter() - t0)
            if k == args.k:
                for qi, (ru, rs, rm) in enumerate(ref_top):
                    assert [c[1] for c in merged[qi]] == [int(u) for u in ru], "sharded CPU top-k differs"
    for s in shards:
        s.close()

    # ---- T2, sqlite stand-in on the 10k-doc configuration
    c1 = O.gen_corpus(O.make_spec(n_terms=V, n_docs=10_000))
    t2 = S.time_standin(c1, qs[:200], ks=ks)

    return {
        "value": n / t_build, "unit": "token rows/s", "cores": 1, "kind": "port",
        "sample": f"{n_docs} docs / {n} rows of the same synthetic corpus, single-threaded C oracle "
                  f"(sort + collapse + df + idf), {t_build:.2f} s",
        "queries_per_s": q1[args.k],
        "query_sample": f"{nq} queries x {args.terms_per_query} terms, top-{args.k}, over the {n_docs}-doc sample",
        "host_cpus": os.cpu_count(),
        "tiers": {
            "T1_reference_postgres": "unavailable: no PostgreSQL server on this box",
            "T2_sqlite_standin": {
                "label": "sqlite stand-in, not Postgres", "cores": 1, "corpus": "10k docs (configuration 1)",
                "rows": t2["rows"], "build_rows_per_s": t2["build_rows_per_s"], "build_s": t2["build_s"],
                **{f"top{k}_queries_per_s": t2[f"top{k}_queries_per_s"] for k in ks}, "queries": t2["queries"]},
            "T3_oracle_1_core": {
                "label": "CPU restatement (C oracle), 1 core", "cores": 1, "corpus": f"{n_docs} docs",
                "build_rows_per_s": n / t_build, "build_s": t_build,
                **{f"top{k}_queries_per_s": q1[k] for k in ks}, "queries": nq},
            f"T3_oracle_{P}_threads": {
                "label": f"CPU restatement (C oracle), doc-sharded over {P} threads, df summed and top-k merged",
                "cores": P, "corpus": f"{n_docs} docs", "build_rows_per_s": n / t_build_p, "build_s": t_build_p,
                **{f"top{k}_queries_per_s": qp[k] for k in ks}, "queries": nq},
        },
    }


def load_traffic():
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# ------------------------------------------------------------------------------------ main
def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import search_engine_amd as se
    from search_engine_amd.sharded import ShardedIndex

    if os.environ.get("SEBM25_BENCH_WATCHDOG"):
        # debugging aid: dump every thread's stack and exit if the run takes longer than this many seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["SEBM25_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # SEBM25_BENCH_REHEARSAL=1: every rank on GPU 0 and collectives over gloo, to rehearse the
    # multi-rank code path on a one-GPU box (the numbers of such a run mean nothing)
    rehearsal = os.environ.get("SEBM25_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    docs = args.docs or (1_000_000 if world == 1 else 1_250_000)
    V = args.vocab

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_corpus(ix, n_docs, n_articles, doc_begin, article_begin):
        spec = se.make_spec(n_terms=V, n_docs=n_docs, n_articles=n_articles, doc_begin=doc_begin, article_begin=article_begin)
        n = ix.gen_corpus_rows(spec)
        c = dict(term=torch.empty(n, dtype=torch.int32, device=dev), unit=torch.empty(n, dtype=torch.int32, device=dev),
                 tf=torch.empty(n, dtype=torch.int32, device=dev), meta=torch.empty(n, dtype=torch.uint8, device=dev),
                 wc=torch.empty(n_docs, dtype=torch.int32, device=dev),
                 flags=torch.full((n_docs,), se.DOC_DONE, dtype=torch.uint8, device=dev), n=n)
        ix.gen_corpus(spec, c["term"], c["unit"], c["tf"], c["meta"], c["wc"])
        return c

    ix = se.Index(local_rank)
    ix.set_stream(torch.cuda.current_stream().cuda_stream)
    c = make_corpus(ix, docs, args.articles, rank * docs, rank * args.articles)
    n_rows = c["n"]
    sharded = ShardedIndex(ix, dev) if world > 1 else None

    def step(with_flags=True):
        fl = c["flags"] if with_flags else None
        if world == 1:
            ix.build(c["term"], c["unit"], c["tf"], c["meta"], V, docs, args.articles, c["wc"], fl)
        else:
            # build_local -> one all-reduce of df[V] and the document totals (RCCL over xGMI) -> idf
            sharded.build(c["term"], c["unit"], c["tf"], c["meta"], V, docs, args.articles, c["wc"], fl,
                          doc_base=rank * docs, article_base=rank * args.articles)

    # ---- the timed region: exactly K steps between two barriers + device syncs, no timing events
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if world > 1:
        rows_all = torch.tensor([n_rows], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(rows_all)
        total_rows = int(rows_all.item())
    else:
        total_rows = n_rows
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rows * args.steps / elapsed

    # ---- separate passes: every step timed alone (median), then HIP events around each kernel
    n_single = max(args.steps, 20)
    singles = []
    for _ in range(n_single):
        sync()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        singles.append((time.perf_counter() - t0) * 1e3)
    median_ms = max_over_ranks(statistics.median(singles))

    def kernel_phases(with_flags, n_steps=10):
        ix.enable_timing(True)
        acc: dict[str, float] = {}
        for _ in range(n_steps):
            step(with_flags)
            for name, ms in ix.timings():
                acc[name] = acc.get(name, 0.0) + ms
        ix.enable_timing(False)
        return {n: ms / n_steps for n, ms in acc.items()}

    phases = kernel_phases(True)
    info = ix.info()

    # ---- roofline of the dominant kernel (per GPU, per launch)
    # algorithmic bytes per input row of each build kernel (what it must read + what it must write)
    KERNELS = {
        "sort_hist_0": ("k_sort_hist<0>", 4), "sort_hist_1": ("k_sort_hist<2>", 8),
        "sort_scatter_0": ("k_sort_scatter_tile<0, false>", 13 + 8),  # raw row in, packed row out
        "sort_scatter_1": ("k_sort_scatter_tile<2, false>", 8 + 8),   # packed row in, packed row out
        "scatter_postings": ("k_sort_scatter_tile<2, true>", 8 + 8),  # packed row in, posting slot out (the last pass)
        "postings": ("k_postings", 8 + 8),                            # packed row in, posting slot out (separate pass)
    }
    traffic_db = load_traffic()
    build_traffic = traffic_db.get("build", {}) if traffic_db.get("build", {}).get("rows") == int(info.n_rows) else {}
    dev_ms = sum(phases.values())
    dom_name, dom_ms = max(phases.items(), key=lambda p: p[1]) if phases else ("", 0.0)
    kname, bpr = KERNELS.get(dom_name, (dom_name, 0))
    dom_bytes = bpr * info.n_rows
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
    traffic = build_traffic.get("kernels", {}).get(kname, {}).get("hbm_bytes_per_launch")
    # the whole build against the build's algorithmic bytes: 13 B/row in, 8 B/posting out, 4 B/doc, 20 B/term
    alg_bytes = 13 * info.n_rows + 8 * info.n_postings + 4 * info.n_docs + 20 * info.n_terms
    pipe = alg_bytes / (median_ms * 1e-3) / 1e9
    per_kernel = {}
    for ph, ms in phases.items():
        if ph in KERNELS and ms:
            kn, b = KERNELS[ph]
            per_kernel[kn] = {"ms": round(ms, 4), "alg_bytes_per_row": b, "frac": round(b * info.n_rows / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "traffic": build_traffic.get("kernels", {}).get(kn, {}).get("hbm_bytes_per_launch")}
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "kernel": kname, "kernel_ms": dom_ms, "alg_bytes_per_row": bpr, "alg_bytes_per_launch": dom_bytes,
        "pipeline": {"alg_bytes": alg_bytes, "achieved": pipe, "frac": pipe / HBM_PEAK_GBS,
                     "scope": "whole build of one GPU: 13 B/row + 8 B/posting + 4 B/doc + 20 B/term over the median "
                              "wall time of one step", "device_ms_sum_of_phases": dev_ms,
                     "traffic": sum(v.get("hbm_bytes_per_launch", 0) for v in build_traffic.get("kernels", {}).values()) or None},
        "kernels": per_kernel,
        "phases_ms": {n: round(ms, 4) for n, ms in phases.items()},
    }

    if world == 1:
        workload = (f"BM25 index build, {docs} docs, vocab {V}, Zipf s=1, ~300 body draws/doc "
                    f"(benchmark configuration 2), doc status flags supplied")
    else:
        workload = (f"BM25 index build, {docs} docs per GPU x {world} GPUs = {docs * world} docs, vocab {V}, Zipf s=1 "
                    f"(benchmark configuration 3 when {world} x {docs} = 10M), df / totals all-reduce per step")
    out = {
        "metric": "bm25_index_build_token_rows_per_s", "value": value, "unit": "token rows/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_step_median": median_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "rows_per_gpu": int(info.n_rows), "postings_per_gpu": int(info.n_postings),
                   "sort_passes": int(info.sort_passes), "parallelism": f"doc-sharded x{world}"},
        "roofline": roofline,
        "arithmetic": "u64 packed rows, u32 counts and posting weights (exact tenths), f64 idf and scores",
    }
    if not args.no_variants:
        ph2 = kernel_phases(False, 5)
        out["variant_without_doc_flags"] = {"device_ms_sum_of_phases": sum(ph2.values()),
                                            "sort_scatter_0_ms": ph2.get("sort_scatter_0"),
                                            "note": "doc_flags = NULL: the first sort pass reads no status bytes"}

    # ---- incremental update: a crawl cycle's worth of re-crawled docs applied to the built index
    if world == 1 and not args.no_update:
        upd_docs = min(args.update_docs, docs // 8)
        rows_u = 0
        step()                                             # a fresh base segment
        cycles = []
        for cyc in range(2):                               # the second cycle reuses the first one's segment storage
            times = []
            for i in range(4):
                a = ((cyc * 4 + i) * 2 + 1) * docs // 17   # docs [a, a + upd_docs) are crawled again with new text
                spec = se.make_spec(seed=777 + cyc * 4 + i, n_terms=V, n_docs=upd_docs, doc_begin=a)
                n_u = ix.gen_corpus_rows(spec)
                u = dict(term=torch.empty(n_u, dtype=torch.int32, device=dev), unit=torch.empty(n_u, dtype=torch.int32, device=dev),
                         tf=torch.empty(n_u, dtype=torch.int32, device=dev), meta=torch.empty(n_u, dtype=torch.uint8, device=dev),
                         wc=torch.empty(upd_docs, dtype=torch.int32, device=dev))
                ix.gen_corpus(spec, u["term"], u["unit"], u["tf"], u["meta"], u["wc"])
                u["unit"] += a                             # generated ids are local to the range
                c["wc"][a:a + upd_docs] = u["wc"]
                changed = np.arange(a, a + upd_docs, dtype=np.uint32)
                sync()
                t0 = time.perf_counter()
                ix.update(u["term"], u["unit"], u["tf"], u["meta"], V, docs, args.articles, c["wc"], c["flags"], changed_docs=changed,
                          unit_major=True)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
                rows_u = n_u
            inf_u = ix.info()
            sync()
            t0 = time.perf_counter()
            ix.compact()
            torch.cuda.synchronize()
            cycles.append(dict(ms_each=[round(t, 3) for t in times], compact_ms=(time.perf_counter() - t0) * 1e3,
                               segments=int(inf_u.n_segments), dead_rows=int(inf_u.n_dead_rows)))
        last = cycles[-1]
        out["incremental_update"] = {
            "docs_per_update": upd_docs, "rows_per_update": int(rows_u), "ms_per_update": statistics.median(last["ms_each"]),
            "ms_each": last["ms_each"], "full_build_ms": median_ms,
            "segments_after": last["segments"], "dead_rows_after": last["dead_rows"], "compact_ms": last["compact_ms"],
            "first_cycle": cycles[0],
            "note": "four sebm25_update calls of re-crawled docs on the configuration-2 index (rows resident in HBM; each folds "
                    "its rows into the one update segment), then sebm25_compact (base and update segment merged), twice: the "
                    "first cycle allocates the segments' storage, the second reuses it; correctness of both is covered by "
                    "tests/test_gpu_update.py"}

    # ---- batched scoring
    if not args.no_query:
        q_ix, q_docs, q_sharded = ix, docs, sharded
        if world == 1 and args.query_docs != 0:
            q_docs = 10_000_000 if args.query_docs < 0 else args.query_docs
        if world == 1 and q_docs != docs:
            # the corpus of the build leg is not needed any more: free it before the larger index is made
            del c
            torch.cuda.empty_cache()
            q_ix = se.Index(local_rank)
            q_ix.set_stream(torch.cuda.current_stream().cuda_stream)
            qc = make_corpus(q_ix, q_docs, 0, 0, 0)
            q_ix.build(qc["term"], qc["unit"], qc["tf"], qc["meta"], V, q_docs, 0, qc["wc"], qc["flags"])
            q_rows = qc["n"]
            del qc
            torch.cuda.empty_cache()
        else:
            q_rows = n_rows
        # url ranks as the drop-in sets them: the synthetic urls sort like the doc ids, so rank = global doc id
        ident = torch.arange(rank * q_docs, (rank + 1) * q_docs, dtype=torch.int64, device=dev).to(torch.int32)
        qs = se.gen_queries(20260228, V, args.queries, args.terms_per_query, 100)
        df_local, _ = q_ix.local_stats()
        sum_df = int(df_local.astype(np.int64)[qs.reshape(-1)].sum())

        def run_queries(k, steps, with_rank=True, appended_keys=False):
            # (through the sharded index when there is one: the exchange carries tie-break keys only when ranks are set)
            if appended_keys:
                # the index after updates that appended 1 % of its docs: keys that need no array for the rest, the appended
                # docs spread evenly over the url order (sebm25_set_url_rank_appended)
                nb = q_docs - q_docs // 100
                gaps = torch.arange(q_docs - nb, dtype=torch.int64, device=dev) * nb // max(q_docs - nb, 1)
                q_ix.set_url_rank_appended(se.SRC_WEB, nb, 8, ((gaps << 8) + 1).to(torch.int32))
            else:
                (q_sharded if q_sharded is not None else q_ix).set_url_rank(se.SRC_WEB, ident if with_rank else None)
            res = {key: torch.empty((args.queries, k), dtype=dt, device=dev) for key, dt in
                   (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
            res["count"] = torch.empty(args.queries, dtype=torch.int32, device=dev)

            def one():
                if q_sharded is not None:
                    q_sharded.search(qs, k, out_device=True)
                else:
                    q_ix.search(qs, k, out=res)
            one()
            one()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            sync()
            wall = max_over_ranks(time.perf_counter() - t0)
            q_ix.enable_timing(True)
            one()
            ph = dict(q_ix.timings())
            q_ix.enable_timing(False)
            return wall / steps * 1e3, ph

        # the drop-in numbers docs in url order at export, so a freshly built index breaks ties on the unit id and holds
        # no rank array; explicit ranks (an index that took updates which added docs) are the variant below
        q_ms, qph = run_queries(args.k, args.query_steps, with_rank=False)
        alg_q = 8 * sum_df + args.queries * (13 * args.k + 48)
        score_ms = qph.get("score_topk", 0.0)
        q_traffic = None
        tq = traffic_db.get("query", {})
        if tq.get("postings_scored_per_batch") == sum_df:
            q_traffic = tq.get("hbm_bytes_per_launch")
        kq = ("k_score_join<3, RegTopK<false>, false, 6, 4, 8>" if args.k <= 48 else
              "k_score_join<3, TopK<192, false, false>, false, 4, 3, 8>" if args.k <= 128 else "k_score_join<3, TopK<304, false, false>, false, 4, 3, 8>")
        out["query"] = {
            "metric": f"top{args.k}_queries_per_s", "value": args.queries / (q_ms * 1e-3), "unit": "queries/s",
            "ms_per_batch": q_ms, "queries": args.queries, "terms_per_query": args.terms_per_query, "k": args.k,
            "workload": (f"10k x 3-term batch over a {q_docs}-doc index on one GPU (benchmark configuration 4, one-GPU leg), "
                         f"ties broken on the unit id (docs numbered in url order)" if world == 1 else
                         f"10k x 3-term batch over {q_docs} docs per GPU x {world} GPUs: local top-k, all-to-all of the lists by query "
                         f"partition, merge of 1/{world} of the queries per rank, all-gather of the merged rows (benchmark configuration 4, "
                         f"{world}-GPU leg when the shards add up to 10M docs)"),
            "index_rows_per_gpu": int(q_rows), "postings_scored_per_batch_per_gpu": sum_df,
            "roofline": {"bound": "hbm", "achieved": alg_q / (score_ms * 1e-3) / 1e9 if score_ms else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_q / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if score_ms else None,
                         "traffic": q_traffic, "kernel": kq, "kernel_ms": score_ms, "alg_bytes": alg_q,
                         "phases_ms": {n: round(ms, 4) for n, ms in qph.items()}},
        }
        if args.k2 > 0:
            q2_ms, q2ph = run_queries(args.k2, max(3, args.query_steps // 4), with_rank=False)
            out["query"]["candidate_limit_run"] = {
                "k": args.k2, "value": args.queries / (q2_ms * 1e-3), "unit": "queries/s", "ms_per_batch": q2_ms,
                "score_ms": q2ph.get("score_topk"),
                "note": "k = the candidate limit of the reference's top-10 call (search_service.py:236-239)"}
        if not args.no_variants:
            q3_ms, q3ph = run_queries(args.k, max(3, args.query_steps // 4), with_rank=True)
            out["query"]["variant_with_url_ranks"] = {"ms_per_batch": q3_ms, "score_ms": q3ph.get("score_topk"),
                                                      "note": "explicit rank array (HAS_RANK = true kernels): one gather per candidate that "
                                                              "reaches the k-th best score"}
            if args.k2 > 0:
                q4_ms, q4ph = run_queries(args.k2, max(3, args.query_steps // 4), with_rank=True)
                out["query"]["variant_with_url_ranks"]["candidate_limit_run"] = {"k": args.k2, "ms_per_batch": q4_ms,
                                                                                "score_ms": q4ph.get("score_topk")}
                if world == 1:
                    q5_ms, q5ph = run_queries(args.k2, max(3, args.query_steps // 4), appended_keys=True)
                    out["query"]["variant_with_url_ranks"]["keys_for_appended_docs_only"] = {
                        "k": args.k2, "ms_per_batch": q5_ms, "score_ms": q5ph.get("score_topk"),
                        "note": "1 % of the docs appended since the last rebuild: (unit + 1) << 8 for the others, no array read"}
        if world == 1 and q_ix is not ix:
            q_ix.close()

    if rank == 0 and world == 1 and not args.no_cpu and args.articles == 0:
        # rows of the first cpu_docs documents: a prefix of the unit-major corpus
        cpu_docs = min(args.cpu_docs, docs)
        tmp_ix = se.Index(local_rank)
        cc = make_corpus(tmp_ix, cpu_docs, 0, 0, 0)
        host = [cc[key].cpu().numpy().view(dt) for key, dt in
                (("term", np.uint32), ("unit", np.uint32), ("tf", np.uint32), ("meta", np.uint8), ("wc", np.uint32))]
        tmp_ix.close()
        del cc
        out["cpu_baseline"] = cpu_baseline(args, host + [cpu_docs])

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ix.close()


if __name__ == "__main__":
    main()
