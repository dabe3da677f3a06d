#!/usr/bin/env python3
"""Benchmark of the BM25 index-build path (and the batched top-k scoring path) on MI355X.

One "step" = one full statistics job over this rank's shard of the synthetic Zipfian corpus,
rows already resident in HBM: radix sort -> segmented reduction into the CSR index -> df /
document totals -> (N > 1: all-reduce of df[V] and the document totals over RCCL) -> idf.
Workload at N = 1: benchmark configuration 2 (1M docs, 500k-term vocabulary, ~300 draws per doc);
at N > 1 every rank builds its own 1M-doc shard (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--docs D] [--queries Q] ...

Rank 0 prints one JSON line.  `value` = token rows of all ranks / max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000, help="web docs per GPU")
    ap.add_argument("--articles", type=int, default=0, help="news articles per GPU")
    ap.add_argument("--vocab", type=int, default=500_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--terms-per-query", type=int, default=3)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--k2", type=int, default=220, help="second scoring run at this k (0 = skip)")
    ap.add_argument("--query-steps", type=int, default=5)
    ap.add_argument("--cpu-docs", type=int, default=1_000_000, help="docs in the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-query", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, rows):
    """The CPU oracle (a single-threaded C restatement of the reference's SQL) timed on the first
    `cpu_docs` documents of the same synthetic corpus (the rows come from the GPU generator, which
    the tests hold bit-identical to the oracle's own).  Checker code, never the product path."""
    import numpy as np

    from oracle import oracle as O

    term, unit, tf, meta, wc, n_docs = rows
    n = len(term)
    t0 = time.perf_counter()
    orc = O.OracleIndex(term, unit, tf, meta, args.vocab, n_docs, 0, wc)
    t_build = time.perf_counter() - t0
    nq = 1000
    qs = O.gen_queries(20260228, args.vocab, O.zipf_cdf(args.vocab), 100, nq, args.terms_per_query)
    t0 = time.perf_counter()
    for q in qs:
        orc.search(q, args.k, level="rows")
    t_q = time.perf_counter() - t0
    orc.close()
    return {
        "value": n / t_build, "unit": "token rows/s", "cores": 1, "kind": "port",
        "sample": f"{n_docs} docs / {n} rows of the same synthetic corpus, single-threaded C oracle "
                  f"(sort + collapse + df + idf), {t_build:.2f} s",
        "queries_per_s": nq / t_q,
        "query_sample": f"{nq} queries x {args.terms_per_query} terms, top-{args.k}, over the {n_docs}-doc sample, {t_q:.2f} s",
    }


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import search_engine_amd as se

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
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

    ix = se.Index(local_rank)
    ix.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- this rank's shard of the corpus, generated in HBM
    spec = se.make_spec(n_terms=args.vocab, n_docs=args.docs, n_articles=args.articles,
                        doc_begin=rank * args.docs, article_begin=rank * args.articles)
    n_rows = ix.gen_corpus_rows(spec)
    term = torch.empty(n_rows, dtype=torch.int32, device=dev)
    unit = torch.empty_like(term)
    tf = torch.empty_like(term)
    meta = torch.empty(n_rows, dtype=torch.uint8, device=dev)
    wc = torch.empty(args.docs, dtype=torch.int32, device=dev)
    ix.gen_corpus(spec, term, unit, tf, meta, wc)
    ix.set_unit_base(rank * args.docs, rank * args.articles)

    from search_engine_amd.sharded import ShardedIndex

    sharded = ShardedIndex(ix, dev) if world > 1 else None

    def step():
        if world == 1:
            ix.build(term, unit, tf, meta, args.vocab, args.docs, args.articles, wc)
        else:
            # build_local -> all-reduce of df[V] and the document totals (RCCL over xGMI) -> idf
            sharded.build(term, unit, tf, meta, args.vocab, args.docs, args.articles, wc,
                          doc_base=rank * args.docs, article_base=rank * args.articles)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ix.enable_timing(True)
    sync()
    t0 = time.perf_counter()
    phase_sum: dict[str, float] = {}
    for _ in range(args.steps):
        step()
        for name, ms in ix.timings():  # HIP events around each kernel of this step, on the library's stream
            phase_sum[name] = phase_sum.get(name, 0.0) + ms
    sync()
    elapsed = time.perf_counter() - t0
    phases = [(n, ms / args.steps) for n, ms in phase_sum.items()]  # average launch duration per kernel
    ix.enable_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        rows_all = torch.tensor([n_rows], dtype=torch.int64, device=dev)
        dist.all_reduce(rows_all)
        total_rows = int(rows_all.item())
    else:
        total_rows = n_rows

    info = ix.info()
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rows * args.steps / elapsed

    # ---- roofline of the dominant kernel (per GPU, per launch), HIP events on the library's stream
    # algorithmic bytes per input row of each build kernel (what it must read + what it must write)
    KERNELS = {
        "sort_hist_0": ("k_sort_hist<0>", 4), "sort_hist_1": ("k_sort_hist<2>", 8),
        "sort_scatter_0": ("k_sort_scatter_tile<0>", 13 + 8),   # raw row in, packed row out
        "sort_scatter_1": ("k_sort_scatter_tile<2>", 8 + 8),    # packed row in, packed row out
        "postings": ("k_postings", 8 + 8),                 # packed row in, posting slot out
    }
    dev_ms = sum(ms for _, ms in phases)
    dom_name, dom_ms = max(phases, key=lambda p: p[1]) if phases else ("", 0.0)
    kname, bpr = KERNELS.get(dom_name, (dom_name, 0))
    dom_bytes = bpr * info.n_rows
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
    # HBM traffic of that kernel from rocprofv3 PMC passes (collected separately, see profiles/)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("rows") == int(info.n_rows) and kname in pmc.get("kernels", {}):
            traffic = pmc["kernels"][kname]["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    # the whole build against the build's algorithmic bytes: 13 B/row in, 8 B/posting out, 4 B/doc, 20 B/term
    alg_bytes = 13 * info.n_rows + 8 * info.n_postings + 4 * info.n_docs + 20 * info.n_terms
    pipe = alg_bytes / (ms_per_step * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "kernel": kname, "kernel_ms": dom_ms, "alg_bytes_per_row": bpr, "alg_bytes_per_launch": dom_bytes,
        "pipeline": {"alg_bytes": alg_bytes, "achieved": pipe, "frac": pipe / HBM_PEAK_GBS,
                     "scope": "whole build of one GPU: 13 B/row + 8 B/posting + 4 B/doc + 20 B/term over the wall "
                              "time of one step", "device_ms_sum_of_phases": dev_ms},
        "phases_ms": {n: round(ms, 4) for n, ms in phases},
    }

    out = {
        "metric": "bm25_index_build_token_rows_per_s", "value": value, "unit": "token rows/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"BM25 index build, {args.docs} docs per GPU, vocab {args.vocab}, Zipf s=1, "
                               f"~300 body draws/doc (benchmark configuration 2 per GPU)",
                   "rows_per_gpu": int(info.n_rows), "postings_per_gpu": int(info.n_postings),
                   "sort_passes": int(info.sort_passes), "parallelism": f"doc-sharded x{world}"},
        "roofline": roofline,
        "arithmetic": "u64 packed rows, u32 counts and posting weights (exact tenths), f64 idf and scores",
    }

    # ---- batched scoring on the index just built
    if not args.no_query:
        qs = se.gen_queries(20260228, args.vocab, args.queries, args.terms_per_query, 100)
        res = {k: torch.empty((args.queries, args.k), dtype=dt, device=dev) for k, dt in
               (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
        res["count"] = torch.empty(args.queries, dtype=torch.int32, device=dev)
        ix.search(qs, args.k, out=res)
        ix.enable_timing(True)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.query_steps):
            ix.search(qs, args.k, out=res)
        sync()
        tq = time.perf_counter() - t0
        qphases = ix.timings()
        ix.enable_timing(False)
        if world > 1:
            tmax = torch.tensor([tq], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tq = float(tmax.item())
        # postings read by the batch: sum over query terms of the web list length (= local df)
        df_local, _ = ix.local_stats()
        sum_df = int(df_local.astype(np.int64)[qs.reshape(-1)].sum())
        alg_q = 8 * sum_df + args.queries * (13 * args.k + 48)
        q_ms = tq / args.query_steps * 1e3
        score_ms = dict(qphases).get("score_topk", 0.0)
        out["query"] = {
            "metric": f"top{args.k}_queries_per_s", "value": args.queries * args.query_steps / tq, "unit": "queries/s",
            "ms_per_batch": q_ms, "queries": args.queries, "terms_per_query": args.terms_per_query, "k": args.k,
            "note": "every rank scores the whole batch over its own shard" if world > 1 else "one GPU, whole index",
            "postings_scored_per_batch": sum_df,
            "roofline": {"bound": "hbm", "achieved": alg_q / (score_ms * 1e-3) / 1e9 if score_ms else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_q / (score_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if score_ms else None,
                         "traffic": None, "kernel": "k_score", "kernel_ms": score_ms, "alg_bytes": alg_q,
                         "phases_ms": {n: round(ms, 4) for n, ms in qphases}},
        }

    # ---- the same batch at the reference's own candidate limit (limit=20 -> k=220, search_service.py:236-239)
    if not args.no_query and args.k2 > 0:
        res2 = {k: torch.empty((args.queries, args.k2), dtype=dt, device=dev) for k, dt in
                (("unit", torch.int32), ("score", torch.float64), ("matched", torch.int32), ("tie", torch.int32))}
        res2["count"] = torch.empty(args.queries, dtype=torch.int32, device=dev)
        ix.search(qs, args.k2, out=res2)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.query_steps):
            ix.search(qs, args.k2, out=res2)
        sync()
        tq2 = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([tq2], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tq2 = float(tmax.item())
        out["query"]["candidate_limit_run"] = {"k": args.k2, "value": args.queries * args.query_steps / tq2,
                                               "unit": "queries/s", "ms_per_batch": tq2 / args.query_steps * 1e3}

    if rank == 0 and world == 1 and not args.no_cpu:
        # rows of the first cpu_docs documents: a prefix of the unit-major corpus
        cpu_docs = min(args.cpu_docs, args.docs)
        n_cpu = ix.gen_corpus_rows(se.make_spec(n_terms=args.vocab, n_docs=cpu_docs)) if args.articles == 0 else 0
        if n_cpu:
            host = [x[:n_cpu].cpu().numpy().view(dt) for x, dt in
                    ((term, np.uint32), (unit, np.uint32), (tf, np.uint32), (meta, np.uint8))]
            host.append(wc[:cpu_docs].cpu().numpy().view(np.uint32))
            out["cpu_baseline"] = cpu_baseline(args, host + [cpu_docs])

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ix.close()


if __name__ == "__main__":
    main()
