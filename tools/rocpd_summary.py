#!/usr/bin/env python3
"""Summaries of the rocprofv3 runs made by tools/collect_profiles.sh (rocpd sqlite output).

    python tools/rocpd_summary.py gpurun_out/prof_r1 profiles r1

writes <out>/<tag>_bench_kernel_stats.csv (per-kernel launch statistics of the default bench
command) and <out>/<tag>_pmc_traffic.json (memory-side traffic per launch of each build kernel).

Traffic: FETCH_SIZE and WRITE_SIZE (KiB, separate passes) as rocprofv3 derives them, next to the
raw request counters they come from.  On gfx950 a 128-byte read request is tallied by FETCH_SIZE as
64 bytes, so read bytes are recomputed from the request counters (32 / 64 / 128-byte requests) and
checked against a kernel whose read volume is known exactly (k_sort_hist reads 4 or 8 bytes per
row and nothing else); writes are 64-byte requests unless counted as partial (32 bytes)."""
import collections, csv, json, re, sqlite3, sys, glob, os

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("sebm25::", "").replace("void ", "")
    m = re.match(r"(k_[a-z_0-9]+(<.*>)?)\(", name)
    if m:
        return m.group(1)
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def db(path):
    return sqlite3.connect(glob.glob(os.path.join(src, path, "*.db"))[0])


# ---- launch statistics
c = db("stats")
# one line per (kernel, grid size): the same kernels run on the 1M-doc corpus of the build leg and once on the
# 10M-doc corpus of the query leg
rows = c.execute("select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                 "group by name, grid_x order by sum(duration) desc").fetchall()
total = sum(r[3] for r in rows)
with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "GridX", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, grid, n, tot, avg, mn, mx in rows:
        w.writerow([name, grid, n, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])
print("kernel stats:", len(rows), "(kernel, grid) pairs")
for name, grid, n, tot, avg, mn, mx in rows[:14]:
    print(f"  {short(name):44s} grid={grid:10d} calls={n:4d} avg={avg / 1e6:8.4f} ms  {100.0 * tot / total:5.1f} %")


# ---- counters per kernel (average per launch)
def counters(path):
    """average per launch, over the launches with the kernel's most frequent grid size (the build leg's)"""
    c = db(path)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for name, grid, cn, v in c.execute("select kernel_name, grid_size, counter_name, value from counters_collection"):
        agg[short(name)][grid][cn].append(v)
    res = {}
    for k, by_grid in agg.items():
        grid = max(by_grid, key=lambda g: max(len(x) for x in by_grid[g].values()))
        res[k] = {cn: sum(x) / len(x) for cn, x in by_grid[grid].items()}
    return res


fetch, write, rd, wr = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_rd"), counters("pmc_wr")
bench = json.loads([l for l in open(os.path.join(src, "pmc_fetch.json")) if l.startswith("{")][-1])
R = bench["config"]["rows_per_gpu"]
SRC = ("rocprofv3 --pmc, one pass per counter group, bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-update "
       "--query-steps 2 --k2 0; averages per launch")


def traffic(k):
    n = rd[k].get("TCC_EA0_RDREQ_sum", 0.0)
    n32 = rd[k].get("TCC_EA0_RDREQ_32B_sum", 0.0)
    n128 = rd[k].get("TCC_EA0_RDREQ_128B_sum", 0.0)
    rbytes = 32 * n32 + 128 * n128 + 64 * (n - n32 - n128)
    m = wr[k].get("TCC_EA0_WRREQ_sum", 0.0)
    m64 = wr[k].get("TCC_EA0_WRREQ_64B_sum", 0.0)
    wbytes = 64 * m64 + 32 * (m - m64)
    return {
        "FETCH_SIZE_KiB": fetch[k].get("FETCH_SIZE"), "WRITE_SIZE_KiB": write[k].get("WRITE_SIZE"),
        "rd_requests": n, "rd_requests_32B": n32, "rd_requests_128B": n128,
        "wr_requests": m, "wr_requests_64B": m64,
        "read_bytes": rbytes, "write_bytes": wbytes, "hbm_bytes_per_launch": rbytes + wbytes,
    }


common = sorted(set(fetch) & set(write) & set(rd) & set(wr))
build = {"rows": R, "workload": bench["config"]["workload"], "source": SRC, "kernels": {}}
for k in common:
    if k.startswith("k_") and not k.startswith(("k_score", "k_merge", "k_gen")):
        build["kernels"][k] = traffic(k)
for k, bpr in (("k_sort_hist<0>", 4), ("k_sort_hist<2>", 8)):
    if k in build["kernels"]:
        d = build["kernels"][k]
        d["known_read_bytes"] = bpr * R
        d["read_bytes_over_known"] = d["read_bytes"] / (bpr * R)
        d["FETCH_SIZE_over_known"] = d["FETCH_SIZE_KiB"] * 1024 / (bpr * R)
build["bytes_per_row_all_kernels"] = sum(d["hbm_bytes_per_launch"] for d in build["kernels"].values()) / R
res = {"build": build}
q = bench.get("query")
if q:
    # the scoring kernel of the batch: the k_score_step instantiation with the largest traffic
    ks = [k for k in common if k.startswith("k_score")]
    if ks:
        # the counters average over every launch of a name; the bench launches one instantiation per (list count, k)
        best = max(ks, key=lambda k: traffic(k)["hbm_bytes_per_launch"])
        t = traffic(best)
        res["query"] = {"kernel": best, "postings_scored_per_batch": q["postings_scored_per_batch_per_gpu"],
                        "alg_bytes": q["roofline"]["alg_bytes"], "workload": q["workload"], "source": SRC, **t,
                        "traffic_over_alg_bytes": t["hbm_bytes_per_launch"] / q["roofline"]["alg_bytes"]}
with open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(res, f, indent=1)
for k, d in build["kernels"].items():
    print(f"  {k:26s} read {d['read_bytes'] / 1e9:7.3f} GB (FETCH_SIZE {d['FETCH_SIZE_KiB'] * 1024 / 1e9:7.3f})  "
          f"write {d['write_bytes'] / 1e9:7.3f} GB (WRITE_SIZE {d['WRITE_SIZE_KiB'] * 1024 / 1e9:7.3f})  "
          f"per row {d['hbm_bytes_per_launch'] / R:6.2f} B" + (f"  read/known {d['read_bytes_over_known']:.3f}" if "known_read_bytes" in d else ""))
print("  build, all kernels: %.2f B/row" % build["bytes_per_row_all_kernels"])
if "query" in res:
    d = res["query"]
    print(f"  {d['kernel']}: {d['hbm_bytes_per_launch'] / 1e9:.3f} GB per batch against {d['alg_bytes'] / 1e9:.3f} GB algorithmic "
          f"(x{d['traffic_over_alg_bytes']:.3f})")
