#!/bin/bash
# Run on the GPU box from the repository root: the per-kernel trace of the default bench command
# and the HBM-traffic counter passes (one rocprofv3 run per counter group, build and query legs).
# Results land under gpurun_out/prof_$TAG/; tools/rocpd_summary.py turns them into
# profiles/${TAG}_bench_kernel_stats.csv and profiles/${TAG}_pmc_traffic.json.
set -e
TAG=${1:-r2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o $TAG -- python3 "$REPO/bench.py" > "$OUT/bench_under_trace.json" 2> "$OUT/stats.log"
SHORT="--steps 3 --warmup 1 --no-cpu --no-variants --no-update --query-steps 2 --k2 0"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o $TAG -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o $TAG -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.log"
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d "$OUT/pmc_rd" -o $TAG -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_rd.json" 2> "$OUT/pmc_rd.log"
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d "$OUT/pmc_wr" -o $TAG -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_wr.json" 2> "$OUT/pmc_wr.log"
cd "$REPO"
python3 tools/rocpd_summary.py "$OUT" "$OUT" $TAG
cp "$OUT/bench_under_trace.json" "$OUT/${TAG}_bench_under_trace.json"
ls -la "$OUT"/*.csv "$OUT"/*.json
