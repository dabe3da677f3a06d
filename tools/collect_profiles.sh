#!/bin/bash
# Run on the GPU box from the repository root: the per-kernel trace of the default bench command
# and the HBM-traffic counter passes (one rocprofv3 run per counter group).  Results land under
# gpurun_out/prof_r1/; tools/pmc_traffic.py turns the counter CSVs into profiles/r1_pmc_traffic.json.
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r1
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o r1 -- python3 "$REPO/bench.py" > "$OUT/bench_under_trace.json" 2> "$OUT/stats.log"
SHORT="--steps 3 --warmup 1 --no-cpu --no-query"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o r1 -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.log"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o r1 -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.log"
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d "$OUT/pmc_rd" -o r1 -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_rd.json" 2> "$OUT/pmc_rd.log"
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d "$OUT/pmc_wr" -o r1 -- python3 "$REPO/bench.py" $SHORT > "$OUT/pmc_wr.json" 2> "$OUT/pmc_wr.log"
cd "$REPO"
find "$OUT" -name "*.csv" | head -40
