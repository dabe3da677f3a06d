#!/bin/bash
# Run on the GPU box from the repository root: shader-core counters of the build kernels only (no update or
# query leg, so every launch is a full-size one).  Prints per-kernel sums per launch.
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/sqb
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
SHORT="--steps 3 --warmup 1 --no-cpu --no-query --no-update --no-variants"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d "$OUT/a" -o sq -- python3 "$REPO/bench.py" $SHORT > "$OUT/a.json" 2> "$OUT/a.log"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT -d "$OUT/b" -o sq -- python3 "$REPO/bench.py" $SHORT > "$OUT/b.json" 2> "$OUT/b.log"
cd "$REPO"
python3 tools/sq_counters.py "$OUT/a" "$OUT/b" > gpurun_out/sq_build.txt
rm -rf "$OUT"
grep -A17 "k_sort_scatter_tile<${1:-2, true}>" gpurun_out/sq_build.txt
