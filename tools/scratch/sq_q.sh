#!/bin/bash
# scratch: shader-core counters of the query batch (tools/scratch/qexp.py), two counter passes
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/sqq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
export KS=${KS:-10} DOCS=${DOCS:-10000000}   # RANK=1 in the environment adds an explicit url-rank array
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d "$OUT/a" -o sq -- python3 "$REPO/tools/scratch/qexp.py" > "$OUT/a.json" 2> "$OUT/a.log"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT -d "$OUT/b" -o sq -- python3 "$REPO/tools/scratch/qexp.py" > "$OUT/b.json" 2> "$OUT/b.log"
cd "$REPO"
python3 tools/sq_counters.py "$OUT/a" "$OUT/b"
