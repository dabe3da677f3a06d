#!/bin/bash
# Run on the GPU box from the repository root: build-only bench phases for each experiment library named.
# Usage: exp_phases.sh <name> ...   ("base" = the regular library)
for n in "$@"; do
  if [ "$n" = base ]; then unset SEBM25_LIB; else export SEBM25_LIB=$PWD/tools/scratch/lib/libsebm25_$n.so; fi
  timeout -k 10 200 python bench.py --no-cpu --no-query --no-update --no-variants --steps 10 --warmup 3 > gpurun_out/exp_$n.json 2> gpurun_out/exp_$n.err || { echo "$n FAILED"; tail -5 gpurun_out/exp_$n.err; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.load(open(f"gpurun_out/exp_{n}.json"))
print(n, "ms_per_step=%.3f"%d["ms_per_step"], " ".join(f"{k}={v:.3f}" for k,v in d["roofline"]["phases_ms"].items()))
PY
done
