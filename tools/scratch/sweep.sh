set -e
for c in 512 2048 8192 32768 65536; do
  echo "chunks=$c"
  SEBM25_SORT_CHUNKS=$c timeout -k 10 200 python -u bench.py --no-cpu --no-query --steps 5 > gpurun_out/b_$c.log 2>&1
  python - <<EOF2
import json
d=json.loads([l for l in open("gpurun_out/b_$c.log") if l.startswith("{")][-1]); print(d["ms_per_step"], d["roofline"]["phases_ms"])
EOF2
done
