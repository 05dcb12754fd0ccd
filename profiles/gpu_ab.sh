#!/bin/bash
# A/B on ONE box (boxes differ by up to 20 %): config-5 bench alternating between two environments.
# usage: bash profiles/gpu_ab.sh <tag> "<env A>" "<env B>" [scenario counts...]
set -u
TAG=$1; ENVA=$2; ENVB=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for S in "${@:-256}"; do
for rep in 1 2 3; do
for V in A B; do
  if [ $V = A ]; then E=$ENVA; else E=$ENVB; fi
  env $E SIMON_BENCH_C5_SCEN=$S timeout 400 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$V.$S.$rep.json" 2> "$OUT/$V.$S.$rep.err"
  python - "$OUT/$V.$S.$rep.json" $V $S "$E" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "S", sys.argv[3], "wg", d["config"]["workgroup"], "ms", d["roofline"]["kernel_ms"], "|", sys.argv[4])
except Exception as e: print("failed", e)
PY
done; done; done
