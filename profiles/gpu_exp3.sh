#!/bin/bash
# Wide-kernel experiment: phase profile split by pod kind + throughput against the batch size (config 5).
set -u
TAG=${1:-exp3}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export SIMON_WG=${WG:-512}
SIMON_WIDE_PROF=1 timeout 300 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/prof.json" 2> "$OUT/prof.err"
grep SIMON_WIDE_PROF "$OUT/prof.err"
for S in ${SCENS:-256 512 1024}; do
SIMON_BENCH_C5_SCEN=$S timeout 400 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/s$S.json" 2> "$OUT/s$S.err"
python - "$OUT/s$S.json" $S <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("S", sys.argv[2], "ms", d["roofline"]["kernel_ms"], "scen/s", round(d["value"],1))
except Exception as e: print("failed", e)
PY
done
