#!/bin/bash
# GPU visit: parity tests + config-5 bench at the default workgroup choice for two batch sizes.
set -u
TAG=${1:-quick2}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
for S in ${SCENS:-256 1024}; do
SIMON_BENCH_C5_SCEN=$S timeout 400 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/c5_s$S.json" 2> "$OUT/c5_s$S.err"
python - "$OUT/c5_s$S.json" $S <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("config5 S", sys.argv[2], "wg", d["config"]["workgroup"], "ms", d["roofline"]["kernel_ms"], "scen/s", round(d["value"],1))
except Exception as e: print("failed", e)
PY
done
