#!/bin/bash
# Quick GPU visit: parity tests (-x) + config-5 bench (+ optional default bench with "full").
set -u
TAG=${1:-quick}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_EXTRA:-} 2>&1 | tail -40 ) > "$OUT/pytest.log"
tail -8 "$OUT/pytest.log"
for wg in ${WGS:-1024}; do
SIMON_WG=$wg timeout 420 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_config5_wg$wg.json" 2> "$OUT/bench_config5.err"; python - "$OUT/bench_config5_wg$wg.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("config5", d["config"]["workgroup"], "ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "plan", d["config"]["plan"])
except Exception as e: print("config5 failed", e)
PY
done
tail -3 "$OUT/bench_config5.err"
if [ "${2:-}" = "full" ]; then
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 600 "$OUT/bench_default.json"
fi
