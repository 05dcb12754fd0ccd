#!/bin/bash
# the very last check of the round on the final tree: every GPU test, smoke, the default bench line; usage: bash profiles/gpu_r3_last.sh <tag>
set -u
TAG=${1:-r03f}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > "$OUT/pytest_gpu.log"; tail -2 "$OUT/pytest_gpu.log"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open("$OUT/bench_default.json".replace("$OUT","'$OUT'".strip("'"))).read().strip().splitlines()[-1]) if False else None
PY
python -c "
import json,sys
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'parity', d['parity_sample']['mismatches'])
for o in d['other_workloads']: print(o['workload'], o.get('ms_per_step'), o['parity_sample']['mismatches'])
"
