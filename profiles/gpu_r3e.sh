#!/bin/bash
# full GPU suite + the default bench line (timed); usage: bash profiles/gpu_r3e.sh <tag>
set -u
TAG=${1:-r3e}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
echo "pytest $(( $(date +%s) - t0 )) s"
t1=$(date +%s)
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? $(( $(date +%s) - t1 )) s"
tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['kernel_ms'], 'parity', d['parity_sample']['mismatches'], d['parity_sample']['scenarios'])
print('e2e', d.get('end_to_end'))
for w in d.get('other_workloads', []):
    r=w.get('roofline',{})
    print(w.get('workload'), w.get('value'), w.get('kernel_ms'), 'roof', r.get('frac'), r.get('measured_hbm_frac'), 'cpu', w.get('cpu_baseline',{}).get('value'), 'par', w.get('parity_sample',{}).get('mismatches'), w.get('parity_sample',{}).get('scenarios'), w.get('error'))
PY
echo "total $(( $(date +%s) - t0 )) s"
