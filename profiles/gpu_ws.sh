#!/bin/bash
# does the score-table kernel's time at 16 waves per CU depend on the batch's HBM working set?  Same number of scenarios and
# scheduling cycles, different scenario sizes (node counts 488.. + counts): bash profiles/gpu_ws.sh <tag>
set -u
TAG=${1:-ws}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', r['kernel_ms'], 'lds', r.get('lds_bytes_per_workgroup'))"; }
for rep in 1 2; do
run "--counts 1024 --orders-per-gpu 4"
run "--counts 256 --orders-per-gpu 16"
run "--counts 64 --orders-per-gpu 64"
run "--counts 16 --orders-per-gpu 256"
run "--counts 1024 --orders-per-gpu 1"
run "--counts 64 --orders-per-gpu 16"
done
