#!/bin/bash
# Service workload where 20 services also carry a hard zone constraint (maxSkew 2): generation 7 vs the all-feature kernel; usage: bash profiles/gpu_r3at.sh <tag>
set -u
TAG=${1:-r3at}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
timeout 900 python bench.py --workload service --hard 20 --counts 64 --steps 2 --warmup 1 --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + hard zone constraints (20), with the oracle sample: scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'parity', d['parity_sample']['scenarios'], d['parity_sample']['mismatches'])"
for V in table all_feature; do
  E="X=1"; [ $V = all_feature ] && E="SIMON_NO_HARD_FOLD=1"
  env $E timeout 900 python bench.py --workload service --hard 20 --counts 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + hard (20)', '$V', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'kernel', d['config']['kernel'], 'gen', d['config']['kernel_generation'])"
done
} | tee "$OUT/service_hard_ab.txt"
