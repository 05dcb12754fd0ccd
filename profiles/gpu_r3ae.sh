#!/bin/bash
# Service workload where 20 of the 60 services also require anti-affinity to their own pods on the hostname key: generation 7 + the fold vs the all-feature kernel; usage: bash profiles/gpu_r3ae.sh <tag>
set -u
TAG=${1:-r3ae}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
timeout 900 python bench.py --workload service --anti 20 --counts 64 --steps 2 --warmup 1 --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + anti(20), with the oracle sample: scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'parity', d['parity_sample']['scenarios'], d['parity_sample']['mismatches'], 'cpu', d['cpu_baseline']['value'])"
for CNT in 64 1024; do
  for V in fold no_fold; do
    E="X=1"; [ $V = no_fold ] && E="SIMON_NO_FOLD=1"
    env $E timeout 900 python bench.py --workload service --anti 20 --counts $CNT --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + anti(20)', '$V', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'kernel', d['config']['kernel'], 'gen', d['config']['kernel_generation'])"
  done
done
} | tee "$OUT/service_anti_ab.txt"
