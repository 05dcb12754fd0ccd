#!/bin/bash
# Service workload where every service also PREFERS not to sit next to its own pods (hostname 100, zone 50): generation 7 vs the all-feature kernel; usage: bash profiles/gpu_r3ai.sh <tag>
set -u
TAG=${1:-r3ai}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
timeout 900 python bench.py --workload service --pref 60 --counts 64 --steps 2 --warmup 1 --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + preferred self anti-affinity (60), with the oracle sample: scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'parity', d['parity_sample']['scenarios'], d['parity_sample']['mismatches'])"
for CNT in 64 1024; do
  for V in table all_feature; do
    E="X=1"; [ $V = all_feature ] && E="SIMON_NO_IPA_FOLD=1"
    env $E timeout 900 python bench.py --workload service --pref 60 --counts $CNT --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service + preferred (60)', '$V', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'kernel', d['config']['kernel'], 'gen', d['config']['kernel_generation'])"
  done
done
} | tee "$OUT/service_pref_ab.txt"
