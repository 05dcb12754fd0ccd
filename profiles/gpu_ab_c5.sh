#!/bin/bash
# Same-box A/B of config 5 (generation 6) over library builds / environment knobs and batch sizes.
# usage: bash profiles/gpu_ab_c5.sh <tag> "<S list>" label=lib[,ENV=VAL...] ...      (lib relative to open-simulator_amd/csrc or profiles/ab)
set -u
TAG=$1; SLIST=$2; shift 2
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
for S in $SLIST; do
  for SPEC in "$@"; do
    LABEL=${SPEC%%=*}; REST=${SPEC#*=}
    LIB=${REST%%,*}; ENVS=""
    if [ "$REST" != "$LIB" ]; then ENVS=$(echo "${REST#*,}" | tr ',' ' '); fi
    P=$ROOT/open-simulator_amd/csrc/$LIB; [ -f "$P" ] || P=$ROOT/profiles/ab/$LIB
    env $ENVS SIMON_HIP_LIB=$P SIMON_BENCH_C5_SCEN=$S timeout 300 python bench.py --workload config5 --steps 2 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$LABEL', 'S', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'scen/s', round(d['config']['scenarios_per_gpu']/d['roofline']['kernel_ms']*1e3), 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
  done
done 2>&1 | tee "$OUT/ab.txt"
echo "total $(( $(date +%s) - t0 )) s"
