#!/bin/bash
# all-feature kernel: which stage-A cache pays; usage: bash profiles/gpu_r3z.sh <tag>
set -u
TAG=${1:-r3z}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
for V in old both ipa_only pts_only none; do
  case $V in
    old) E="SIMON_HIP_LIB=$PWD/profiles/ab/libsimon_r3w0.so";;
    both) E="X=1";;
    ipa_only) E="SIMON_WIDE_NO_CACHE_B=2";;
    pts_only) E="SIMON_WIDE_NO_CACHE_B=1";;
    none) E="SIMON_WIDE_NO_CACHE_B=3";;
  esac
  env $E timeout 600 python profiles/e2e_sweep.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['kernel_ms'], 'unscheduled', d['unscheduled_first_last'])"
done
} | tee "$OUT/e2e_cache_ab.txt"
