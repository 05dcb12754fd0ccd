#!/bin/bash
# all-feature kernel with stage-A caches (InterPodAffinity raw scores, soft-spread counts in LDS): every GPU test, the general-purpose fuzzer, e2e A/B; usage: bash profiles/gpu_r3y.sh <tag>
set -u
TAG=${1:-r3y}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
( timeout 900 python tests/fuzz_gpu.py 200 5000 2>&1 | tail -4 ) > "$OUT/fuzz_gpu.log"; tail -2 "$OUT/fuzz_gpu.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for V in old product nocache; do
  case $V in
    old) E="SIMON_HIP_LIB=$PWD/profiles/ab/libsimon_r3w0.so";;
    product) E="X=1";;
    nocache) E="SIMON_WIDE_NO_CACHE_B=1";;
  esac
  env $E timeout 600 python profiles/e2e_sweep.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', 'kernel_ms', d['kernel_ms'], 'host_s', d['host_s'], 'unscheduled', d['unscheduled_first_last'])"
done
} | tee "$OUT/e2e_ab.txt"
echo "total $(( $(date +%s) - t0 )) s"
