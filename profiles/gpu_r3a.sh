#!/bin/bash
# Round 3, first GPU visit: the REST select rewritten summary-first (row-major filter rows), parity + same-box A/B against the
# round-2 library (profiles/ab/libsimon_r02.so, built from commit a7786fb).
# usage (through gpurun): bash profiles/gpu_r3a.sh <tag>
set -u
TAG=${1:-r3a}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -k "rest or config5 or gpushare or explain or 64_internal or fixtures or golden" 2>&1 | tail -15 ) > "$OUT/pytest_rest.log"; tail -3 "$OUT/pytest_rest.log"
( timeout 600 python tests/fuzz_rest.py 400 7000 2>&1 | tail -5 ) > "$OUT/fuzz_rest.log"; tail -2 "$OUT/fuzz_rest.log"
echo "tests done $(( $(date +%s) - t0 )) s"
{
for S in 256 1024 2048; do
  for LIB in profiles/ab/libsimon_r02.so open-simulator_amd/csrc/libsimon_hip.so; do
    SIMON_HIP_LIB=$ROOT/$LIB SIMON_BENCH_C5_SCEN=$S timeout 300 python bench.py --workload config5 --steps 2 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib', '$LIB'.split('/')[-1], 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'), 'gen', d['config']['kernel_generation'])"
  done
done
} > "$OUT/ab_config5.txt" 2>&1; cat "$OUT/ab_config5.txt"
echo "total $(( $(date +%s) - t0 )) s"
