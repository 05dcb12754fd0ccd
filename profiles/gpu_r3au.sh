#!/bin/bash
# last parity sweep of the round: fuzz_spread with every new feature among its draws + the general fuzzer; usage: bash profiles/gpu_r3au.sh <tag>
set -u
TAG=${1:-r3au}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python tests/fuzz_spread.py 2500 300000 2>&1 | tail -8 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 900 python tests/fuzz_gpu.py 400 310000 2>&1 | tail -4 ) > "$OUT/fuzz_gpu.log"; tail -1 "$OUT/fuzz_gpu.log"
( timeout 900 python tests/fuzz_table.py 300 320000 2>&1 | tail -4 ) > "$OUT/fuzz_table.log"; tail -1 "$OUT/fuzz_table.log"
echo "total $(( $(date +%s) - t0 )) s"
