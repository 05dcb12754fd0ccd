#!/bin/bash
# parity at scale after the fold and the preferred-affinity table: the four fuzzers, many cases; usage: bash profiles/gpu_r3an.sh <tag>
set -u
TAG=${1:-r3an}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python tests/fuzz_spread.py 2500 200000 2>&1 | tail -8 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 1200 python tests/fuzz_table.py 800 210000 2>&1 | tail -8 ) > "$OUT/fuzz_table.log"; tail -2 "$OUT/fuzz_table.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 1200 python tests/fuzz_rest.py 600 220000 2>&1 | tail -8 ) > "$OUT/fuzz_rest.log"; tail -2 "$OUT/fuzz_rest.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 1200 python tests/fuzz_gpu.py 500 230000 2>&1 | tail -8 ) > "$OUT/fuzz_gpu.log"; tail -2 "$OUT/fuzz_gpu.log"
echo "total $(( $(date +%s) - t0 )) s"
