#!/bin/bash
# parity at scale: the three fuzzers of the score-table kernel, many cases; usage: bash profiles/gpu_r3q.sh <tag>
set -u
TAG=${1:-r3q}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python tests/fuzz_spread.py 1500 10000 2>&1 | tail -8 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 1200 python tests/fuzz_rest.py 800 20000 2>&1 | tail -8 ) > "$OUT/fuzz_rest.log"; tail -2 "$OUT/fuzz_rest.log"; echo "$(( $(date +%s) - t0 )) s"
( timeout 1200 python tests/fuzz_table.py 500 30000 2>&1 | tail -8 ) > "$OUT/fuzz_table.log"; tail -2 "$OUT/fuzz_table.log"
echo "total $(( $(date +%s) - t0 )) s"
