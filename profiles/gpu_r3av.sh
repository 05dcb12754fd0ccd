#!/bin/bash
# the general path end to end on a TYPICAL cluster (profiles/e2e_sweep.py --typical): score-table kernel vs the all-feature kernel; usage: bash profiles/gpu_r3av.sh <tag>
set -u
TAG=${1:-r3av}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "typical" 2>&1 | tail -4 ) | tee "$OUT/pytest.log"
{
timeout 900 python profiles/e2e_sweep.py --typical 2>"$OUT/err1.log"
SIMON_NO_SPREAD=1 timeout 900 python profiles/e2e_sweep.py --typical 2>"$OUT/err2.log"
} | tee "$OUT/e2e_typical.txt"
tail -2 "$OUT/err1.log"
