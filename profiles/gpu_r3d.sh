#!/bin/bash
# parity of the REST path (tests + fuzz) followed by a same-box A/B of config 5; usage: bash profiles/gpu_r3d.sh <tag> "<S list>" label=lib ...
set -u
TAG=$1; SL=$2; shift 2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -k "rest or config5 or gpushare or explain or 64_internal or fixtures or golden or ranks" 2>&1 | tail -8 ) > "$OUT/pytest_rest.log"; tail -3 "$OUT/pytest_rest.log"
( timeout 600 python tests/fuzz_rest.py 500 9000 2>&1 | tail -5 ) > "$OUT/fuzz_rest.log"; tail -2 "$OUT/fuzz_rest.log"
bash profiles/gpu_ab_c5.sh $TAG "$SL" "$@"
