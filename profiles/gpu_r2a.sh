#!/bin/bash
# round 2, visit a: all GPU tests (old + new), default bench line (live PMC), self-spawned 2-rank bench.
set -u
TAG=${1:-r2a}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -40 ) > "$OUT/pytest.log"; tail -25 "$OUT/pytest.log"
echo "pytest done $(( $(date +%s) - t0 )) s"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; tail -c 3000 "$OUT/bench_default.json"; tail -5 "$OUT/bench_default.err"
echo "total $(( $(date +%s) - t0 )) s"
