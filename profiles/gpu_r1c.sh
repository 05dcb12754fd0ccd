#!/bin/bash
# GPU-box visit r1c: parity tests (ABI v2 wide kernel), smoke, config-5 bench, default bench.
# usage (through gpurun): bash profiles/gpu_r1c.sh <tag>
set -u
TAG=${1:-r1c}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_size_independent_properties_full_batch 2>&1 | tail -40 ) > "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
echo "pytest took $(( $(date +%s) - t0 )) s"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > "$OUT/smoke.log"; tail -2 "$OUT/smoke.log"
timeout 420 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; tail -c 1500 "$OUT/bench_config5.json"; tail -3 "$OUT/bench_config5.err"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 2500 "$OUT/bench_default.json"
echo "total $(( $(date +%s) - t0 )) s"
