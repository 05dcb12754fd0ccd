#!/bin/bash
# Lean GPU-box visit: parity tests, smoke, default bench, kernel trace, 3 PMC passes, config-5 bench.
# usage (through gpurun): bash profiles/gpu_r1b.sh <tag>
set -u
TAG=${1:-r1b}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > "$OUT/smoke.log"; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 2500 "$OUT/bench_default.json"
cd /tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$i" -o pmc -- $BENCH > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
head -40 "$OUT/summary.txt"
timeout 420 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; tail -c 1200 "$OUT/bench_config5.json"
find "$OUT" -name "*.db" -size +8M -delete
