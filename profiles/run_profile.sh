#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash profiles/run_profile.sh <tag> [bench args]
# 1) kernel trace + stats, 2) separate PMC passes (never combined with trace domains other than
# --kernel-trace).  Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r1}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1
done
find "$OUT" -name "*.csv" | head -40
