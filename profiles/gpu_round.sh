#!/bin/bash
# One GPU-box visit: parity tests, bench variants, rocprofv3 kernel trace (+ optional PMC passes).
# usage (through gpurun): bash profiles/gpu_round.sh <tag> [pmc]
set -u
TAG=${1:-r1}; PMC=${2:-}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 1500 "$OUT/bench_default.json"
SIMON_NO_CACHE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_nocache.json" 2>&1
SIMON_CACHE_BANDS=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_bands1.json" 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --placement 0 > "$OUT/bench_noplace.json" 2>&1
grep -h -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' "$OUT"/bench_*.json | paste - - 
cd /tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
if [ -n "$PMC" ]; then
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 900 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$name" -o pmc -- $BENCH > "$OUT/pmc_$name.log" 2>&1
  done
fi
cd "$ROOT"
python profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
head -30 "$OUT/summary.txt"
# keep the merge-back small: databases are large, the summary is what we commit
find "$OUT" -name "*.db" -size +20M -delete
