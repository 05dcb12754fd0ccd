#!/bin/bash
# GPU-box visit r1e: all parity tests, smoke, default bench, config-5 bench, rocprofv3 kernel trace + PMC passes of BOTH.
# usage (through gpurun): bash profiles/gpu_r1e.sh <tag>
set -u
TAG=${1:-r1e}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 400 "$OUT/bench_default.json"; echo
timeout 420 python bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; tail -c 300 "$OUT/bench_config5.json"; echo
SIMON_BENCH_C5_SCEN=1024 timeout 420 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench_config5_s1024.json" 2> "$OUT/bench_config5_s1024.err"; tail -c 300 "$OUT/bench_config5_s1024.json"; echo
timeout 300 python bench.py --steps 3 --warmup 1 --orders-per-gpu 8 --no-cpu-baseline > "$OUT/bench_8192.json" 2>/dev/null; python - "$OUT/bench_8192.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("8192 scenarios/GPU:", d["value"], "scen/s, kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("8192 run failed", e)
PY
cd /tmp
for W in config3 config5; do
  if [ $W = config3 ]; then B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"; else B="python $ROOT/bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline"; fi
  mkdir -p "$OUT/$W"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$W/trace" -o trace -- $B > "$OUT/$W/trace.log" 2>&1
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/$W/pmc_$i" -o pmc -- $B > "$OUT/$W/pmc_$i.log" 2>&1
  done
  ( cd "$ROOT" && python profiles/summarize.py "$OUT/$W" > "$OUT/${W}_summary.txt" 2>&1 )
  head -12 "$OUT/${W}_summary.txt"
done
cd "$ROOT"
SIMON_WIDE_PROF=1 timeout 300 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep SIMON_WIDE_PROF > "$OUT/config5_phase_profile.txt"; cat "$OUT/config5_phase_profile.txt"
find "$OUT" -name "*.db" -size +6M -delete
echo "total $(( $(date +%s) - t0 )) s"
