#!/bin/bash
# Round 3, second visit: where the cycle of the rewritten generation 6 goes (phase profile build, config 5 at one wave per CU) and what
# the memory system carries at 256 / 2048 scenarios (live PMC passes through bench.py).   usage: bash profiles/gpu_r3b.sh <tag>
set -u
TAG=${1:-r3b}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
L=$ROOT/open-simulator_amd/csrc/libsimon_hip_tprof.so
for S in 256; do
  SIMON_HIP_LIB=$L SIMON_TABLE_PROF=1 SIMON_BENCH_C5_SCEN=$S timeout 300 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off 2> "$OUT/tprof_$S.err" > "$OUT/tprof_$S.json"
  grep SIMON_TABLE_PROF "$OUT/tprof_$S.err"
done
for S in 256 2048; do
  SIMON_BENCH_C5_SCEN=$S timeout 600 python bench.py --workload config5 --steps 2 --warmup 1 --no-sub --pmc live --no-cpu-baseline > "$OUT/bench_config5_$S.json" 2> "$OUT/bench_config5_$S.err"
  python - "$OUT/bench_config5_$S.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('S', d['config']['scenarios_per_gpu'], 'kernel_ms', r['kernel_ms'], 'valu_frac', r['frac'], 'valu_busy', r.get('valu_pipe_busy_frac'), 'hbm_gbs', r['measured_hbm_gbs'], 'traffic_GB', (r['traffic'] or 0)/1e9, 'split', r.get('wave_time_split'), 'lds_conf', r.get('lds_bank_conflict_cycles_frac'), 'instr', r.get('instructions_per_step'))
PY
done
echo "total $(( $(date +%s) - t0 )) s"
