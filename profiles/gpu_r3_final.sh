#!/bin/bash
# Round-3 record run (one GPU box): every GPU test, smoke, the default bench line (live PMC on every record), rocprofv3 kernel trace +
# PMC passes of the config-3 bench, of the Service workload and of config 5 at 256 and 2 048 scenarios, the config-5 batch-size row, the Service row on
# both engines, the general path end to end, two fuzzers.
# usage (through gpurun): bash profiles/gpu_r3_final.sh <tag>
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log"
bash oracle/run_ref.sh > "$OUT/run_ref.log" 2>&1; tail -1 "$OUT/run_ref.log"
timeout 900 python bench.py --steps 5 --warmup 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? $(( $(date +%s) - t0 )) s"
{
for S in 256 512 1024 2048 2304 4096; do
  SIMON_BENCH_C5_SCEN=$S timeout 300 python bench.py --workload config5 --steps 2 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config5 scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'scenarios/s', round(d['config']['scenarios_per_gpu']/d['roofline']['kernel_ms']*1e3), 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
done
} > "$OUT/config5_batch_size_row.txt"; cat "$OUT/config5_batch_size_row.txt"
{
for CNT in 64 256 1024; do
  timeout 600 python bench.py --workload service --counts $CNT --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
done
SIMON_NO_SPREAD=1 timeout 600 python bench.py --workload service --counts 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service on the all-feature kernel (SIMON_NO_SPREAD=1): scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
} > "$OUT/service_row.txt"; cat "$OUT/service_row.txt"
timeout 600 python profiles/e2e_sweep.py > "$OUT/e2e_general_path.json" 2>/dev/null; cat "$OUT/e2e_general_path.json"
( timeout 600 python tests/fuzz_gpu.py 150 9000 2>&1 | tail -2 ) > "$OUT/fuzz_gpu.log"; tail -1 "$OUT/fuzz_gpu.log"
( timeout 600 python tests/fuzz_spread.py 300 9000 2>&1 | tail -2 ) > "$OUT/fuzz_spread.log"; tail -1 "$OUT/fuzz_spread.log"
echo "rows + fuzz $(( $(date +%s) - t0 )) s"
cd /tmp
for W in config3 service config5_S256 config5_S2048; do
  case $W in
    config3) B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off";;
    service) B="python $ROOT/bench.py --workload service --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc off";;
    config5_S256) B="python $ROOT/bench.py --workload config5 --c5-scenarios 256 --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off";;
    config5_S2048) B="python $ROOT/bench.py --workload config5 --c5-scenarios 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off";;
  esac
  mkdir -p "$OUT/$W"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$W/trace" -o trace -- $B > "$OUT/$W/trace.log" 2>&1
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/$W/pmc_$i" -o pmc -- $B > "$OUT/$W/pmc_$i.log" 2>&1
  done
  ( cd "$ROOT" && python profiles/summarize.py "$OUT/$W" > "$OUT/${W}_summary.txt" 2>&1 )
  head -8 "$OUT/${W}_summary.txt"
done
cd "$ROOT"
find "$OUT" -name "*.db" -size +6M -delete
echo "total $(( $(date +%s) - t0 )) s"
