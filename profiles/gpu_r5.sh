#!/bin/bash
# Round-5 launcher.  usage: bash profiles/gpu_r5.sh <tag> <stage> ...   (through gpurun; results under gpurun_out/<tag>/)
#   new_tests  pytest -m gpu -k "$KEXPR" on $TESTS (default: tests/test_gpu_round5.py)
#   tests      every GPU test + smoke
#   regress    kernel ms of workloads under library variants, same box: env LIBS="main abl1 ..", WORKLOADS="config3;config5 --c5-scenarios 256"
#              (open-simulator_amd/csrc/libsimon_hip_<name>.so from profiles/build_variant.sh; "main" = the product build), env ENVS for extra env
#   fuzz       the fuzzers (slices)
#   bench      the default bench line + sidecar;  rocprof  kernel trace of the default bench command
#   pmc        rocprofv3 --pmc passes (separate runs, kernel-trace only) of the workloads in PMC_WORKLOADS, summarised by profiles/summarize.py
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for STAGE in "$@"; do
  case $STAGE in
    new_tests)
      ( timeout ${TEST_TIMEOUT:-1500} python -m pytest ${TESTS:-tests/test_gpu_round5.py} -m gpu -q -x ${KEXPR:+-k "$KEXPR"} 2>&1 | tail -25 ) > "$OUT/pytest_new.log"; tail -6 "$OUT/pytest_new.log" ;;
    tests)
      ( timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
      ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log" ;;
    regress)
      for L in ${LIBS:-main}; do
        LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_$L.so; [ $L = main ] && LIB=$PWD/open-simulator_amd/csrc/libsimon_hip.so
        IFS=';' read -ra WL <<< "${WORKLOADS:-config3;config5 --c5-scenarios 256}"
        for W in "${WL[@]}"; do
          for REP in $(seq 1 ${REPS:-2}); do
            echo "$L [$W] ${ENVS:-} $( env ${ENVS:-} SIMON_HIP_LIB=$LIB SIMON_BENCH_DETAIL=/tmp/d.json timeout 600 python bench.py --workload $W --pmc off --no-cpu-baseline --no-sub --steps ${STEPS:-5} 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["config"]["kernel_generation"], d["config"].get("workgroup"))' )" | tee -a "$OUT/regress.txt"
          done
        done
      done ;;
    fuzz)
      ( timeout 400 python tests/fuzz_table.py ${FT_N:-300} ${FT_SEED:-510000} 2>&1 | tail -4 ) >  "$OUT/fuzzers.txt"
      ( timeout 500 python tests/fuzz_rest.py ${FR_N:-300} ${FR_SEED:-59000} 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 500 python tests/fuzz_spread.py ${FS_N:-150} ${FS_SEED:-531000} 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 300 python tests/fuzz_gpu.py ${FG_N:-100} ${FG_SEED:-55000} 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      grep -v amdgpu.ids "$OUT/fuzzers.txt" ;;
    bench)
      SIMON_BENCH_DETAIL=$OUT/bench_detail.json timeout 1500 python bench.py > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
      tail -1 "$OUT/bench_default.out" > "$OUT/bench_default.json"; wc -c "$OUT/bench_default.json"; cat "$OUT/bench_default.json" ;;
    rocprof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o bench -- python $OLDPWD/bench.py --no-sub --pmc off --no-cpu-baseline --steps 5 > "$OUT/rocprof_bench.out" 2>&1 ); ls "$OUT/rocprof" | head
      mkdir -p "$OUT/rocprof/trace"; find "$OUT/rocprof" -maxdepth 2 -name "*.db" -not -path "*/trace/*" -exec mv {} "$OUT/rocprof/trace/" \;
      python profiles/summarize.py "$OUT/rocprof" > "$OUT/rocprof_bench_default_summary.txt" 2>&1; head -12 "$OUT/rocprof_bench_default_summary.txt"
      rm -rf "$OUT/rocprof" ;;
    pmc)
      # one rocprofv3 run per counter group (never together with a trace domain other than the kernel trace), per workload
      IFS=';' read -ra WL <<< "${PMC_WORKLOADS:-config3;config5 --c5-scenarios 256;config5 --c5-scenarios 2048;widemix}"
      for W in "${WL[@]}"; do
        NAME=$(echo "$W" | tr -c 'A-Za-z0-9\n' '_' | sed 's/__*/_/g; s/_$//')
        D="$OUT/pmc_$NAME"; mkdir -p "$D/trace"
        ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$D/trace" -o t -- python $OLDPWD/bench.py --workload $W --no-sub --pmc off --no-cpu-baseline --steps 2 --warmup 1 > "$D/trace.out" 2>&1 )
        I=0
        for GRP in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
          I=$((I+1)); mkdir -p "$D/pmc_$I"
          ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $GRP -d "$D/pmc_$I" -o p -- python $OLDPWD/bench.py --workload $W --no-sub --pmc off --no-cpu-baseline --steps 1 --warmup 0 > "$D/pmc_$I.out" 2>&1 )
        done
        python profiles/summarize.py "$D" > "$OUT/${NAME}_summary.txt" 2>&1; head -30 "$OUT/${NAME}_summary.txt"
        rm -rf "$D"
      done ;;
  esac
done
