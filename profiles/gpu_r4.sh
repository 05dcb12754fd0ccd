#!/bin/bash
# Round-4 GPU stages (ONE launcher; a stage per argument).  usage: bash profiles/gpu_r4.sh <tag> <stage> [<stage> ...]
#   team_tests  tests/test_gpu_round4.py + the generation-7 tests of round 3
#   fuzz_spread tests/fuzz_spread.py (both shapes per case), N cases (env FUZZ_N, default 300)
#   team_ab     profiles/team_ab.py (one wave against a team of waves over the batch size) + the typical cluster x 64 sizes on both shapes
#   all_tests   every GPU test + smoke
#   bench       the default bench line (what the driver runs) + its sidecar
#   rocprof     rocprofv3 --kernel-trace --stats of the default bench command (no PMC in the same run)
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for STAGE in "$@"; do
  t0=$(date +%s)
  case $STAGE in
    team_tests)
      ( timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -x -k "team or shape or spread or service or generation_7 or fuzz" 2>&1 | tail -15 ) > "$OUT/pytest_team.log"; tail -3 "$OUT/pytest_team.log" ;;
    fuzz_spread)
      ( timeout 1500 python tests/fuzz_spread.py ${FUZZ_N:-300} ${FUZZ_FIRST:-20000} 2>&1 | tail -20 ) > "$OUT/fuzz_spread.log"; tail -2 "$OUT/fuzz_spread.log" ;;
    team_ab)
      ( timeout 900 python profiles/team_ab.py 2>&1 | tail -12 ) > "$OUT/team_ab_service.txt"; cat "$OUT/team_ab_service.txt"
      ( timeout 600 python profiles/team_ab.py --pref 60 --sizes 16,64,256 2>&1 | tail -6 ) > "$OUT/team_ab_service_pref.txt"; cat "$OUT/team_ab_service_pref.txt"
      for T in 0 4; do ( SIMON_TEAM=$T timeout 600 python profiles/e2e_sweep.py --typical --counts 64 2>&1 | tail -1 ) > "$OUT/typical_x64_team$T.txt"; cat "$OUT/typical_x64_team$T.txt"; done
      for C in 8 256; do ( timeout 600 python profiles/e2e_sweep.py --typical --counts $C 2>&1 | tail -1 ) > "$OUT/typical_x${C}_auto.txt"; cat "$OUT/typical_x${C}_auto.txt"; done ;;
    all_tests)
      ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > "$OUT/pytest_gpu.log"; tail -2 "$OUT/pytest_gpu.log"
      ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log" ;;
    bench)
      SIMON_BENCH_DETAIL=$OUT/bench_detail.json timeout 1500 python bench.py > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
      tail -1 "$OUT/bench_default.out" > "$OUT/bench_default.json"; wc -c "$OUT/bench_default.json"; cat "$OUT/bench_default.json" ;;
    rocprof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o bench -- python $OLDPWD/bench.py --no-sub --pmc off --no-cpu-baseline --steps 5 > "$OUT/rocprof_bench.out" 2>&1 ); ls "$OUT/rocprof" | head ;;
    tprof)   # phase profile (bash profiles/build_variant.sh tprof -DSIMON_TABLE_PROFILE first): the leader's ticks per scheduling cycle, both shapes
      for T in 0 4; do
        ( SIMON_TEAM=$T SIMON_TABLE_PROF=1 SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_tprof.so timeout 600 python profiles/team_ab.py --sizes 16 --only $T 2>&1 | grep -v amdgpu.ids | tail -8 ) > "$OUT/tprof_service_S64_team$T.txt"; cat "$OUT/tprof_service_S64_team$T.txt"
        ( SIMON_TEAM=$T SIMON_TABLE_PROF=1 SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_tprof.so timeout 600 python profiles/e2e_sweep.py --typical --counts 64 2>&1 | grep -v amdgpu.ids | tail -4 ) > "$OUT/tprof_typical_x64_team$T.txt"; cat "$OUT/tprof_typical_x64_team$T.txt"
      done ;;
    libab)   # same-box A/B of library variants (env LIBS="name name ..": open-simulator_amd/csrc/libsimon_hip_<name>.so; "main" = the product build)
      for L in ${LIBS:-main}; do
        LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_$L.so; [ $L = main ] && LIB=$PWD/open-simulator_amd/csrc/libsimon_hip.so
        for REP in 1 2; do
          echo "$L service_S64 $( SIMON_HIP_LIB=$LIB timeout 300 python profiles/team_ab.py --sizes 16 --only 4 2>&1 | tail -1 )" | tee -a "$OUT/libab.txt"
          echo "$L typical_x64 $( SIMON_TEAM=4 SIMON_HIP_LIB=$LIB timeout 300 python profiles/e2e_sweep.py --typical --counts 64 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["kernel_ms"], d["workgroup"])' )" | tee -a "$OUT/libab.txt"
        done
      done ;;
    regress)   # the single-wave kernels against another build of simon_table.hip, same box (env LIBS, as libab): kernel ms of the benchmarked workloads
      for L in ${LIBS:-main}; do
        LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_$L.so; [ $L = main ] && LIB=$PWD/open-simulator_amd/csrc/libsimon_hip.so
        IFS=';' read -ra WL <<< "${WORKLOADS:-config3;service;service --pref 60;config3sig --sigs 200;config5 --c5-scenarios 256;config5 --c5-scenarios 2048}"
        for W in "${WL[@]}"; do
          for REP in 1 2; do
            echo "$L [$W] $( SIMON_HIP_LIB=$LIB SIMON_BENCH_DETAIL=/tmp/d.json timeout 600 python bench.py --workload $W --pmc off --no-cpu-baseline --no-sub --steps 5 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["config"]["kernel_generation"])' )" | tee -a "$OUT/regress.txt"
          done
        done
      done ;;
    new_tests)   # the tests added since the last full run (env KEXPR = pytest -k expression)
      ( timeout 1500 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "${KEXPR:-local or wrap or abi_v3 or hundreds}" 2>&1 | tail -15 ) > "$OUT/pytest_new.log"; tail -3 "$OUT/pytest_new.log" ;;
    wide_exp)    # the all-feature kernel on the random mix x 64 sizes: workgroup shapes (env SIMON_WG) and its phase profile (prof build)
      for WG in ${WGS:-512 1024}; do
        echo "WG=$WG $( SIMON_WG=$WG SIMON_BENCH_DETAIL=/tmp/d.json timeout 600 python bench.py --workload widemix --pmc off --no-cpu-baseline --no-sub --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["config"].get("workgroup"))' )" | tee -a "$OUT/wide_exp.txt"
      done
      if [ -f open-simulator_amd/csrc/libsimon_hip_prof.so ]; then
        ( SIMON_WIDE_PROF=1 SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_prof.so SIMON_BENCH_DETAIL=/tmp/d.json timeout 600 python bench.py --workload widemix --pmc off --no-cpu-baseline --no-sub --steps 1 --warmup 0 2>&1 | grep SIMON_WIDE_PROF | tail -2 ) > "$OUT/wide_phase_profile.txt"; cut -c1-900 "$OUT/wide_phase_profile.txt"
      fi ;;
    wide_pmc)    # instruction-cache and wave-state counters of the all-feature kernel on the random mix (separate passes, kernel-trace only)
      ( timeout 900 python profiles/pmc_pass.py "--workload widemix --steps 1 --warmup 0" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU" 2>&1 | tail -6 ) > "$OUT/wide_pmc.txt"; cut -c1-1200 "$OUT/wide_pmc.txt" ;;
    wide_tests)  # the all-feature kernel's parity tests (random feature sets, three workgroup shapes) + the k8s-object sweeps on it
      ( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "v2_features or k8s or explain" 2>&1 | tail -8 ) > "$OUT/pytest_wide.log"; tail -3 "$OUT/pytest_wide.log" ;;
    *) echo "unknown stage $STAGE" ;;
  esac
  echo "[$STAGE] $(( $(date +%s) - t0 )) s"
done
