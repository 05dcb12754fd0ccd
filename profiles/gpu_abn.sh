#!/bin/bash
# N-way comparison of library builds on ONE box: bash profiles/gpu_abn.sh <tag> <scenarios> <reps> name1 name2 ...
# (name "default" = libsimon_hip.so, other names = libsimon_hip_<name>.so from profiles/build_variant.sh)
set -u
TAG=$1; S=$2; REPS=$3; shift 3
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
for V in "$@"; do
  L=$PWD/open-simulator_amd/csrc/libsimon_hip_$V.so; [ $V = default ] && L=$PWD/open-simulator_amd/csrc/libsimon_hip.so
  SIMON_HIP_LIB=$L SIMON_BENCH_C5_SCEN=$S timeout 400 python bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$V.$S.$rep.json" 2> "$OUT/$V.$S.$rep.err"
  python - "$OUT/$V.$S.$rep.json" $V $S <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], "S", sys.argv[3], "wg", d["config"]["workgroup"], "ms", d["roofline"]["kernel_ms"], "plan", d["config"]["plan"])
except Exception as e: print(sys.argv[2], "failed", e)
PY
done; done
