#!/bin/bash
# Round 6: same-box A/B over library builds.  usage: bash profiles/gpu_r6a.sh <tag> <workloads> <reps> [lib ...]   (lib = suffix of libsimon_hip_<suffix>.so; "base" = the product build)
set -u
TAG=$1; WL=$2; REPS=$3; shift 3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
if [ "${PROBE_GO:-0}" = 1 ]; then bash profiles/gpu_probe_go.sh > "$OUT/go_probe.txt" 2>&1; grep -A2 "== go version" "$OUT/go_probe.txt"; tail -2 "$OUT/go_probe.txt"; fi
for L in "$@"; do
  if [ "$L" = base ]; then unset SIMON_HIP_LIB; else export SIMON_HIP_LIB=$C/libsimon_hip_$L.so; fi
  timeout ${AB_TIMEOUT:-900} python profiles/ab_probe.py "$WL" "$REPS" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
