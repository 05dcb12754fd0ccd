#!/bin/bash
# Compiler scheduling strategy A/B (-mllvm -amdgpu-sched-strategy=max-ilp on the kernel units: libsimon_hip_ilp.so, profiles/build_variant.sh) against the product library, same box.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sched; mkdir -p "$OUT"
V=$PWD/open-simulator_amd/csrc/libsimon_hip_${1:-ilp}.so
W=${2:-c3,c3s64,c5s256,c5s2048,service64,c2}
( python profiles/ab_probe.py $W 3; SIMON_HIP_LIB=$V python profiles/ab_probe.py $W 3; python profiles/ab_probe.py $W 3; SIMON_HIP_LIB=$V python profiles/ab_probe.py $W 3 ) 2>&1 | grep "^AB" > "$OUT/ab_${1:-ilp}.txt"; cat "$OUT/ab_${1:-ilp}.txt"
