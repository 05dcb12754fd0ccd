#!/bin/bash
# the general path end to end (profiles/e2e_sweep.py) with the product build and with the phase profiler; usage: bash profiles/gpu_r3p.sh <tag>
set -u
TAG=${1:-r3p}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
timeout 600 python profiles/e2e_sweep.py > "$OUT/e2e.json" 2> "$OUT/e2e.err"; cat "$OUT/e2e.json"
SIMON_HIP_LIB=$C/libsimon_hip_wprof.so SIMON_WIDE_PROF=1 timeout 600 python profiles/e2e_sweep.py > "$OUT/e2e_prof.json" 2> "$OUT/e2e_prof.err"; cat "$OUT/e2e_prof.json"; grep SIMON_WIDE_PROF "$OUT/e2e_prof.err"
