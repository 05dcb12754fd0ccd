#!/bin/bash
# Odd row pitch of the class-term rows (s_sn) in the 129..256-class kernels: same-box A/B, HEAD library against the experiment (libsimon_hip_exp.so), interleaved twice.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cls4p; mkdir -p "$OUT"
EXP=$PWD/open-simulator_amd/csrc/libsimon_hip_exp.so
( for i in 1 2; do python profiles/ab_probe.py c3cls160 3; SIMON_HIP_LIB=$EXP python profiles/ab_probe.py c3cls160 3; done ) 2>&1 | grep "^AB" > "$OUT/ab2.txt"; cat "$OUT/ab2.txt"
