#!/bin/bash
# Round 6: the four fuzzers at scale on the final kernels.  usage: bash profiles/gpu_fuzz_r6.sh <tag>
set -u
TAG=$1
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
  timeout 900 python tests/fuzz_gpu.py ${FG_N:-1000} 61000 2>&1 | tail -2
  timeout 900 python tests/fuzz_table.py ${FT_N:-1200} 62000 2>&1 | tail -2
  timeout 900 python tests/fuzz_table.py 300 630000 2>&1 | tail -2
  timeout 900 python tests/fuzz_rest.py ${FR_N:-1200} 64000 2>&1 | tail -2
  timeout 900 python tests/fuzz_spread.py ${FS_N:-600} 65000 2>&1 | tail -2
  timeout 900 python tests/fuzz_spread.py 200 660000 2>&1 | tail -2
  timeout 900 python tests/fuzz_rest.py ${FR2_N:-400} 510000 2>&1 | tail -2          # 70 .. 110 node classes: two per lane in the REST select (CN2)
  timeout 900 python tests/fuzz_spread.py ${FS2_N:-400} 610000 2>&1 | tail -2        # 65 .. 128 internal classes under the spread walks (CN2)
  timeout 1200 python tests/fuzz_spread.py ${FS3_N:-600} 710000 2>&1 | tail -2       # soft constraints next to filters that do not fold: the walks over the mask rows (REST && SPREAD), both shapes
  timeout 900 python tests/fuzz_rest.py ${FR3_N:-500} 810000 2>&1 | tail -2          # ... drawn from the REST side (required affinity included)
  timeout 900 python tests/fuzz_table.py ${FT4_N:-600} 910000 2>&1 | tail -2         # 129 .. 256 node classes on generation 4 (simon_table_cls4.hip)
  timeout 900 python tests/fuzz_spread.py ${FS4_N:-300} 310000 2>&1 | tail -2        # 129 .. 1 023 signatures next to soft constraints, one wave and the team of four
  timeout 900 python tests/fuzz_rest.py ${FR4_N:-300} 910000 2>&1 | tail -2          # 130 .. 500 pod classes under the REST rows (the signature groups of MANY)
} | grep -v amdgpu.ids > "$OUT/fuzzers_at_scale.txt"
cat "$OUT/fuzzers_at_scale.txt"
