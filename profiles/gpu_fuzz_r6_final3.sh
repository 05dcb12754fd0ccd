#!/bin/bash
# End of round 6, third campaign: the tree with unpermute_lds_kernel, fresh seeds again.
set -u
TAG=$1
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
  timeout 600 python tests/fuzz_gpu.py ${FG_N:-600} 98000 2>&1 | tail -2
  timeout 600 python tests/fuzz_table.py ${FT_N:-1000} 99000 2>&1 | tail -2
  timeout 400 python tests/fuzz_table.py 300 667000 2>&1 | tail -2
  timeout 600 python tests/fuzz_rest.py ${FR_N:-1000} 101000 2>&1 | tail -2
  timeout 600 python tests/fuzz_spread.py ${FS_N:-500} 102000 2>&1 | tail -2
  timeout 400 python tests/fuzz_spread.py 200 697000 2>&1 | tail -2
  timeout 400 python tests/fuzz_rest.py ${FR2_N:-300} 547000 2>&1 | tail -2          # 70 .. 110 node classes: two per lane in the REST select (CN2)
  timeout 400 python tests/fuzz_spread.py ${FS2_N:-300} 647000 2>&1 | tail -2        # 65 .. 128 internal classes under the spread walks (CN2)
  timeout 600 python tests/fuzz_spread.py ${FS3_N:-400} 747000 2>&1 | tail -2        # soft constraints next to filters that do not fold (REST && SPREAD), both shapes
  timeout 400 python tests/fuzz_rest.py ${FR3_N:-400} 847000 2>&1 | tail -2          # ... drawn from the REST side (required affinity included)
  timeout 400 python tests/fuzz_table.py ${FT4_N:-500} 1010000 2>&1 | tail -2         # 129 .. 256 node classes on generation 4 (simon_table_cls4.hip)
  timeout 400 python tests/fuzz_spread.py ${FS4_N:-300} 347000 2>&1 | tail -2        # 129 .. 1 023 signatures next to soft constraints, one wave and the team of four
  timeout 400 python tests/fuzz_rest.py ${FR4_N:-300} 1010000 2>&1 | tail -2          # 130 .. 500 pod classes under the REST rows (the signature groups of MANY)
} | grep -v amdgpu.ids > "$OUT/fuzzers_final3.txt"
cat "$OUT/fuzzers_final3.txt"
