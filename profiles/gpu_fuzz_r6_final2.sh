#!/bin/bash
# End of round 6, second campaign on the final tree (after the team / REST && SPREAD units were halved): every regime, fresh seeds, about three times the first campaign.
set -u
TAG=$1
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
{
  timeout 900 python tests/fuzz_gpu.py ${FG_N:-1500} 121000 2>&1 | tail -2
  timeout 900 python tests/fuzz_table.py ${FT_N:-3000} 122000 2>&1 | tail -2
  timeout 600 python tests/fuzz_table.py 900 720000 2>&1 | tail -2
  timeout 900 python tests/fuzz_rest.py ${FR_N:-3000} 124000 2>&1 | tail -2
  timeout 900 python tests/fuzz_spread.py ${FS_N:-1500} 125000 2>&1 | tail -2
  timeout 600 python tests/fuzz_spread.py 900 670000 2>&1 | tail -2          # 65 .. 128 internal classes under the spread walks (CN2)
  timeout 600 python tests/fuzz_rest.py 900 570000 2>&1 | tail -2            # 70 .. 110 node classes: two per lane in the REST select (CN2)
  timeout 900 python tests/fuzz_spread.py 1200 770000 2>&1 | tail -2         # REST && SPREAD from the spread side, both shapes (simon_table_rs.hip / rsz, the team units)
  timeout 900 python tests/fuzz_rest.py 1200 870000 2>&1 | tail -2           # ... from the REST side (required affinity included)
  timeout 600 python tests/fuzz_table.py 1500 970000 2>&1 | tail -2          # 129 .. 256 node classes on generation 4
  timeout 600 python tests/fuzz_spread.py 900 370000 2>&1 | tail -2          # 129 .. 1 023 signatures next to soft constraints
  timeout 600 python tests/fuzz_rest.py 900 970000 2>&1 | tail -2            # 130 .. 500 pod classes under the REST rows
} | grep -v amdgpu.ids > "$OUT/fuzzers_final2.txt"
cat "$OUT/fuzzers_final2.txt"
