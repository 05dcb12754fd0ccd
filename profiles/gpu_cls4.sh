#!/bin/bash
# 129 .. 256 node classes on generation 4 (simon_table_cls4.hip): tests, fuzz regime, same-box A/B against the previous library.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/cls4; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -k "129_to_256 or 160_node_shapes or beyond_128" 2>&1 | tail -5 ) > "$OUT/tests.txt"; tail -3 "$OUT/tests.txt"
( timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "128_distinct" 2>&1 | tail -5 ) >> "$OUT/tests.txt"; tail -2 "$OUT/tests.txt"
( timeout 600 python tests/fuzz_table.py ${FT_N:-150} 900000 2>&1 | tail -4 ) > "$OUT/fuzz.txt"; grep -v amdgpu.ids "$OUT/fuzz.txt"
( python profiles/ab_probe.py c3cls160,c3cls80,c3,c3s64 3; SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_prev.so python profiles/ab_probe.py c3cls160,c3cls80,c3,c3s64 3 ) 2>&1 | grep "^AB" > "$OUT/ab.txt"; cat "$OUT/ab.txt"
( SIMON_TABLE_COARSE=0 python profiles/ab_probe.py c3cls80 3 ) 2>&1 | grep "^AB" | sed 's/^AB/AB one-level/' >> "$OUT/ab.txt"; tail -1 "$OUT/ab.txt"
