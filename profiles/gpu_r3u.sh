#!/bin/bash
# generation 7: the general walk of spread_select forced for (almost) every pod (a library built with a score table of 8 entries); usage: bash profiles/gpu_r3u.sh <tag>
set -u
TAG=${1:-r3u}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
C=$PWD/open-simulator_amd/csrc
export SIMON_HIP_LIB=$C/libsimon_hip_tab8.so
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "spread or service or generation_7" 2>&1 | tail -5 ) | tee "$OUT/pytest_spread_general_walk.log"
( timeout 900 python tests/fuzz_spread.py 600 50000 2>&1 | tail -4 ) | tee "$OUT/fuzz_spread_general_walk.log"
timeout 600 python bench.py --workload service --counts 64 --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('service, general walk: scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])" | tee -a "$OUT/fuzz_spread_general_walk.log"
