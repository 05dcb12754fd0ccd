#!/bin/bash
# signature twins (65..128 signatures: the upper half sits 64 slots above a signature with the same request): parity + config-5 A/B; usage: bash profiles/gpu_r3aa.sh <tag>
set -u
TAG=${1:-r3aa}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 1500 python -m pytest tests -m gpu -q -x -k "config5 or rest or signatures or round2 or cpu_mem" 2>&1 | tail -5 ) > "$OUT/pytest.log"; tail -2 "$OUT/pytest.log"
( timeout 900 python tests/fuzz_rest.py 300 80000 2>&1 | tail -3 ) > "$OUT/fuzz_rest.log"; tail -1 "$OUT/fuzz_rest.log"
( timeout 900 python tests/fuzz_table.py 300 81000 2>&1 | tail -3 ) > "$OUT/fuzz_table.log"; tail -1 "$OUT/fuzz_table.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
for S in 256 2048; do
  for V in twins no_twins twins no_twins; do
    E="X=1"; [ $V = no_twins ] && E="SIMON_TABLE_NO_TWINS=1"
    env $E SIMON_BENCH_C5_SCEN=$S timeout 300 python bench.py --workload config5 --steps 2 --warmup 1 --no-sub --pmc off --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config5', '$V', 'scenarios', d['config']['scenarios_per_gpu'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done
} | tee "$OUT/config5_twins_ab.txt"
echo "total $(( $(date +%s) - t0 )) s"
