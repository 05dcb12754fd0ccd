#!/bin/bash
# > 128 request signatures on the score-table kernel: parity (tests + fuzz_table) and the cliff as numbers; usage: bash profiles/gpu_r3f.sh <tag>
set -u
TAG=${1:-r3f}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
t0=$(date +%s)
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -x -k "signatures or 64_internal or randomised or many" 2>&1 | tail -8 ) > "$OUT/pytest_sigs.log"; tail -3 "$OUT/pytest_sigs.log"
( timeout 900 python tests/fuzz_table.py 240 3000 2>&1 | tail -6 ) > "$OUT/fuzz_table.log"; tail -3 "$OUT/fuzz_table.log"
echo "tests $(( $(date +%s) - t0 )) s"
{
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sigs 42 (config 3)', 'kernel_ms', d['roofline']['kernel_ms'], 'gen', d['config']['kernel_generation'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
for K in 64 100 128 132 200 300 384 390; do
  timeout 300 python bench.py --workload config3sig --sigs $K --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sigs', $K, 'kernel_ms', d['roofline']['kernel_ms'], 'kernel', d['config']['kernel'], 'gen', d['config']['kernel_generation'], 'lds', d['roofline'].get('lds_bytes_per_workgroup'))"
done
} 2>&1 | tee "$OUT/sig_cliff.txt"
echo "total $(( $(date +%s) - t0 )) s"
