#!/bin/bash
# round 2: generation 4 (simon_table.hip) against generation 3 (simon_cache.hip, SIMON_TABLE=0) on ONE box.
# usage: bash profiles/gpu_ab_table.sh <tag> [reps]
set -u
TAG=${1:-abt}; REPS=${2:-2}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
( timeout 900 python -m pytest tests -m gpu -q -x -k "config3 or config2 or config4 or cpu_mem or golden or known_answer or many_sig or large_pool or group or threads or size_independent or narrow or explain or pin" 2>&1 | tail -15 ) > "$OUT/pytest.log"; tail -8 "$OUT/pytest.log"
for rep in $(seq 1 $REPS); do
for V in 1 0; do
  SIMON_TABLE=$V timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --pmc off > "$OUT/bench_t$V.$rep.json" 2> "$OUT/bench_t$V.$rep.err"
  python - "$OUT/bench_t$V.$rep.json" $V <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("SIMON_TABLE="+sys.argv[2], "kernel_ms", d["roofline"]["kernel_ms"], "value", d["value"], "plan", d["config"]["plan"])
except Exception as e: print("failed", e, open(sys.argv[1].replace(".json",".err")).read()[-500:])
PY
done; done
for C in 64 256; do
for V in 1 0; do
  SIMON_TABLE=$V timeout 300 python bench.py --steps 5 --warmup 2 --counts $C --no-cpu-baseline --no-sub --pmc off > "$OUT/bench_c${C}_t$V.json" 2>/dev/null
  python - "$OUT/bench_c${C}_t$V.json" $V $C <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("counts", sys.argv[3], "SIMON_TABLE="+sys.argv[2], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("failed", e)
PY
done; done
timeout 300 python bench.py --workload config3sig --sigs 100 --steps 3 --warmup 1 --no-cpu-baseline --no-sub --pmc off > "$OUT/bench_sig100.json" 2> "$OUT/bench_sig100.err"; python -c "
import json; d=json.load(open('$OUT/bench_sig100.json')); print('100 sigs kernel_ms', d['roofline']['kernel_ms'], d['config']['kernel'])"
