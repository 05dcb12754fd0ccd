#!/bin/bash
# generation 7: PMC counters of the Service workload (4 096 scenarios); usage: bash profiles/gpu_r3l.sh <tag>
set -u
TAG=${1:-r3l}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 900 python bench.py --workload service --steps 2 --warmup 1 --no-cpu-baseline --no-sub --pmc live 2>"$OUT/err.log" | tee "$OUT/service_pmc.json" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['roofline'], indent=1))"
tail -3 "$OUT/err.log"
