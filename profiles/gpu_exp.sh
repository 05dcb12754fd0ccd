#!/bin/bash
# Quick A/B on the GPU box: bash profiles/gpu_exp.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one bench run)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for envs in "$@"; do
  i=$((i+1))
  ( env $envs timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > "$OUT/run$i.json" 2> "$OUT/run$i.err"
  echo "[$envs] $(grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"kernel": "[a-z_]*"' "$OUT/run$i.json" | tr '\n' ' ') $(tail -2 "$OUT/run$i.err" | tr '\n' ' ' | cut -c1-300)"
done
