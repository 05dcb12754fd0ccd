#!/bin/bash
# bash profiles/gpu_exp2.sh <tag> "ENV=.. -- bench args" ...   (each arg: env assignments, then --, then bench.py args)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for spec in "$@"; do
  i=$((i+1))
  envs="${spec%%--*}"; args="${spec#*--}"
  ( env $envs timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args ) > "$OUT/run$i.json" 2> "$OUT/run$i.err"
  echo "[$spec] $(grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"scenarios_per_gpu": [0-9]*' "$OUT/run$i.json" | tr '\n' ' ') $(grep -v amdgpu.ids "$OUT/run$i.err" | tail -2 | tr '\n' ' ' | cut -c1-200)"
done
