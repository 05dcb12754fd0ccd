#!/bin/bash
# bash profiles/gpu_pmc.sh <tag> "<bench args>" "CTR1 CTR2 ..." "CTR3 ..."   -- one rocprofv3 --pmc pass per counter group
set -u
TAG=$1; ARGS=$2; shift 2
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$i" -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > "$OUT/pmc_$i.log" 2>&1
done
cd "$ROOT"
python profiles/summarize.py "$OUT" 2>&1 | grep -v "plan_kernel\|unpermute" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
find "$OUT" -name "*.db" -size +8M -delete
