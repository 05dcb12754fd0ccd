#!/bin/bash
# live-PMC roofline records of the default bench for the given environment settings (one box): bash profiles/gpu_pmc_quick.sh <tag> "ENV=.. ENV=.." ["ENV.."...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for E in "$@"; do
  i=$((i+1))
  env $E timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub --pmc live > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
  python - "$OUT/bench_$i.json" "$E" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], "| kernel_ms", r["kernel_ms"], "valu_frac", r["frac"], "hbm_gbs", r["measured_hbm_gbs"], "traffic", r["traffic"], "lds_frac", r["lds_frac"], "bank_conf", r.get("lds_bank_conflict_cycles_frac"), r.get("wave_time_split"), {k: (v and round(v/4096e4,1)) for k,v in r["instructions_per_step"].items()})
except Exception as e: print("failed", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
