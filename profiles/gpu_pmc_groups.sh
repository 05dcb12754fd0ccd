#!/bin/bash
# arbitrary PMC groups over the default bench (1 step): bash profiles/gpu_pmc_groups.sh <tag> "<env>" "CTR CTR .." "CTR .." ...
set -u
TAG=$1; ENVS=$2; shift 2
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp
i=0
for G in "$@"; do
  i=$((i+1))
  env $ENVS timeout 300 rocprofv3 --kernel-trace --pmc $G -d "$OUT/p$i" -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sub --pmc off > "$OUT/p$i.log" 2>&1
done
cd $ROOT
python - "$OUT" <<'PY'
import glob,sqlite3,sys,os
for db in sorted(glob.glob(os.path.join(sys.argv[1],"p*","**","*.db"),recursive=True)):
    con=sqlite3.connect(db)
    agg={}
    for k,c,n,avg,tot in con.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"):
        if "table_kernel" in k or "cache_kernel" in k:
            a=agg.setdefault(c,[0,0.0]); a[0]+=n; a[1]+=tot
    for c,(n,tot) in sorted(agg.items()): print(f"{c:28s} dispatches={n} sum={tot:.6g} per scenario pod-cycle={tot/4096e4:.2f}")
    con.close()
    for f in glob.glob(os.path.dirname(db)+"/*.db"):
        if os.path.getsize(f) > 6e6: os.remove(f)
PY
