export TMPDIR=/tmp; O=gpurun_out/r06c; mkdir -p $O
for S in 256 16; do
SIMON_DEBUG_ROUTE=1 SIMON_BENCH_DETAIL=/tmp/d.json SIMON_BENCH_CPU_BUDGET_S=8 timeout 600 python bench.py --workload config5service --c5-scenarios $S --pmc off --no-sub --steps 3 --warmup 1 > $O/c5s_$S.out 2> $O/c5s_$S.err; echo rc=$?
grep -h "route" $O/c5s_$S.err | head -3
tail -1 $O/c5s_$S.out | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel"], d["config"]["kernel_generation"], d["roofline"]["kernel_ms"], d.get("parity_sample"), d.get("cpu_baseline",{}).get("value"))'
done
