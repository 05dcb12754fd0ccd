#!/bin/bash
# Closing stages of a round (rounds 4 - 6).  usage: bash profiles/gpu_r6.sh <tag> <stage> ...
#   fuzz   the four fuzzers on the new regimes and a slice of the old ones
#   tests  every GPU test + smoke
#   bench  the default bench line + sidecar;  rocprof  kernel trace of the default bench command
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
# Standing first stage: a box that has a Go toolchain closes SURVEY 8(c) and (N2) by itself -- the determinised
# reference against the oracle AND the HIP library, then the never-compiled cgo package through `go vet`.  Prints one line otherwise.
if command -v go > /dev/null 2>&1 || [ -x /usr/local/go/bin/go ]; then
  export PATH=$PATH:/usr/local/go/bin
  ( go version; SIMON_PARITY_ENGINE=both bash oracle/run_ref.sh; echo "run_ref rc=$?";
    if [ -d "${REF:-/root/reference}" ]; then ( cd integration/go && go vet ./... ); echo "go vet rc=$?"; fi ) > "$OUT/go_parity.txt" 2>&1
  tail -5 "$OUT/go_parity.txt"
else
  echo "go: not on this box (SURVEY 8c / N2 stay unpinned)" | tee "$OUT/go_parity.txt"
fi
for STAGE in "$@"; do
  case $STAGE in
    fuzz)
      ( timeout 400 python tests/fuzz_table.py ${FT_N:-300} 300000 2>&1 | tail -4 ) >  "$OUT/fuzzers.txt"
      ( timeout 400 python tests/fuzz_table.py ${FT_N:-300} 7000 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 500 python tests/fuzz_spread.py ${FS_N:-200} 300000 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 500 python tests/fuzz_spread.py ${FS_N:-200} 31000 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 400 python tests/fuzz_rest.py ${FR_N:-200} 9000 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      ( timeout 300 python tests/fuzz_gpu.py ${FG_N:-100} 5000 2>&1 | tail -4 ) >> "$OUT/fuzzers.txt"
      grep -v amdgpu.ids "$OUT/fuzzers.txt" ;;
    tests)
      ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > "$OUT/pytest_gpu.log"; tail -2 "$OUT/pytest_gpu.log"
      ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > "$OUT/smoke.log"; tail -1 "$OUT/smoke.log" ;;
    bench)
      SIMON_BENCH_DETAIL=$OUT/bench_detail.json timeout 1500 python bench.py > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
      tail -1 "$OUT/bench_default.out" > "$OUT/bench_default.json"; wc -c "$OUT/bench_default.json"; cat "$OUT/bench_default.json" ;;
    rocprof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o bench -- python $OLDPWD/bench.py --no-sub --pmc off --no-cpu-baseline --steps 5 > "$OUT/rocprof_bench.out" 2>&1 ); ls "$OUT/rocprof" | head
      mkdir -p "$OUT/rocprof/trace"; find "$OUT/rocprof" -maxdepth 2 -name "*.db" -not -path "*/trace/*" -exec mv {} "$OUT/rocprof/trace/" \;
      python profiles/summarize.py "$OUT/rocprof" > "$OUT/rocprof_bench_default_summary.txt" 2>&1; head -12 "$OUT/rocprof_bench_default_summary.txt"
      rm -rf "$OUT/rocprof" ;;
  esac
done
