#!/bin/bash
# Is config 3's batch (4 096 single-wave workgroups, 9 408 B of LDS each) placed evenly over the CUs and SIMDs?  The placement probe, then the
# kernel with extra LDS per workgroup (SIMON_CACHE_LDS_PAD) so that exactly 16 fit per CU.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/balance; mkdir -p "$OUT"
{
  profiles/micro/placement_probe 4096 9408
  profiles/micro/placement_probe 4096 10240
  profiles/micro/placement_probe 2048 9408
} > "$OUT/placement.txt" 2>&1; cat "$OUT/placement.txt"
( for pad in 0 640 896 0 640; do SIMON_CACHE_LDS_PAD=$pad python profiles/ab_probe.py c3 3 2>&1 | grep "^AB" | sed "s/^AB/AB pad=$pad/"; done ) > "$OUT/ab_pad.txt"; cat "$OUT/ab_pad.txt"
