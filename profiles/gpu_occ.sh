#!/bin/bash
# cache kernel occupancy experiment: extra LDS per workgroup (fewer resident waves per CU) against time, same box
for rep in 1 2; do for pad in 0 3000 6000 12000 26000; do
SIMON_CACHE_LDS_PAD=$pad python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pad', $pad, 'kernel_ms', d['roofline']['kernel_ms'], 'lds', d['config'].get('lds_bytes'))"
done; done
