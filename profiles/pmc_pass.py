#!/usr/bin/env python3
"""One or more rocprofv3 PMC passes over a bench.py workload (separate passes, --kernel-trace only: the recipe of
MI355X_MICROARCH.md), printed as per-dispatch averages of the simon kernels.
usage: python profiles/pmc_pass.py "<bench args>" "CTR CTR ..." ["CTR CTR ..." ...]"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

child = sys.argv[1].split()
for grp in sys.argv[2:]:
    try:
        res = bench._pmc_pass(grp.split(), child, timeout_s=600)
    except Exception as e:      # noqa: BLE001
        print("pass failed:", grp, e)
        continue
    for k, cs in res.items():
        if "simon" in k and "unpermute" not in k and "mask_lanes" not in k:
            print(json.dumps({"kernel": k[:70], **{c: {"dispatches": n, "avg": avg} for c, (n, avg) in cs.items()}}))
