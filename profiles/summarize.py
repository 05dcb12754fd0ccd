#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) output directories into a small text report.
usage: python profiles/summarize.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def main(root):
    for db in sorted(glob.glob(os.path.join(root, "trace", "*.db"))):
        con = sqlite3.connect(db)
        print("== kernel trace (rocprofv3 --kernel-trace --stats):", os.path.relpath(db, root))
        print(f"{'kernel':90s} {'calls':>6s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
        for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f"{name[:90]:90s} {calls:6d} {total:14.1f} {avg:12.2f} {pct:7.2f}")
        for r in con.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                             "from kernels group by name"):
            print("   dispatch:", r[0][:70], dict(zip(["grid", "wg", "lds", "vgpr", "agpr", "sgpr", "scratch"], r[1:])))
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
        con = sqlite3.connect(db)
        print("== PMC pass:", os.path.relpath(db, root))
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        for k, c, n, avg in con.execute(q):
            if "simon" in k:
                print(f"   {k[:60]:60s} {c:24s} dispatches={n:3d} avg/dispatch={avg:.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
