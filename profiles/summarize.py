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


def derived(root):
    """Figures the reviews recompute by hand, from the passes above, for the simon kernels that ran more than a blink: VALU issue fraction
    (SQ_INSTS_VALU / average duration / 1 228.8 G wave-instructions per second = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles), VALU pipe occupancy,
    HBM traffic (FETCH_SIZE and WRITE_SIZE are in KB; FETCH_SIZE doubled: the gfx950 counter tallies 128-byte requests at 64 bytes,
    MI355X_MICROARCH.md section HBM) against 8 TB/s, the share of LDS-active cycles lost to bank conflicts, the wave-time split."""
    dur = {}
    for db in sorted(glob.glob(os.path.join(root, "trace", "*.db"))):
        con = sqlite3.connect(db)
        for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            dur[name] = avg * 1e-6                                                   # seconds per dispatch
    ctr = {}
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
        con = sqlite3.connect(db)
        for k, c, avg in con.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            ctr.setdefault(k, {})[c] = avg
    print("== derived (per dispatch; durations from the kernel trace of the same command)")
    for k, c in sorted(ctr.items()):
        t = next((d for n, d in dur.items() if n[:60] == k[:60]), None)
        if "simon" not in k or not t or t < 2e-4:
            continue
        print(f"   {k[:100]}")
        print(f"      average duration {t * 1e3:.3f} ms")
        if "SQ_INSTS_VALU" in c:
            print(f"      VALU issue: {c['SQ_INSTS_VALU']:.4g} wave-instructions / {t * 1e3:.3f} ms = {c['SQ_INSTS_VALU'] / t / 1e9:.1f} G/s = {c['SQ_INSTS_VALU'] / t / 1228.8e9:.4f} of the 1 228.8 G/s peak"
                  + (f"; SALU {c['SQ_INSTS_SALU']:.4g}" if "SQ_INSTS_SALU" in c else ""))
        if "SQ_ACTIVE_INST_VALU" in c:
            # quad-cycles in which a wave has a VALU instruction executing, summed over the SIMDs (bench.py: valu_pipe_busy_frac)
            print(f"      VALU pipe busy: SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x {t * 1e3:.3f} ms x 2.4 GHz) = {c['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * t * 2.4e9):.3f}"
                  + (f"; {c['SQ_ACTIVE_INST_VALU'] * 4 / c['SQ_INSTS_VALU']:.2f} cycles per VALU instruction" if c.get("SQ_INSTS_VALU") else ""))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            gb = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9
            print(f"      HBM traffic: 2 x FETCH_SIZE {2 * c['FETCH_SIZE'] * 1024 / 1e9:.3f} GB + WRITE_SIZE {c['WRITE_SIZE'] * 1024 / 1e9:.3f} GB = {gb:.3f} GB -> {gb / t / 1e3:.3f} TB/s = {gb / t / 8e3:.4f} of 8 TB/s")
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print(f"      LDS: bank-conflict cycles / LDS-active cycles = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}"
                  + (f"; LDS instructions {c['SQ_INSTS_LDS']:.4g}" if "SQ_INSTS_LDS" in c else ""))
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            w = c["SQ_WAVE_CYCLES"]
            parts = [f"{n} {c[key] / w:.3f}" for n, key in (("parked (s_waitcnt / barrier)", "SQ_WAIT_ANY"), ("issue-stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY")) if key in c]
            print("      wave time: " + ", ".join(parts))
        if "SQ_INSTS_VMEM_WR" in c:
            print(f"      vector memory instructions: {c.get('SQ_INSTS_VMEM_RD', 0):.4g} loads, {c['SQ_INSTS_VMEM_WR']:.4g} stores")


if __name__ == "__main__":
    main(sys.argv[1])
    derived(sys.argv[1])
