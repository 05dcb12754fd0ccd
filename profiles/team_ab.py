#!/usr/bin/env python3
"""Same-box A/B of the two shapes of generation 7 (simon_table.hip): one wave per scenario (SIMON_TEAM=0) against a team of
4 waves (SIMON_TEAM=4), and the width the library picks (unset), over the batch size -- kernel milliseconds (HIP events, best of 3 after a warm-up).  Workloads:
config 3's pool with every pod behind a Service (synth.config_service; optionally with preferred self anti-affinity), at
S = 4 x counts scenarios.  Needs a GPU:  python profiles/team_ab.py [--pref 60] [--sizes 4,16,64,128,256,512]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_simulator_amd import capi, synth  # noqa: E402


def kernel_ms(prob, scen, orders, team):
    if team is not None:
        os.environ["SIMON_TEAM"] = team
    try:
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ms = []
            for _ in range(4):
                ctx.run_loaded(True)
                ms.append(ctx.stats().kernel_ms)
            st = ctx.stats()
            res = ctx.fetch(False)
        return min(ms[1:]), st.workgroup_size, st.kernel_generation, res
    finally:
        os.environ.pop("SIMON_TEAM", None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pref", type=int, default=0)
    ap.add_argument("--sizes", default="4,16,64,128,256,512,1024")
    ap.add_argument("--only", default=None, help="one shape only (SIMON_TEAM value): profile builds")
    a = ap.parse_args()
    for counts in [int(x) for x in a.sizes.split(",")]:
        prob, scen, orders = synth.config_service(n_counts=counts, n_orders=4, n_pref=a.pref)
        if counts < 4:
            scen = scen[:counts]
        if a.only is not None:
            ms, wg, g, _ = kernel_ms(prob, scen, orders, a.only)
            print(json.dumps({"scenarios": len(scen), "team": a.only, "kernel_ms": round(ms, 3), "workgroup": wg, "generation": g}), flush=True)
            continue
        one, wg1, g1, r1 = kernel_ms(prob, scen, orders, "0")
        row = {"workload": f"config3_service{'_pref%d' % a.pref if a.pref else ''}", "scenarios": len(scen), "pods": int(prob.n_pods), "one_wave_ms": round(one, 3)}
        same = True
        for w in ("4",):
            if len(scen) * int(w) > 16384:              # (more waves than the chip holds at once: not a shape anyone would pick)
                continue
            ms, wg, g, r = kernel_ms(prob, scen, orders, w)
            row[f"team{w}_ms"] = round(ms, 3)
            same = same and wg == 64 * int(w) and g == g1 and bool((r1.unscheduled == r.unscheduled).all() and (r1.used_cpu == r.used_cpu).all())
        ms, wg, g, r = kernel_ms(prob, scen, orders, None)
        row.update({"auto_ms": round(ms, 3), "auto_workgroup": wg, "speedup_auto": round(one / ms, 3), "same_counts": same})
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
