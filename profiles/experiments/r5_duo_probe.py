#!/usr/bin/env python3
"""Round 5: the leader / refresher pair (SIMON_DUO) against one wave per scenario on a workload, same process, same box.
usage: python profiles/duo_probe.py config5 256 | config3 16 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open_simulator_amd import capi, synth

wl, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if wl == "config5":
    prob, scen, orders = synth.config5(n_scen=n, n_orders=4)
else:
    prob, scen, orders = synth.config3(n_counts=n)
out = {}
for duo in ("0", "1"):
    os.environ["SIMON_DUO"] = duo
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ts = []
        for _ in range(reps + 1):
            ctx.run_loaded(True)
            ts.append(ctx.stats().kernel_ms)
        st = ctx.stats()
        res = ctx.fetch(True)
    out[duo] = res
    print(f"{wl} S={len(scen)} SIMON_DUO={duo}: workgroup {st.workgroup_size} generation {st.kernel_generation} lds {st.lds_bytes} kernel_ms {min(ts[1:]):.3f} (runs {['%.2f' % t for t in ts]})", flush=True)
same = (out["0"].placement == out["1"].placement).all() and out["0"].unscheduled.tolist() == out["1"].unscheduled.tolist()
print("both shapes agree on every placement:", bool(same))
