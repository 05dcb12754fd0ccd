#!/usr/bin/env python3
"""Round 5: generation 6 with its mask rows in LDS (SIMON_LDS_WS) against the rows in HBM on config 5, same process, same box.
usage: python profiles/ldsx_probe.py [scenarios=256] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_simulator_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prob, scen, orders = synth.config5(n_scen=n, n_orders=4)
out = {}
for mode in ("0", "1"):
    os.environ["SIMON_LDS_WS"] = mode
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ts = []
        for _ in range(reps + 1):
            ctx.run_loaded(True)
            ts.append(ctx.stats().kernel_ms)
        st = ctx.stats()
        out[mode] = ctx.fetch(True)
    print(f"config5 S={len(scen)} SIMON_LDS_WS={mode}: generation {st.kernel_generation} lds {st.lds_bytes} kernel_ms {min(ts[1:]):.3f} (runs {['%.2f' % t for t in ts]})", flush=True)
same = (out["0"].placement == out["1"].placement).all() and out["0"].unscheduled.tolist() == out["1"].unscheduled.tolist()
print("both homes of the mask rows agree on every placement:", bool(same))
