#!/usr/bin/env python3
"""Round 5: generation 4 with the scenario's workspace in LDS (SIMON_LDS_WS) against the workspace in HBM, same process, same box.
usage: python profiles/ldsws_probe.py config3 16 | config2 1 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_simulator_amd import capi, synth

wl, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
prob, scen, orders = synth.config2() if wl == "config2" else synth.config3(n_counts=n)
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
    print(f"{wl} S={len(scen)} SIMON_LDS_WS={mode}: generation {st.kernel_generation} lds {st.lds_bytes} kernel_ms {min(ts[1:]):.3f} (runs {['%.3f' % t for t in ts]})", flush=True)
same = (out["0"].placement == out["1"].placement).all() and out["0"].unscheduled.tolist() == out["1"].unscheduled.tolist() and out["0"].used_cpu.tolist() == out["1"].used_cpu.tolist()
print("both workspaces agree on every placement:", bool(same))
