#!/usr/bin/env python3
"""BASELINE config 5 as it is drawn with its anti-affinity groups behind Services (synth.config5(services=True)), EVERY scenario of the batch against the
oracle, placement by placement (VERDICT r5 next-2's bar: all 256 scenarios checked).  usage: python profiles/check_c5asdrawn.py [n_scen=256]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from open_simulator_amd import capi, synth
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob, scen, orders = synth.config5(n_scen=n_scen, n_orders=4, services=True)
with capi.Context(0) as ctx:
    ctx.load_problem(prob)
    ctx.load_scenarios(scen, orders)
    ctx.run_loaded(True); ctx.run_loaded(True)
    st = ctx.stats()
    res = ctx.fetch(True)
t0 = time.perf_counter()
bad = rows = 0
for s0 in range(0, len(scen), 32):
    ref = O.run_threaded(prob, scen[s0:s0 + 32], orders)
    rows += len(ref.unscheduled)
    bad += int((ref.unscheduled != res.unscheduled[s0:s0 + 32]).sum()) + int((ref.placement != res.placement[s0:s0 + 32]).any(axis=1).sum()) \
        + int((ref.used_cpu != res.used_cpu[s0:s0 + 32]).sum())
dt = time.perf_counter() - t0
print(f"config5_asdrawn_service_S{len(scen)}: generation {st.kernel_generation} variant {st.kernel_variant} kernel {st.kernel_ms:.1f} ms; {rows} scenarios x {prob.n_pods} pods "
      f"checked against the oracle ({dt:.0f} s on the host), mismatching scenarios {bad}, unscheduled pods in total {int(res.unscheduled.sum())}")
sys.exit(1 if bad else 0)
