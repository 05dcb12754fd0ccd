#!/usr/bin/env python3
"""Where a fresh context's batch time goes (bench.py's end_to_end leg, call by call): python profiles/e2e_phases.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_simulator_amd import capi, synth
prob, scen, orders = synth.config3(n_counts=1024, n_orders=4, n_pods=10000, seed=synth.SEED + 3)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    with capi.Context(0) as ctx:
        t = [time.perf_counter()]
        ctx.load_problem(prob); t.append(time.perf_counter())
        ctx.load_scenarios(scen, orders); t.append(time.perf_counter())
        ctx.run_loaded(want_placement=True); t.append(time.perf_counter())
        res = ctx.fetch(want_placement=False); t.append(time.perf_counter())
        plan = ctx.min_plan(); t.append(time.perf_counter())
        row = ctx.fetch_placement(plan.scenario if plan.found else 0); t.append(time.perf_counter())
        st = ctx.stats()
    names = ["load_problem", "load_scenarios", "run_loaded", "fetch", "min_plan", "fetch_placement"]
    print(f"rep {rep}: " + " | ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t, t[1:])) + f" | kernel {st.kernel_ms:.2f} ms", flush=True)
