"""Phase profile of generation 6's REST select on config 5 (a -DSIMON_TABLE_PROFILE build through SIMON_HIP_LIB, env SIMON_TABLE_PROF=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open_simulator_amd import capi, synth
n = int(sys.argv[1])
prob, scen, orders = synth.config5(n_scen=n, n_orders=4)
with capi.Context(0) as ctx:
    ctx.load_problem(prob); ctx.load_scenarios(scen, orders)
    ctx.run_loaded(True); ctx.run_loaded(True)
    print("kernel_ms", ctx.stats().kernel_ms, "lds", ctx.stats().lds_bytes)
