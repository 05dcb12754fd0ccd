"""Debug: the FIRST team-mode run of a process on one fuzz_spread case (optionally a variant of it), differing placements against the oracle."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_spread as F, oracle_lib as O
from open_simulator_amd import capi
case, variant = int(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else "asdrawn"
cap = {}
orig = O.run_threaded
def grab(prob, scen, orders, *a, **k):
    cap["p"] = (prob, scen, orders); raise KeyboardInterrupt
O.run_threaded = grab
orig2 = O.run
def grab2(prob, scen, orders, *a, **k):
    cap["p"] = (prob, scen, orders); cap["ranks"] = k.get("node_ranks"); raise KeyboardInterrupt
O.run = grab2
try: F.one_case(case)
except KeyboardInterrupt: pass
O.run_threaded = orig; O.run = orig2
ranks = cap.get("ranks")
if variant == "noranks": ranks = None
prob, scen, orders = cap["p"]
p = copy.deepcopy(prob)
if variant == "nogpu": p.gpu_mem = np.zeros_like(p.gpu_mem); p.pod_gpu_cnt = np.zeros_like(p.pod_gpu_cnt)
if variant == "noanti": p.anti_off = np.zeros_like(p.anti_off)
if variant == "nozero": p.req_cpu = np.maximum(p.req_cpu, 100); p.req_mem = np.maximum(p.req_mem, 1 << 20); p.nz_cpu = p.nz_mem = None
p = p.normalise()
ref = orig2(p, scen, orders, node_ranks=ranks) if ranks is not None else orig(p, scen, orders)
os.environ.update(SIMON_TEAM=os.environ.get("SIMON_TEAM", "1"), SIMON_NO_FOLD="1", SIMON_NO_GPU_FOLD="1")
out = []
for _ in range(3):
    with capi.Context(0) as ctx:
        ctx.load_problem(p); ctx.load_scenarios(scen, orders)
        if ranks is not None: ctx.set_node_ranks(ranks)
        ctx.run_loaded(True); res = ctx.fetch(True); st = ctx.stats()
    out.append(int((res.placement != ref.placement).sum()))
print(case, variant, "generation", st.kernel_generation, "wg", st.workgroup_size, "differing in runs 1..3:", out, flush=True)
