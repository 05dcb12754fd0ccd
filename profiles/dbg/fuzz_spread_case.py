"""Debug: one case of tests/fuzz_spread.py in detail -- which shape, which scenario, which step first differs from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_spread as F, oracle_lib as O, randprob
from open_simulator_amd import capi
case = int(sys.argv[1])
cap = {}
orig_run = O.run_threaded
def grab(prob, scen, orders, *a, **k):
    cap["p"] = (prob, scen, orders); return orig_run(prob, scen, orders, *a, **k)
O.run_threaded = grab
orig_run2 = O.run
def grab2(prob, scen, orders, *a, **k):
    cap["p"] = (prob, scen, orders); cap["ranks"] = k.get("node_ranks"); return orig_run2(prob, scen, orders, *a, **k)
O.run = grab2
ok, info = F.one_case(case)
print(ok, info)
prob, scen, orders = cap["p"]
ranks = cap.get("ranks")
ref = orig_run2(prob, scen, orders, node_ranks=ranks) if ranks is not None else orig_run(prob, scen, orders)
os.environ["SIMON_DEBUG_ROUTE"] = "1"
for env in ({"SIMON_TEAM": "0"}, {"SIMON_TEAM": "1"}, {"SIMON_NO_RS": "1"}):
    env = dict(env)
    if case >= 700000: env.update(SIMON_NO_FOLD="1", SIMON_NO_GPU_FOLD="1")
    os.environ.update(env)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob); ctx.load_scenarios(scen, orders)
        if ranks is not None: ctx.set_node_ranks(ranks)
        ctx.run_loaded(True); res = ctx.fetch(True); st = ctx.stats()
    for k in env: os.environ.pop(k)
    bad = np.argwhere(res.placement != ref.placement)
    print(env, "generation", st.kernel_generation, "wg", st.workgroup_size, "differing", len(bad), "unsched", res.unscheduled.tolist(), ref.unscheduled.tolist())
    if len(bad):
        s = bad[0][0]
        order = orders[scen[s][1]]
        pos = {int(p): i for i, p in enumerate(order.tolist())}
        first = min(bad[bad[:, 0] == s][:, 1], key=lambda p: pos[int(p)])
        print("  scenario", s, scen[s].tolist(), "first differing pod", int(first), "at step", pos[int(first)], "class", int(prob.pod_class[first]), "gpu", int(prob.gpu_mem[first]) if prob.gpu_mem is not None else None,
              "got", int(res.placement[s, first]), "want", int(ref.placement[s, first]))
import copy
def run_team(p, label):
    env = dict(SIMON_TEAM="1", SIMON_NO_FOLD="1", SIMON_NO_GPU_FOLD="1")
    os.environ.update(env)
    r0 = orig_run(p, scen, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(p); res = ctx.run_batch(scen, orders); st = ctx.stats()
    for k in env: os.environ.pop(k)
    out = [int((res.placement != r0.placement).sum())]
    os.environ.update(env)
    for _ in range(7):
        with capi.Context(0) as ctx:
            ctx.load_problem(p); res = ctx.run_batch(scen, orders)
        out.append(int((res.placement != r0.placement).sum()))
    for k in env: os.environ.pop(k)
    print(label, "generation", st.kernel_generation, "wg", st.workgroup_size, "differing over 8 runs", out)
if len(sys.argv) > 2:
    run_team(prob, "as drawn")
    p = copy.deepcopy(prob); p.gpu_mem = np.zeros_like(p.gpu_mem); p.pod_gpu_cnt = np.zeros_like(p.pod_gpu_cnt); run_team(p.normalise(), "no gpu requests")
    p = copy.deepcopy(prob); p.pref_off = p.pref_idx = p.pref_w = p.own_off = p.own_idx = p.own_w = None; run_team(p.normalise(), "no preferred terms")
    p = copy.deepcopy(prob); p.anti_off = np.zeros_like(p.anti_off); run_team(p.normalise(), "no anti-affinity")
    p = copy.deepcopy(prob); p.req_cpu = np.maximum(p.req_cpu, 100); p.req_mem = np.maximum(p.req_mem, 1 << 20); p.nz_cpu = p.nz_mem = None; run_team(p.normalise(), "no zero pods")
