import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
import fuzz_table, oracle_lib as O, randprob
from open_simulator_amd import capi
case = int(sys.argv[1])
# re-create the case exactly as fuzz_table.one_case does
import types
src = open('/root/repo/tests/fuzz_table.py').read()
rng = np.random.default_rng(41000 + case)
size = case % 4
N = int(rng.integers(1, 80)) if size == 0 else int(rng.integers(100, 1100)) if size == 1 else int(rng.integers(1100, 2100)) if size == 2 else int(rng.integers(2100, 4096))
P = int(rng.integers(20, 400 if size == 0 else 2500))
feat = {f: True for f in fuzz_table.FEATURES if rng.random() < 0.3}
n_node_classes = int(rng.choice([1, 2, 4, 9, 20, 40])); n_pod_classes = int(rng.choice([1, 3, 8, 30, 64, 65, 100, 128]))
if size == 3: feat.pop("static_mask", None)
prob = randprob.rand_problem(52000 + case, N=N, P=P, n_node_classes=n_node_classes, n_pod_classes=n_pod_classes, **feat)
if rng.random() < 0.3:
    print("refined shapes case")
    shapes_c = np.array([4000, 8000, 16000, 32000]) + (rng.integers(0, 5, 4) if "odd_units" in feat else 0)
    shapes_m = (np.array([8, 16, 64, 128]) << 30) + (rng.integers(0, 7, 4) if "odd_units" in feat else 0)
    pick = rng.integers(0, 4, N)
    if n_node_classes * 4 > 64: pick = prob.node_class % 3
    prob.alloc_cpu = shapes_c[pick].astype(np.int64); prob.alloc_mem = shapes_m[pick].astype(np.int64)
    if prob.init_req_cpu is not None:
        prob.init_req_cpu = np.minimum(prob.init_req_cpu, prob.alloc_cpu // 2); prob.init_req_mem = np.minimum(prob.init_req_mem, prob.alloc_mem // 2)
        prob.init_nz_cpu = np.maximum(prob.init_nz_cpu, prob.init_req_cpu); prob.init_nz_mem = np.maximum(prob.init_nz_mem, prob.init_req_mem)
S = int(rng.integers(1, 9))
scen, orders = randprob.rand_scenarios(case, prob, S=S, min_n=1 if rng.random() < 0.5 else None)
ref = O.run_threaded(prob, scen, orders)
with capi.Context(0) as ctx:
    ctx.load_problem(prob); ctx.load_scenarios(scen, orders); ctx.run_loaded(True); res = ctx.fetch(True)
print("N", N, "P", P, "S", S, feat, "classes", n_node_classes, n_pod_classes)
for s in range(len(scen)):
    bad = np.flatnonzero(res.placement[s] != ref.placement[s])
    if len(bad) == 0: continue
    n, o = scen[s]
    order = orders[o]
    inv = np.empty(P, int); inv[order] = np.arange(P)
    first = bad[np.argmin(inv[bad])]
    step = inv[first]
    print("scenario", s, "n", n, "order", o, "mismatching pods", len(bad), "first at step", step, "pod", first, "gpu", res.placement[s][first], "oracle", ref.placement[s][first],
          "unsched gpu/ref", res.unscheduled[s], ref.unscheduled[s])
    best, sc = O.score_pod_after(prob, int(n), int(step), int(first), order)
    g, r = int(res.placement[s][first]), int(ref.placement[s][first])
    for name, j in (("gpu", g), ("oracle", r)):
        if j >= 0: print("  ", name, "node", j, "class", prob.node_class[j], "alloc", prob.alloc_cpu[j], prob.alloc_mem[j], {k: int(v[j]) for k, v in sc.items() if k != "codes"})
    tot = sc["total"]; feas = sc["feasible"]
    top = tot[feas > 0].max() if (feas > 0).any() else None
    # class-major layout of this scenario, as the kernel builds it
    import collections
    ncls_t = {}
    key_of = {}
    for j in range(prob.n_nodes):
        k = (int(prob.node_class[j]), int(prob.alloc_cpu[j]), int(prob.alloc_mem[j]))
        if k not in key_of: key_of[k] = len(key_of)
        ncls_t[j] = key_of[k]
    cnt = collections.Counter(ncls_t[j] for j in range(int(n)))
    seg, off = {}, 0
    for d in range(len(key_of)):
        seg[d] = off; off += (cnt.get(d, 0) + 15) // 16 * 16
    pos_of = {}
    rank = collections.Counter()
    for j in range(int(n)):
        d = ncls_t[j]; pos_of[j] = seg[d] + rank[d]; rank[d] += 1
    print("   internal classes", len(key_of), "ni", off, "gpu node pos", pos_of.get(g), "class", ncls_t.get(g), "oracle node pos", pos_of.get(r), "class", ncls_t.get(r))
    print("   top total", top, "nodes at top", np.flatnonzero((tot == top) & (feas > 0))[:10], "their classes", prob.node_class[np.flatnonzero((tot == top) & (feas > 0))[:10]])
    break
