import sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
import fuzz_spread, oracle_lib as O, randprob
from open_simulator_amd import capi
case = int(sys.argv[1])
rng = np.random.default_rng(88000 + case)
size = case % 3
N = int(rng.integers(2, 90)) if size == 0 else int(rng.integers(100, 900)) if size == 1 else int(rng.integers(900, 3000))
P = int(rng.integers(20, 500 if size == 0 else 2500))
feat = {f: True for f in fuzz_spread.FEATURES if rng.random() < 0.3}
if size == 2: feat.pop("static_mask", None)
prob = randprob.rand_problem(99000 + case, N=N, P=P, spread_soft=True, n_node_classes=int(rng.choice([1, 2, 4, 9])), n_pod_classes=int(rng.choice([1, 3, 8, 30, 60])), **feat)
scen, orders = randprob.rand_scenarios(case, prob, S=int(rng.integers(1, 8)), min_n=1 if rng.random() < 0.4 else None)
ref = O.run_threaded(prob, scen, orders)
with capi.Context(0) as ctx:
    ctx.load_problem(prob); res = ctx.run_batch(scen, orders); st = ctx.stats()
print("N", N, "P", P, "S", len(scen), feat, "gen", st.kernel_generation)
print("zones", prob.topo_dom[1][:min(N, 40)].tolist(), "term_set", prob.term_node_set.tolist())
for s in range(len(scen)):
    bad = np.flatnonzero(res.placement[s] != ref.placement[s])
    if len(bad) == 0: continue
    n, o = scen[s]
    order = orders[o]
    inv = np.empty(P, int); inv[order] = np.arange(P)
    first = bad[np.argmin(inv[bad])]; step = inv[first]
    c = int(prob.pod_class[first])
    soft = prob.spread_soft_idx[prob.spread_soft_off[c]:prob.spread_soft_off[c + 1]].tolist()
    skew = prob.spread_soft_skew[prob.spread_soft_off[c]:prob.spread_soft_off[c + 1]].tolist()
    match = prob.match_idx[prob.match_off[c]:prob.match_off[c + 1]].tolist()
    print("scenario", s, "n", n, "mismatching pods", len(bad), "first at step", step, "pod", first, "class", c, "soft", soft, skew, "match", match,
          "preset", None if prob.preset_node is None else prob.preset_node[first], "pin", None if prob.pin_node is None else prob.pin_node[first],
          "gpu", res.placement[s][first], "oracle", ref.placement[s][first])
    best, sc = O.score_pod_after(prob, int(n), int(step), int(first), order)
    g, r = int(res.placement[s][first]), int(ref.placement[s][first])
    for name, j in (("gpu", g), ("oracle", r)):
        if j >= 0: print("  ", name, "node", j, "zone", prob.topo_dom[1][j], "class", prob.node_class[j], {k: int(v[j]) for k, v in sc.items() if k != "codes"})
    tot = sc["total"]; feas = sc["feasible"]
    top = tot[feas > 0].max() if (feas > 0).any() else None
    print("   top total", top, "nodes at top", np.flatnonzero((tot == top) & (feas > 0))[:10], "feasible", int((feas > 0).sum()))
    # counts of the soft terms on both nodes after `step` placements by the oracle
    placed = ref.placement[s]
    for t in soft:
        key = prob.term_topo_key[t]; dom = prob.topo_dom[key]
        cnt = {}
        for q in order[:step]:
            j = placed[q]
            if j < 0: continue
            cq = int(prob.pod_class[q])
            m = prob.match_idx[prob.match_off[cq]:prob.match_off[cq + 1]].tolist().count(t)
            if not m: continue
            ts = prob.term_node_set[t]
            if ts >= 0 and not (int(prob.node_sets[ts][j // 64]) >> (j % 64)) & 1: continue
            if dom[j] >= 0: cnt[int(dom[j])] = cnt.get(int(dom[j]), 0) + m
        print("   term", t, "key", key, "set", prob.term_node_set[t], "count at gpu node dom", cnt.get(int(dom[g]), 0) if g >= 0 else None, "oracle node dom", cnt.get(int(dom[r]), 0) if r >= 0 else None, "n doms", len(cnt))
    break
