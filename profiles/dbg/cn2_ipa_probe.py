"""Debug probe (round 6): generation 7 with 65 .. 128 internal node classes AND preferred / hard terms in the walk (CN2 && AFF), route printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, randprob
from open_simulator_amd import capi
os.environ["SIMON_DEBUG_ROUTE"] = "1"
bad = 0
cases = [(400, 600, 20, dict(ipa_self=True)), (900, 1200, 25, dict(hard_simple=True)), (1500, 1500, 22, dict(ipa_self=True, hard_simple=True, nz_differs=True)),
         (700, 900, 18, dict(ipa_self=True, static_small=True, tight_pods=True)), (3000, 1500, 24, dict(hard_simple=True, pins=True, gates=True)), (600, 800, 20, dict(ipa_self=True, anti_host=True))]
for seed, (N, P, ncls, feat) in enumerate(cases):
    prob = randprob.rand_problem(4300 + seed, N=N, P=P, spread_soft=True, n_node_classes=ncls, n_pod_classes=[3, 30, 60, 8, 100, 5][seed], **feat)
    scen, orders = randprob.rand_scenarios(seed, prob, S=4, min_n=N // 2)
    ref = O.run_threaded(prob, scen, orders)
    for team in ("0", "1"):
        os.environ["SIMON_TEAM"] = team
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
        ok = res.unscheduled.tolist() == ref.unscheduled.tolist() and bool((res.placement == ref.placement).all()) and res.used_cpu.tolist() == ref.used_cpu.tolist()
        bad += not ok
        print(f"seed {seed} N={N} {sorted(feat)} team={team}: generation {st.kernel_generation} wg {st.workgroup_size} {'ok' if ok else 'MISMATCH'} "
              f"(unscheduled {ref.unscheduled.tolist()}, differing placements {int((res.placement != ref.placement).sum())})", flush=True)
sys.exit(1 if bad else 0)
