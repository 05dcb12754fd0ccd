"""Debug probe (round 6): generation 7 with 65 .. 128 internal node classes (CN2) -- a few hand-made cases against the oracle, with the route printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, randprob
from open_simulator_amd import capi
os.environ["SIMON_DEBUG_ROUTE"] = "1"
bad = 0
for seed, (N, P, ncls, feat) in enumerate([(400, 600, 20, {}), (900, 1500, 25, dict(nz_differs=True)), (2500, 2000, 22, dict(gates=True, presets=True)),
                                           (700, 900, 18, dict(static_small=True, tight_pods=True)), (5000, 1500, 24, dict(pins=True)), (300, 500, 30, {})]):
    prob = randprob.rand_problem(4242 + seed, N=N, P=P, spread_soft=True, n_node_classes=ncls, n_pod_classes=[3, 30, 60, 8, 100, 5][seed], **feat)
    scen, orders = randprob.rand_scenarios(seed, prob, S=4)
    ref = O.run_threaded(prob, scen, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders)
        st = ctx.stats()
    ok = res.unscheduled.tolist() == ref.unscheduled.tolist() and bool((res.placement == ref.placement).all()) and res.used_cpu.tolist() == ref.used_cpu.tolist()
    bad += not ok
    print(f"seed {seed} N={N} P={P} node classes {ncls}: generation {st.kernel_generation} variant {st.kernel_variant} {'ok' if ok else 'MISMATCH'} "
          f"(unscheduled {ref.unscheduled.tolist()}, differing placements {int((res.placement != ref.placement).sum())})", flush=True)
sys.exit(1 if bad else 0)
