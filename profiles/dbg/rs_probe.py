"""Debug probe (round 6): generation 7's walks over generation 6's position-mask rows (REST && SPREAD, simon_table_rs.hip) against the oracle, route printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, randprob
from open_simulator_amd import capi, synth
os.environ["SIMON_DEBUG_ROUTE"] = "1"
bad = 0
def check(name, prob, scen, orders):
    global bad
    ref = O.run_threaded(prob, scen, orders)
    for env in ({}, {"SIMON_NO_RS": "1"}):
        for k, v in env.items(): os.environ[k] = v
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
        for k in env: os.environ.pop(k)
        ok = res.unscheduled.tolist() == ref.unscheduled.tolist() and bool((res.placement == ref.placement).all()) and res.used_cpu.tolist() == ref.used_cpu.tolist()
        bad += not ok
        print(f"{name} {env}: generation {st.kernel_generation} variant {st.kernel_variant} {st.kernel_ms:.1f} ms {'ok' if ok else 'MISMATCH'} "
              f"(unscheduled {ref.unscheduled.tolist()[:6]}, differing placements {int((res.placement != ref.placement).sum())})", flush=True)
check("c5svc small", *synth.config5(n_pods=1500, n_nodes=300, n_scen=4, n_orders=2, n_groups=10, group_size=20, services=True))
check("c5svc mid", *synth.config5(n_pods=8000, n_nodes=1200, n_scen=8, n_orders=2, n_groups=20, group_size=40, services=True))
os.environ["SIMON_NO_GPU_FOLD"] = "1"; os.environ["SIMON_NO_FOLD"] = "1"
check("200 services, 10 anti, 20 gpu", *synth.config_service(n_counts=4, n_pods=5000, n_services=200, n_anti=10, n_gpu=20))
for seed, feat in enumerate([dict(gpu=True), dict(anti_host=True), dict(gpu=True, anti_host=True, nz_differs=True), dict(gpu=True, presets=True, gates=True), dict(anti=True, tight_pods=True), dict(gpu=True, eph=True, scalars=2)]):
    prob = randprob.rand_problem(7300 + seed, N=[200, 500, 900, 300, 700, 400][seed], P=600, spread_soft=True, n_node_classes=4, n_pod_classes=[3, 30, 60, 8, 20, 5][seed], **feat)
    scen, orders = randprob.rand_scenarios(seed, prob, S=4)
    os.environ["SIMON_NO_GPU_FOLD"] = "1"; os.environ["SIMON_NO_FOLD"] = "1"
    check(f"rand {sorted(feat)}", prob, scen, orders)
sys.exit(1 if bad else 0)
