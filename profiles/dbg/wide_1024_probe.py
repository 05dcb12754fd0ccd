"""Where does the all-feature kernel at 1 024 threads first differ from the oracle on V2_FEATURES[idx]?  (debug probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, randprob
from open_simulator_amd import capi
from test_gpu_parity import V2_FEATURES, run_gpu
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
feat = V2_FEATURES[idx]
for seed in range(3):
    N, P = 50 + 41 * seed, 350
    prob = randprob.rand_problem(3000 + 100 * idx + seed, N=N, P=P, **feat)
    scen, orders = randprob.rand_scenarios(seed, prob, S=6)
    ref = O.run(prob, scen, orders)
    for wg in ("128", "256", "512", "1024"):
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_WG": wg})
        bad = np.argwhere(res.placement != ref.placement)
        msg = "ok"
        if len(bad):
            s, _ = bad[0]
            order = orders[scen[s, 1]]
            first = next(i for i, pid in enumerate(order) if res.placement[s, pid] != ref.placement[s, pid])
            pid = int(order[first])
            msg = (f"{len(bad)} differ; scenario {s} (n={scen[s,0]}) step {first} pod {pid} class {prob.pod_class[pid]}: gpu {res.placement[s,pid]} oracle {ref.placement[s,pid]}; "
                   f"local spec {None if prob.local_spec_of is None else prob.local_spec_of[prob.pod_class[pid]]}")
        print(f"N={N} P={P} wg={wg}: {msg}", flush=True)
