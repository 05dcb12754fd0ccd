"""Debug probe: the all-feature kernel at several workgroup sizes on V2_FEATURES[idx] -- placements and the end-of-run sums (used cpu /
memory / volume groups) against the oracle, for the product library and, if present, libsimon_hip_lanes.so (-DSIMON_WG_REDUCE_LANES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, randprob
from open_simulator_amd import capi
from test_gpu_parity import V2_FEATURES
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
feat = V2_FEATURES[idx]
name = "lanes" if "lanes" in os.environ.get("SIMON_HIP_LIB", "") else "product"
for seed in range(3):
    N, P = 50 + 41 * seed, 350
    prob = randprob.rand_problem(3000 + 100 * idx + seed, N=N, P=P, **feat)
    scen, orders = randprob.rand_scenarios(seed, prob, S=6)
    ref = O.run(prob, scen, orders)
    if True:
        for wg in ("256", "512", "1024"):
            os.environ["SIMON_WG"] = wg
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                res = ctx.run_batch(scen, orders)
            print(f"seed {seed} N={N} {name} wg={wg}: placements {'ok' if (res.placement == ref.placement).all() else 'DIFF'} "
                  f"cpu {'ok' if res.used_cpu.tolist() == ref.used_cpu.tolist() else 'DIFF'} mem {'ok' if res.used_mem.tolist() == ref.used_mem.tolist() else 'DIFF'} "
                  f"vg {'ok' if res.used_vg.tolist() == ref.used_vg.tolist() else 'DIFF ' + str(res.used_vg.tolist()) + ' vs ' + str(ref.used_vg.tolist())}", flush=True)
