"""Which route a problem takes (SIMON_DEBUG_ROUTE=1): the MANY + SPREAD cases of tests/test_gpu_round4.py that left generation 7."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SIMON_DEBUG_ROUTE"] = "1"
import numpy as np
import randprob
from open_simulator_amd import capi
from test_gpu_round4 import MANY_SPREAD_FEATURES
for idx in (3, 4):
    for seed, (n_pc, N, P) in enumerate([(130, 60, 700), (200, 500, 1800), (384, 900, 2600)]):
        prob = randprob.rand_problem(12000 + 10 * idx + seed, N=N, P=P, spread_soft=True, n_node_classes=3, n_pod_classes=n_pc, **MANY_SPREAD_FEATURES[idx])
        scen, orders = randprob.rand_scenarios(120 + seed, prob, S=4)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.run_batch(scen, orders)
            st = ctx.stats()
        print("case", idx, seed, n_pc, "variant", st.kernel_variant, "generation", st.kernel_generation, flush=True)
