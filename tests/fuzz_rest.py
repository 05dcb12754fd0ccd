#!/usr/bin/env python3
"""Randomised parity sweep aimed at generation 6 of the score-table kernel (simon_table.hip, REST path): cpu+memory problems with
Open-Gpu-Share devices and / or required anti-affinity on node-level topology keys, ephemeral storage, extended resources, 1 ... 8 191 nodes, 1 ... 64 internal node
classes, up to 120 pod classes (term classes, table classes), presets (bound without Reserve), gates, pinned pods, static masks,
initial node and device state, zero requests, tight pod counts.  Not collected by pytest (a slice runs in
tests/test_gpu_round2.py); by hand on a GPU box:   python tests/fuzz_rest.py [n_cases] [first_seed]
Every placement, unscheduled count and used cpu / memory is compared with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402,F401
import oracle_lib as O  # noqa: E402
import randprob  # noqa: E402
from open_simulator_amd import capi  # noqa: E402

FEATURES = ["nz_differs", "init_state", "static_mask", "presets", "gates", "zero_pods", "tight_pods", "pins", "static_small"]


def one_case(case):
    rng = np.random.default_rng(61000 + case)
    size = case % 4
    N = int(rng.integers(1, 80)) if size == 0 else int(rng.integers(100, 1100)) if size == 1 else int(rng.integers(1100, 4200)) if size == 2 \
        else int(rng.integers(4200, 8192))
    P = int(rng.integers(20, 400 if size == 0 else 1800))
    feat = {f: True for f in FEATURES if rng.random() < 0.3}
    kind = case % 3
    if kind != 1:
        feat["gpu"] = True
    if kind != 0:
        feat["anti" if case % 4 == 1 else "anti_host"] = True          # anti: terms on a zone-like key too (4 zones, unlabeled nodes)
    if rng.random() < 0.3:                # ephemeral storage / extended resources (on generation 6 unless presets or initial
        feat["eph"] = True                # state over-commit a node: then the all-feature kernel takes the case)
    if rng.random() < 0.3:
        feat["scalars"] = int(rng.integers(1, 5))
    if rng.random() < 0.25:
        feat["ports"] = True              # NodePorts: port terms on the hostname key
    if "anti" in feat and rng.random() < 0.5:
        feat["aff"] = True                # required affinity (derived terms, first-pod escape) beside the anti-affinity terms
    if size >= 2:                         # static masks are O(Cp N) Python work in the generator
        feat.pop("static_mask", None)
    if 800000 <= case < 900000 or (case >= 900000 and case % 2):   # cases from 800 000 on (every other one from 900 000 on): the same filters next to soft PodTopologySpread constraints -- generation 7's walks over
        feat["spread_soft"] = True        # these mask rows (REST && SPREAD, simon_table_rs.hip), required affinity included
    n_node_classes = int(rng.choice([70, 90, 110] if 500000 <= case < 800000 else [1, 2, 4, 9, 20, 40]))   # cases from 500 000 on: 65 .. 128 internal node classes (CN2 in rest_select)
    n_pod_classes = int(rng.choice([130, 200, 300, 500] if case >= 900000 else [1, 3, 8, 30, 64, 120]))   # cases from 900 000 on: 129 .. 1 023 signatures under the REST rows (MANY && REST; with spread_soft: && SPREAD)
    prob = randprob.rand_problem(62000 + case, N=N, P=P, n_node_classes=n_node_classes, n_pod_classes=n_pod_classes, **feat)
    S = int(rng.integers(1, 7))
    scen, orders = randprob.rand_scenarios(case, prob, S=S, min_n=1 if rng.random() < 0.5 else None)
    ranks = None
    if case % 5 == 4:                     # per-scenario node order (simon_set_node_ranks)
        ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
        for s, (n, _) in enumerate(np.asarray(scen).tolist()):
            ranks[s, :n] = rng.permutation(n)
    ref = O.run_threaded(prob, scen, orders) if ranks is None else O.run(prob, scen, orders, node_ranks=ranks)
    # every other case with the GPU fold switched off: the position masks of generation 6 keep their coverage, the fold (GPU share as
    # monotone infeasibility of the table, few signatures) gets the other half
    if case % 2:
        os.environ["SIMON_NO_GPU_FOLD"] = "1"
    else:
        os.environ.pop("SIMON_NO_GPU_FOLD", None)
    if (case // 2) % 2:                   # round 5: every other pair of cases keeps the mask rows in HBM (the others: in LDS where the batch is resident)
        os.environ["SIMON_LDS_WS"] = "0"
    else:
        os.environ.pop("SIMON_LDS_WS", None)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        if ranks is not None:
            ctx.set_node_ranks(ranks)
        ctx.run_loaded(True)
        res = ctx.fetch(True)
        st = ctx.stats()
    os.environ.pop("SIMON_NO_GPU_FOLD", None)
    os.environ.pop("SIMON_LDS_WS", None)
    ok = (res.unscheduled.tolist() == ref.unscheduled.tolist() and res.used_cpu.tolist() == ref.used_cpu.tolist() and
          res.used_mem.tolist() == ref.used_mem.tolist() and (res.placement == ref.placement).all())
    return ok, dict(case=case, N=N, P=P, S=S, classes=n_node_classes, pod_classes=n_pod_classes, feat=sorted(feat),
                    variant=st.kernel_variant, generation=st.kernel_generation, unscheduled=int(ref.unscheduled.sum()))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, on_rest, on_walks, aff_walks = 0, 0, 0, 0
    for case in range(first, first + n_cases):
        ok, info = one_case(case)
        on_rest += info["generation"] == 6
        on_walks += info["generation"] == 7
        aff_walks += info["generation"] == 7 and "aff" in info["feat"]
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
    print(f"fuzz_rest: {n_cases} cases from {first}, {on_rest} on generation 6, {on_walks} on generation 7 ({aff_walks} of them with required affinity), mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
