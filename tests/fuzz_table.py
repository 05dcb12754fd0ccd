#!/usr/bin/env python3
"""Randomised parity sweep aimed at the score-table kernel (simon_table.hip): cpu+memory problems at the sizes and shapes its
templates switch on -- 1 ... 4 095 nodes (1, 2 or 4 blocks per lane), 1 ... 384 request signatures (one or two per lane in registers, further
groups of 128 from memory), 1 ...
128 internal node classes (cases from 300 000 on: 40 ... 128) incl. caller classes that do NOT share their allocatable (the kernel refines them), presets, gates,
pinned pods, static masks, initial state, NonZeroRequested != Requested, zero requests, tight pod counts, gcd-1 units.  Every
third case forces the two-level summary (SIMON_TABLE_COARSE=1: classes padded to 64, per-16 entries in HBM).
Not collected by pytest (a slice runs in tests/test_gpu_round2.py); by hand on a GPU box:
    python tests/fuzz_table.py [n_cases] [first_seed]
Every placement, unscheduled count and used cpu / memory is compared with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402,F401
import oracle_lib as O  # noqa: E402
import randprob  # noqa: E402
from open_simulator_amd import capi  # noqa: E402

FEATURES = ["nz_differs", "init_state", "static_mask", "presets", "gates", "zero_pods", "tight_pods", "pins", "odd_units", "static_small", "anti_host", "ports"]


def one_case(case, coarse=None):
    coarse = (case % 3 == 2) if coarse is None else coarse
    rng = np.random.default_rng(41000 + case)
    size = case % 4
    N = int(rng.integers(1, 80)) if size == 0 else int(rng.integers(100, 1100)) if size == 1 else int(rng.integers(1100, 2100)) if size == 2 \
        else int(rng.integers(2100, 4096))
    P = int(rng.integers(20, 400 if size == 0 else 2500))
    feat = {f: True for f in FEATURES if rng.random() < 0.3}
    wide = case >= 300000                 # cases from 300 000 on: 40 ... 128 node classes (two per lane beyond 64, round 4)
    n_node_classes = int(rng.choice([40, 65, 70, 100, 128] if wide else [1, 2, 4, 9, 20, 40]))
    n_pod_classes = int(rng.choice([1, 3, 8, 30, 64, 65, 100, 128, 129, 200, 256, 384]))
    if case >= 900000:                    # cases from 900 000 on: 129 ... 256 node classes (simon_table_cls4.hip, end of round 6): one-level layout, few enough
        n_node_classes = int(rng.choice([129, 140, 160, 192, 200, 256, 257]))   # signatures for its LDS (257 classes: off the table)
        n_pod_classes = int(rng.choice([1, 3, 8, 20, 40]))
        N = max(N, int(rng.integers(300, 1500)))
    if size == 3:                         # static masks are O(Cp N) Python work in the generator
        feat.pop("static_mask", None)
    prob = randprob.rand_problem(52000 + case, N=N, P=P, n_node_classes=n_node_classes, n_pod_classes=n_pod_classes, **feat)
    if rng.random() < 0.3:                # caller classes that do not determine the allocatable: the kernel splits them
        shapes_c = np.array([4000, 8000, 16000, 32000]) + (rng.integers(0, 5, 4) if "odd_units" in feat else 0)
        shapes_m = (np.array([8, 16, 64, 128]) << 30) + (rng.integers(0, 7, 4) if "odd_units" in feat else 0)
        pick = rng.integers(0, 4, N)
        lim = 256 if case >= 900000 else 128 if wide else 64
        if n_node_classes * 4 > lim:
            pick = prob.node_class % (3 if n_node_classes * 3 <= lim else 1)
        prob.alloc_cpu = shapes_c[pick].astype(np.int64)
        prob.alloc_mem = shapes_m[pick].astype(np.int64)
        if prob.init_req_cpu is not None:
            prob.init_req_cpu = np.minimum(prob.init_req_cpu, prob.alloc_cpu // 2)
            prob.init_req_mem = np.minimum(prob.init_req_mem, prob.alloc_mem // 2)
            prob.init_nz_cpu = np.maximum(prob.init_nz_cpu, prob.init_req_cpu)
            prob.init_nz_mem = np.maximum(prob.init_nz_mem, prob.init_req_mem)
    S = int(rng.integers(1, 9))
    scen, orders = randprob.rand_scenarios(case, prob, S=S, min_n=1 if rng.random() < 0.5 else None)
    ranks = None
    if case % 5 == 4:                     # per-scenario node order (simon_set_node_ranks): own class lists per scenario
        ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
        for s, (n, _) in enumerate(np.asarray(scen).tolist()):
            ranks[s, :n] = rng.permutation(n)
    ref = O.run_threaded(prob, scen, orders) if ranks is None else O.run(prob, scen, orders, node_ranks=ranks)
    saved = os.environ.get("SIMON_TABLE_COARSE")
    saved_ws = os.environ.get("SIMON_LDS_WS")
    if saved is None:
        os.environ["SIMON_TABLE_COARSE"] = "1" if coarse else "0"      # read once, when the context is created
    if saved_ws is None and (case // 3) % 2 == 1:
        os.environ["SIMON_LDS_WS"] = "0"                               # round 5: every other case keeps the workspace in HBM (the others: in LDS where it fits)
    try:
        ctx = capi.Context(0)
    finally:
        if saved is None:
            del os.environ["SIMON_TABLE_COARSE"]
        if saved_ws is None:
            os.environ.pop("SIMON_LDS_WS", None)
    with ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        if ranks is not None:
            ctx.set_node_ranks(ranks)
        ctx.run_loaded(True)
        res = ctx.fetch(True)
        st = ctx.stats()
    ok = (res.unscheduled.tolist() == ref.unscheduled.tolist() and res.used_cpu.tolist() == ref.used_cpu.tolist() and
          res.used_mem.tolist() == ref.used_mem.tolist() and (res.placement == ref.placement).all())
    n_sigs = len({(a, b, c, d, e) for a, b, c, d, e in zip(prob.req_cpu.tolist(), prob.req_mem.tolist(),
                                                          (prob.nz_cpu if prob.nz_cpu is not None else prob.req_cpu).tolist(),
                                                          (prob.nz_mem if prob.nz_mem is not None else prob.req_mem).tolist(), prob.pod_class.tolist())})
    return ok, dict(case=case, N=N, P=P, S=S, classes=n_node_classes, pod_classes=n_pod_classes, sigs=n_sigs, feat=sorted(feat),
                    variant=st.kernel_variant, generation=st.kernel_generation)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad, on_table, two_level = 0, 0, 0
    for case in range(first, first + n_cases):
        ok, info = one_case(case)
        on_table += info["generation"] in (4, 5)
        two_level += info["generation"] == 5
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
    print(f"fuzz_table: {n_cases} cases from {first}, {on_table} on the score-table kernel ({two_level} with the two-level summary), "
          f"mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
