#!/usr/bin/env python3
"""Randomised parity sweep of generation 7 (soft PodTopologySpread constraints on the score-table kernel): random sizes, feature
subsets, scenario batches and (every fourth case) per-scenario node ranks, each case on the single-wave AND the team shape; every placement, unscheduled count and used cpu / memory against the oracle.
Not collected by pytest; by hand on a GPU box:   python tests/fuzz_spread.py [n_cases] [first_seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402,F401
import oracle_lib as O  # noqa: E402
import randprob  # noqa: E402
from open_simulator_amd import capi  # noqa: E402

FEATURES = ["nz_differs", "init_state", "static_mask", "presets", "gates", "zero_pods", "tight_pods", "pins", "odd_units", "static_small", "anti_host", "ipa_self", "ipa", "hard_simple", "gpu"]


def one_case(case):
    rng = np.random.default_rng(88000 + case)
    size = case % 3
    N = int(rng.integers(2, 90)) if size == 0 else int(rng.integers(100, 900)) if size == 1 else int(rng.integers(900, 3000))
    P = int(rng.integers(20, 500 if size == 0 else 2500))
    feat = {f: True for f in FEATURES if rng.random() < 0.3}
    if size == 2:
        feat.pop("static_mask", None)
    mask_rows = case >= 700000                          # cases from 700 000 on: GPU share / required anti-affinity / ports / extra resources that do NOT fold into
    if mask_rows:                                       # the table, next to the soft constraints: the walks over the position-mask rows (REST && SPREAD)
        for f, pr in (("gpu", 0.6), ("anti_host", 0.4), ("anti", 0.3), ("ports", 0.25), ("eph", 0.2)):
            if rng.random() < pr:
                feat[f] = True
        if rng.random() < 0.2:
            feat["scalars"] = int(rng.integers(1, 4))
        if "anti" in feat:
            feat.pop("anti_host", None)
            if rng.random() < 0.4:
                feat["aff"] = True                      # required affinity beside the anti-affinity terms (rows that must be SET, first-pod escape)
    many_classes = 600000 <= case < 700000                       # cases from 600 000 on: 65 .. 128 internal node classes (node shapes x zones; simon_table.hip: CN2)
    prob = randprob.rand_problem(99000 + case, N=N, P=P, spread_soft=(case % 7 != 6 or "ipa_self" not in feat), n_node_classes=int(rng.choice([14, 18, 22, 25] if many_classes else [1, 2, 4, 9])),
                                 n_pod_classes=int(rng.choice([130, 200, 300, 384, 600, 1000] if 300000 <= case < 600000 else [1, 3, 8, 30, 60])), **feat)   # cases from 300 000 on: 129 ... 384 signatures (MANY && SPREAD)
    if "gpu" in feat:                                   # GPU share folded into the table (few request kinds: <= 128 signatures), behind the spread walk
        G_ = 1 << 30
        prob.gpu_mem = np.where(prob.gpu_mem > 4 * G_, 8 * G_, np.where(prob.gpu_mem > 0, 2 * G_, 0)).astype(np.int64)
        prob.pod_gpu_cnt = np.where(prob.gpu_mem > 0, np.where(prob.pod_gpu_cnt >= 2, 2, 1), 0).astype(np.int32)
        prob.normalise()
    scen, orders = randprob.rand_scenarios(case, prob, S=int(rng.integers(1, 8)), min_n=1 if rng.random() < 0.4 else None)
    ranks = None
    if case % 4 == 3:                                   # per-scenario node order (simon_set_node_ranks): ties follow the ranks
        ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
        for s_, (n, _) in enumerate(np.asarray(scen).tolist()):
            ranks[s_, :n] = rng.permutation(n)
    ref = O.run(prob, scen, orders, node_ranks=ranks) if ranks is not None else O.run_threaded(prob, scen, orders)
    if mask_rows:
        os.environ["SIMON_NO_FOLD"] = "1"; os.environ["SIMON_NO_GPU_FOLD"] = "1"
    ok, teams = True, 0
    for team in ("0", "1"):                             # one wave per scenario, then a team of waves (simon_table.hip: NW waves per scenario)
        os.environ["SIMON_TEAM"] = team
        try:
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                if ranks is None:
                    res = ctx.run_batch(scen, orders)
                else:
                    ctx.load_scenarios(scen, orders)
                    ctx.set_node_ranks(ranks)
                    ctx.run_loaded(True)
                    res = ctx.fetch(True)
                st = ctx.stats()
        finally:
            os.environ.pop("SIMON_TEAM", None)
        teams += st.workgroup_size > 64 and st.kernel_generation == 7
        ok = ok and (res.unscheduled.tolist() == ref.unscheduled.tolist() and res.used_cpu.tolist() == ref.used_cpu.tolist() and
                     res.used_mem.tolist() == ref.used_mem.tolist() and bool((res.placement == ref.placement).all()))
    os.environ.pop("SIMON_NO_FOLD", None); os.environ.pop("SIMON_NO_GPU_FOLD", None)
    return ok, dict(case=case, N=N, P=P, S=len(scen), feat=sorted(feat), generation=st.kernel_generation, variant=st.kernel_variant, team=teams)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = on7 = teams = 0
    for case in range(first, first + n_cases):
        ok, info = one_case(case)
        on7 += info["generation"] == 7
        teams += info["team"]
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
    print(f"fuzz_spread: {n_cases} cases from {first} (each with one wave and with a team of waves per scenario), {on7} on generation 7, {teams} of them also in team mode, mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
