#!/usr/bin/env python3
"""Randomised HIP-vs-oracle parity sweep beyond the committed -m gpu cases (not collected by pytest; run by hand on a
GPU box):  python tests/fuzz_gpu.py [n_cases] [first_seed].  Every case draws a random subset of the generator's features,
random sizes and scenarios, optionally per-scenario node ranks, and compares every placement with the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402,F401
import oracle_lib as O  # noqa: E402
import randprob  # noqa: E402
from open_simulator_amd import capi  # noqa: E402

NARROW = ["nz_differs", "init_state", "static_mask", "presets", "gates", "zero_pods", "tight_pods", "pins"]
WIDE = NARROW + ["eph", "gpu", "anti", "aff", "ipa", "spread_hard", "spread_soft", "static_scores", "local"]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for case in range(first, first + n_cases):
        rng = np.random.default_rng(90000 + case)
        pool = NARROW if case % 3 == 0 else WIDE
        feat = {f: True for f in pool if rng.random() < 0.35}
        if rng.random() < 0.3 and pool is WIDE:
            feat["scalars"] = int(rng.integers(1, 4))
        N, P, S = int(rng.integers(3, 300)), int(rng.integers(1, 500)), int(rng.integers(1, 10))
        prob = randprob.rand_problem(70000 + case, N=N, P=P, **feat)
        scen, orders = randprob.rand_scenarios(case, prob, S=S)
        v6 = []
        if rng.random() < 0.35:                              # ABI v6: pods of unequal priority -> the DefaultPreemption-risk flag per scenario
            prob.priority = rng.choice([0, 0, 5, 100, -1], P).astype(np.int32)
            prob.init_min_priority = int(rng.choice([0x7fffffff, 0, 50]))
            v6.append("priority")
        if rng.random() < 0.35:                              # ABI v6: ScalarResources entries of quantity 0 (bit k < K: tracked resources, bit 7: any other)
            K = 0 if prob.scalar_req is None else prob.scalar_req.shape[0]
            ent = np.where(rng.random(P) < 0.4, 0x80, 0).astype(np.uint8)
            for k in range(K):
                ent |= np.where(rng.random(P) < 0.4, 1 << k, 0).astype(np.uint8)
            if "zero_pods" in feat and rng.random() < 0.7:   # some pods that request nothing at all: the entry is all that keeps them off the shortcut
                z = rng.random(P) < 0.3
                prob.req_cpu = np.where(z, 0, prob.req_cpu).astype(np.int64)
                prob.req_mem = np.where(z, 0, prob.req_mem).astype(np.int64)
            prob.scalar_entries = ent
            v6.append("entries")
        ranks = None
        if rng.random() < 0.3:
            ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
            for s, (n, _) in enumerate(np.asarray(scen).tolist()):
                ranks[s, :n] = rng.permutation(n)
        ref = O.run(prob, scen, orders, node_ranks=ranks)
        env = {}
        if rng.random() < 0.25:
            env["SIMON_WG"] = str(rng.choice([64, 128, 256, 512]))
        for k, v in env.items():
            os.environ[k] = v
        try:
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                ctx.load_scenarios(scen, orders)
                if ranks is not None:
                    ctx.set_node_ranks(ranks)
                ctx.run_loaded(True)
                res = ctx.fetch(True)
                risk = ctx.fetch_preempt_risk()
                variant = ctx.stats().kernel_variant
        finally:
            for k in env:
                os.environ.pop(k, None)
        ok = (res.unscheduled.tolist() == ref.unscheduled.tolist() and res.used_cpu.tolist() == ref.used_cpu.tolist() and
              res.used_mem.tolist() == ref.used_mem.tolist() and (res.placement == ref.placement).all() and
              res.used_vg.tolist() == ref.used_vg.tolist() and risk.tolist() == ref.preempt_risk.tolist())
        if not ok:
            bad += 1
            print("MISMATCH case", case, "N", N, "P", P, "S", S, "variant", variant, "feat", feat, "ranks", ranks is not None, "env", env, "v6", v6, flush=True)
    print(f"fuzz: {n_cases} cases from {first}, mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
