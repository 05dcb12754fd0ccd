#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

Each fixture = SoA inputs + scenarios + the oracle's outputs, plus the reference-derived expectation it was pinned
against (see README.md here).  Full-size cases store SHA-256 digests of the placement matrices instead of the data."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fixture_simple  # noqa: E402
import oracle_lib as O  # noqa: E402
from open_simulator_amd import capi, synth  # noqa: E402
from test_oracle import gpushare_problem, kav_problem  # noqa: E402

FIELDS = ["alloc_cpu", "alloc_mem", "alloc_pods", "node_class", "gpu_cnt", "gpu_mem_total", "req_cpu", "req_mem", "nz_cpu",
          "nz_mem", "pod_class", "preset_node", "gpu_mem", "pod_gpu_cnt", "static_mask", "static_reason", "simon_raw",
          "const_score"]


def dump_problem(prob):
    d = {"n_pod_classes": prob.n_pod_classes, "n_node_classes": prob.n_node_classes}
    for f in FIELDS:
        v = getattr(prob, f)
        if v is not None:
            d[f] = np.asarray(v).tolist()
    return d


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def small(name, prob, scen, orders, reference_expectation, extra=None):
    res = O.run(prob, scen, orders)
    out = {"name": name, "problem": dump_problem(prob), "scenarios": np.asarray(scen).tolist(), "orders": np.asarray(orders).tolist(),
           "reference_expectation": reference_expectation,
           "oracle": {"unscheduled": res.unscheduled.tolist(), "used_cpu": res.used_cpu.tolist(), "used_mem": res.used_mem.tolist(),
                      "placement": res.placement.tolist()}}
    out.update(extra or {})
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    return res


def main():
    prob, raw = kav_problem()
    small("kav_two_nodes", prob, [[2, 0]], np.arange(2)[None],
          "SURVEY.md 8(c) hand-derived vector: LA 87/93, BA 100/100, Simon 100/0 -> pod 0 and pod 1 both on node A (index 0)")
    prob = gpushare_problem()
    small("example_gpushare", prob, [[2, 0]], np.arange(9)[None],
          "example/simon-gpushare-config.yaml: 0 unscheduled pods on the 2 existing nodes")
    prob, names = fixture_simple.problem()
    P = prob.n_pods
    small("core_test_simple", prob, [[4, 0]], np.arange(P)[None],
          "pkg/simulator/core_test.go:346 failedPodsNum == 0; per-workload pod counts :364-591",
          {"pod_names": names, "expected_counts": fixture_simple.EXPECTED_COUNTS})
    # full-size synthetic cases: digests only
    big = {}
    for hom in (False, True):
        prob, scen, orders = synth.config2(hom)
        r = O.run(prob, scen, orders)
        big[f"config2_{'homogeneous' if hom else 'heterogeneous'}"] = {
            "unscheduled": r.unscheduled.tolist(), "used_cpu": r.used_cpu.tolist(), "placement_sha256": digest(r.placement)}
    prob, scen, orders = synth.config3()
    pick = [0, 1, 2, 3, 4 * 7 + 1, 4 * 100, 4 * 300 + 2, 4 * 511 + 3, 4 * 512, 4 * 800 + 1, 4 * 1023 + 2, 4 * 1023 + 3]
    r = O.run(prob, scen[pick], orders)
    big["config3_subset"] = {"pick": pick, "scenarios": scen[pick].tolist(), "unscheduled": r.unscheduled.tolist(),
                             "used_cpu": r.used_cpu.tolist(), "placement_sha256": [digest(row) for row in r.placement]}
    with open(os.path.join(HERE, "synthetic_digests.json"), "w") as f:
        json.dump(big, f, indent=1)


if __name__ == "__main__":
    main()
