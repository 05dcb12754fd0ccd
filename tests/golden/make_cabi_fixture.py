#!/usr/bin/env python3
"""Writes the binary fixtures tests/cabi/cabi_smoke.c reads (a consumer of the C-ABI that is neither Python nor the
library itself -- it stands in for the cgo host while no Go toolchain exists).

    python tests/golden/make_cabi_fixture.py        # rewrites tests/golden/cabi_*.bin from the oracle

Layout (little endian): "SIMONFX1", int32 N P Cp Cn S n_orders, then
  int64 alloc_cpu[N] alloc_mem[N], int32 alloc_pods[N] node_class[N],
  int64 req_cpu[P] req_mem[P], int32 pod_class[P], int64 simon_raw[Cp*Cn],
  int32 scen[S][2], int32 orders[n_orders][P],
  golden: int32 unscheduled[S], int64 used_cpu[S] used_mem[S], int32 placement[S][P],
  plan (caps 100/100): int32 found scenario n_nodes,
  explain: int32 scenario n_failed, int32 failed[n_failed], uint16 codes[n_failed][n_nodes of that scenario].
The golden values come from oracle/simon_oracle.c (TEST INFRASTRUCTURE)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from open_simulator_amd import synth  # noqa: E402


def write(path, prob, scen, orders, explain_scenario):
    prob.normalise()
    scen = np.ascontiguousarray(scen, np.int32)
    orders = np.ascontiguousarray(orders, np.int32)
    ref = O.run(prob, scen, orders)
    plan = O.min_plan(prob, scen, ref)
    _, (nf, failed, codes) = O.run(prob, scen, orders, explain_scenario=explain_scenario, max_failed=8)
    assert nf > 0, "pick an explain scenario with unscheduled pods"
    k = len(failed)
    with open(path, "wb") as f:
        f.write(b"SIMONFX1")
        f.write(np.array([prob.n_nodes, prob.n_pods, prob.n_pod_classes, prob.n_node_classes, len(scen), len(orders)], "<i4").tobytes())
        pod_class = prob.pod_class if prob.pod_class is not None else np.zeros(prob.n_pods, np.int32)
        node_class = prob.node_class if prob.node_class is not None else np.zeros(prob.n_nodes, np.int32)
        for a, t in ((prob.alloc_cpu, "<i8"), (prob.alloc_mem, "<i8"), (prob.alloc_pods, "<i4"), (node_class, "<i4"),
                     (prob.req_cpu, "<i8"), (prob.req_mem, "<i8"), (pod_class, "<i4"), (prob.simon_raw, "<i8"),
                     (scen, "<i4"), (orders, "<i4"), (ref.unscheduled, "<i4"), (ref.used_cpu, "<i8"), (ref.used_mem, "<i8"),
                     (ref.placement, "<i4")):
            f.write(np.ascontiguousarray(a).astype(t).tobytes())
        f.write(np.array([plan.found, plan.scenario, plan.n_nodes], "<i4").tobytes())
        f.write(np.array([explain_scenario, k], "<i4").tobytes())
        f.write(failed.astype("<i4").tobytes())
        f.write(codes.astype("<u2").tobytes())
    print(path, os.path.getsize(path), "bytes; unscheduled", ref.unscheduled.tolist(), "plan", plan.as_dict())


def main():
    sys.path.insert(0, os.path.dirname(HERE))
    from test_oracle import kav_problem
    prob, _ = kav_problem(n_pods=10)
    # the known-answer vector of SURVEY 8(c) (first two pods land on A) with ten pods, plus a one-node scenario that leaves
    # two of them out (so explain has work)
    P = prob.n_pods
    write(os.path.join(HERE, "cabi_kav.bin"), prob, [[2, 0], [1, 0]], np.arange(P, dtype=np.int32)[None], 1)
    # a config-2-sized sweep: 1000 pods, 128-node pool, 6 cluster sizes x 2 orders
    prob = synth.build_problem(synth.SEED + 2, 1000, 128, 128)
    orders = synth.make_orders(synth.SEED + 2, prob.req_cpu, prob.req_mem, int(prob.alloc_cpu.sum()), int(prob.alloc_mem.sum()), 2)
    counts = [20, 35, 50, 64, 100, 128]
    scen = [[c, o] for c in counts for o in (0, 1)]
    write(os.path.join(HERE, "cabi_config2_sweep.bin"), prob, scen, orders, 0)


if __name__ == "__main__":
    main()
