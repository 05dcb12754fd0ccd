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
"SIMONFX2" = the same followed by the optional arrays integration/go/hipengine/engine.go fills:
  int32 K has_gpu has_mask, int64 alloc_eph[N] req_eph[P] scalar_alloc[K][N] scalar_req[K][P], int32 preset[P] gate[P] pin[P],
  int64 const_score[Cp], (has_gpu) int32 gpu_cnt[N], int64 gpu_mem_total[N] pod_gpu_mem[P], int32 pod_gpu_cnt[P], uint32 gpu_index[P],
  uint64 golden gpu_slices[S][P], (has_mask) uint64 static_mask[Cp][ceil(N/64)], uint8 static_reason[Cp][N].
"SIMONFX3" (tests/cabi/cabi_terms.c) is self-describing: every optional array of the three input structs travels by NAME
(see write_fx3), followed by the golden placements, failure codes and Open-Local error sizes.
The golden values come from oracle/simon_oracle.c (TEST INFRASTRUCTURE)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from open_simulator_amd import synth  # noqa: E402


def write(path, prob, scen, orders, explain_scenario, ext=False):
    prob.normalise()
    scen = np.ascontiguousarray(scen, np.int32)
    orders = np.ascontiguousarray(orders, np.int32)
    ref = O.run(prob, scen, orders, want_gpu_slices=ext)
    plan = O.min_plan(prob, scen, ref)
    _, (nf, failed, codes) = O.run(prob, scen, orders, explain_scenario=explain_scenario, max_failed=8)
    assert nf > 0, "pick an explain scenario with unscheduled pods"
    k = len(failed)
    with open(path, "wb") as f:
        f.write(b"SIMONFX2" if ext else b"SIMONFX1")
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
        if ext:
            N, P, Cp = prob.n_nodes, prob.n_pods, prob.n_pod_classes
            K = 0 if prob.scalar_alloc is None else prob.scalar_alloc.shape[0]
            has_gpu, has_mask = prob.gpu_cnt is not None, prob.static_mask is not None

            def arr(a, t, n):
                return (np.zeros(n, t) if a is None else np.ascontiguousarray(a).astype(t)).tobytes()
            f.write(np.array([K, has_gpu, has_mask], "<i4").tobytes())
            f.write(arr(prob.alloc_eph, "<i8", N)); f.write(arr(prob.req_eph, "<i8", P))
            f.write(arr(prob.scalar_alloc, "<i8", K * N)); f.write(arr(prob.scalar_req, "<i8", K * P))
            f.write((np.full(P, -1, "<i4") if prob.preset_node is None else prob.preset_node.astype("<i4")).tobytes())
            f.write((np.full(P, -1, "<i4") if prob.gate_node is None else prob.gate_node.astype("<i4")).tobytes())
            f.write((np.full(P, -1, "<i4") if prob.pin_node is None else prob.pin_node.astype("<i4")).tobytes())
            f.write(arr(prob.const_score, "<i8", Cp))
            if has_gpu:
                f.write(arr(prob.gpu_cnt, "<i4", N)); f.write(arr(prob.gpu_mem_total, "<i8", N))
                f.write(arr(prob.gpu_mem, "<i8", P)); f.write(arr(prob.pod_gpu_cnt, "<i4", P)); f.write(arr(prob.gpu_index, "<u4", P))
                f.write(ref.gpu_slices.astype("<u8").tobytes())
            if has_mask:
                f.write(prob.static_mask.astype("<u8").tobytes()); f.write(arr(prob.static_reason, "u1", Cp * N))
    print(path, os.path.getsize(path), "bytes; unscheduled", ref.unscheduled.tolist(), "plan", plan.as_dict())


def write_fx3(path, prob, scen, orders, explain_scenario, max_failed=6):
    """SIMONFX3 (tests/cabi/cabi_terms.c): every optional array by NAME -- the three ctypes structs of capi.py walked field by field."""
    import ctypes as C
    from open_simulator_amd import capi
    prob.normalise()
    scen = np.ascontiguousarray(scen, np.int32)
    orders = np.ascontiguousarray(orders, np.int32)
    ref = O.run(prob, scen, orders)
    _, (nf, failed, codes) = O.run(prob, scen, orders, explain_scenario=explain_scenario, max_failed=max_failed)
    detail = O.LAST_LOCAL_DETAIL[0]
    assert nf > 0, "pick an explain scenario with unscheduled pods"
    structs = [(0, prob.c_nodes()), (1, prob.c_pods()), (2, prob.c_tables())]
    attr_of = {(1, "gpu_cnt"): "pod_gpu_cnt"}                 # the Problem attribute behind a struct member (same name otherwise)
    fields = []
    for which, st in structs:
        for name, ctype in st._fields_:
            if not hasattr(ctype, "contents"):                 # counters, struct_size
                continue
            if not getattr(st, name):                          # NULL: optional array absent
                continue
            arr = getattr(prob, attr_of.get((which, name), name))
            fields.append((name, which, np.ascontiguousarray(arr).tobytes()))
    nd, tb = structs[0][1], structs[2][1]
    with open(path, "wb") as f:
        f.write(b"SIMONFX3")
        f.write(np.array([prob.n_nodes, prob.n_pods, prob.n_pod_classes, prob.n_node_classes, len(scen), len(orders), nd.n_scalar, nd.n_topo_keys,
                          tb.n_terms, tb.n_node_sets, tb.n_local_specs, len(fields)], "<i4").tobytes())
        for name, which, data in fields:
            f.write(name.encode().ljust(24, b"\0")); f.write(np.array([which, 0], "<i4").tobytes()); f.write(np.array([len(data)], "<i8").tobytes())
            f.write(data + b"\0" * (-len(data) % 8))
        f.write(scen.astype("<i4").tobytes()); f.write(orders.astype("<i4").tobytes())
        f.write(ref.unscheduled.astype("<i4").tobytes()); f.write(ref.placement.astype("<i4").tobytes())
        k = len(failed)
        f.write(np.array([explain_scenario, k, 0 if detail is None else 1], "<i4").tobytes())
        fb = failed.astype("<i4").tobytes()
        f.write(fb + b"\0" * (4 if k & 1 else 0))
        cb = codes.astype("<u2").tobytes()
        f.write(cb + b"\0" * (-len(cb) % 8))
        if detail is not None:
            f.write(detail.astype("<i8").tobytes())
    print(path, os.path.getsize(path), "bytes;", len(fields), "fields; unscheduled", ref.unscheduled.tolist())


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
    # every optional array the Go shim fills: ephemeral storage, two extended resources, GPU share with arriving gpu-index lists,
    # static masks, presets, gates, pinned pods
    import randprob
    from open_simulator_amd import capi
    prob = randprob.rand_problem(77, N=48, P=400, eph=True, scalars=2, gpu=True, static_mask=True, presets=True, gates=True, pins=True,
                                 tight_pods=True)
    rng = np.random.default_rng(77)
    g = np.zeros(prob.n_pods, np.uint32)
    for p in np.flatnonzero(prob.gpu_mem > 0)[::4]:
        g[p] = capi.pack_gpu_index(rng.integers(0, 8, int(rng.integers(1, 3))).tolist())
    prob.gpu_index = g
    scen, orders = randprob.rand_scenarios(77, prob, S=6)
    ref = O.run(prob, scen, orders)
    write(os.path.join(HERE, "cabi_features.bin"), prob, scen, orders, int(np.argmax(ref.unscheduled)), ext=True)
    # SIMONFX3: what integration/go/hipengine/flatten_terms.go fills -- required / preferred (anti-)affinity, hard and soft spread
    # constraints with node sets, host ports, static score tables (the all-feature kernel), and Open-Local with failures whose reasons
    # carry sizes (simon_explain_local_detail)
    prob = randprob.rand_problem(91, N=40, P=350, anti=True, aff=True, ipa=True, spread_hard=True, spread_soft=True, static_scores=True, ports=True,
                                 tight_pods=True, presets=True)
    scen, orders = randprob.rand_scenarios(91, prob, S=4)
    ref = O.run(prob, scen, orders)
    write_fx3(os.path.join(HERE, "cabi_terms.bin"), prob, scen, orders, int(np.argmax(ref.unscheduled)))
    prob = randprob.rand_problem(3, N=40, P=400, local=True, tight_pods=True, init_state=True)
    scen, orders = randprob.rand_scenarios(3, prob, S=3)
    ref = O.run(prob, scen, orders)
    write_fx3(os.path.join(HERE, "cabi_local.bin"), prob, scen, orders, int(np.argmax(ref.unscheduled)))


if __name__ == "__main__":
    main()
