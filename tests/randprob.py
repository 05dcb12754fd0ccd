"""Seeded random problems that exercise every input feature of include/simon_hip.h (tests only)."""
import numpy as np

from open_simulator_amd import capi

MiB = 1 << 20
GiB = 1 << 30


def rand_problem(seed, N=40, P=200, *, nz_differs=False, init_state=False, static_mask=False, presets=False,
                 gates=False, eph=False, scalars=0, gpu=False, anti=False, zero_pods=False, tight_pods=False,
                 odd_units=False, n_node_classes=5, n_pod_classes=6):
    rng = np.random.default_rng(seed)
    ncls = rng.integers(0, n_node_classes, N).astype(np.int32)
    cls_cpu = rng.choice([2000, 4000, 8000, 16000, 32000, 64000], n_node_classes)
    cls_mem = rng.choice([4, 8, 16, 32, 64, 128], n_node_classes) * GiB
    if odd_units:   # gcd 1: forces wide-range arithmetic in the narrow normalisation
        cls_cpu = cls_cpu + rng.integers(1, 7, n_node_classes)
        cls_mem = cls_mem + rng.integers(1, 999, n_node_classes)
    alloc_cpu = cls_cpu[ncls].astype(np.int64)
    alloc_mem = cls_mem[ncls].astype(np.int64)
    alloc_pods = (rng.integers(3, 12, N) if tight_pods else np.full(N, 110)).astype(np.int32)
    pcls = rng.integers(0, n_pod_classes, P).astype(np.int32)
    shape_cpu = rng.choice([0, 100, 250, 500, 1000, 2000, 4000], n_pod_classes)
    shape_mem = rng.choice([0, 64, 128, 256, 512, 1024, 4096], n_pod_classes) * MiB
    if not zero_pods:
        shape_cpu = np.maximum(shape_cpu, 100)
        shape_mem = np.maximum(shape_mem, 64 * MiB)
    if odd_units:
        shape_cpu = shape_cpu + rng.integers(0, 3, n_pod_classes)
        shape_mem = shape_mem + rng.integers(0, 5, n_pod_classes)
    req_cpu = shape_cpu[pcls].astype(np.int64)
    req_mem = shape_mem[pcls].astype(np.int64)
    prob = capi.Problem(alloc_cpu=alloc_cpu, alloc_mem=alloc_mem, alloc_pods=alloc_pods, node_class=ncls,
                        req_cpu=req_cpu, req_mem=req_mem, pod_class=pcls,
                        n_pod_classes=n_pod_classes, n_node_classes=n_node_classes,
                        simon_raw=rng.integers(0, 120, (n_pod_classes, n_node_classes)).astype(np.int64),
                        const_score=np.full(n_pod_classes, 1000300, np.int64))
    if nz_differs:  # containers without requests: defaults 100m / 200MiB (V/util/non_zero.go:35-38)
        prob.nz_cpu = np.where(req_cpu == 0, 100, req_cpu).astype(np.int64)
        prob.nz_mem = np.where(req_mem == 0, 200 * MiB, req_mem).astype(np.int64)
        extra = rng.integers(0, 2, P)
        prob.nz_cpu = prob.nz_cpu + extra * 100
        prob.nz_mem = prob.nz_mem + extra * 200 * MiB
    if init_state:
        f = rng.random(N) * 0.5
        prob.init_req_cpu = (alloc_cpu * f // 100 * 100).astype(np.int64)
        prob.init_req_mem = (alloc_mem * f // MiB * MiB).astype(np.int64)
        prob.init_nz_cpu = prob.init_req_cpu + rng.integers(0, 3, N) * 100
        prob.init_nz_mem = prob.init_req_mem + rng.integers(0, 3, N) * 200 * MiB
        prob.init_npods = rng.integers(0, 5, N).astype(np.int32)
    if static_mask:
        words = (N + 63) // 64
        bits = rng.random((n_pod_classes, N)) < 0.8
        mask = np.zeros((n_pod_classes, words), np.uint64)
        for c in range(n_pod_classes):
            for j in range(N):
                if bits[c, j]:
                    mask[c, j // 64] |= np.uint64(1) << np.uint64(j % 64)
        prob.static_mask = mask
        prob.static_reason = np.where(bits, 0, rng.integers(1, 5, (n_pod_classes, N))).astype(np.uint8)
    if presets:
        pr = np.full(P, -1, np.int32)
        idx = rng.choice(P, P // 10, replace=False)
        pr[idx] = rng.integers(0, max(1, N // 2), len(idx))
        prob.preset_node = pr
    if gates:
        g = np.full(P, -1, np.int32)
        idx = rng.choice(P, P // 8, replace=False)
        g[idx] = rng.integers(0, N, len(idx))
        prob.gate_node = g
        if presets:  # a preset pod must be gated on its own node at least
            prob.gate_node = np.where(prob.preset_node >= 0, np.maximum(g, prob.preset_node), g).astype(np.int32)
    elif presets:
        prob.gate_node = np.where(prob.preset_node >= 0, prob.preset_node, -1).astype(np.int32)
    if eph:
        prob.alloc_eph = (rng.choice([20, 50, 100], N) * GiB).astype(np.int64)
        prob.req_eph = (rng.choice([0, 1, 2, 8], P) * GiB).astype(np.int64)
        if init_state:
            prob.init_req_eph = (rng.integers(0, 5, N) * GiB).astype(np.int64)
    if scalars:
        K = scalars
        prob.scalar_alloc = rng.integers(0, 9, (K, N)).astype(np.int64)
        prob.scalar_req = (rng.integers(0, 3, (K, P)) * (rng.random((K, P)) < 0.3)).astype(np.int64)
        if init_state:
            prob.init_scalar_req = rng.integers(0, 2, (K, N)).astype(np.int64)
    if gpu:
        cnt = rng.choice([0, 0, 1, 2, 4, 8], N).astype(np.int32)
        prob.gpu_cnt = cnt
        prob.gpu_mem_total = (cnt.astype(np.int64) * rng.choice([8, 16, 16280 / 1024], N) * GiB).astype(np.int64)
        gm = (rng.choice([0, 0, 0, 1, 2, 4, 8, 16], P) * GiB).astype(np.int64)
        prob.gpu_mem = gm
        prob.pod_gpu_cnt = np.where(gm > 0, rng.choice([0, 1, 1, 1, 2, 3], P), 0).astype(np.int32)
        if init_state:
            used = np.zeros((N, capi.MAX_GPU_DEV), np.int64)
            for j in range(N):
                for d in range(cnt[j]):
                    used[j, d] = rng.choice([0, 0, 1, 2]) * GiB
            prob.init_gpu_used = used
    if anti:
        # two topology keys: hostname (domain = node) and zone (4 zones, some nodes unlabeled)
        zone = rng.integers(-1, 4, N).astype(np.int32)
        prob.topo_dom = np.stack([np.arange(N, dtype=np.int32), zone])
        prob.topo_n_dom = np.array([N, 4], np.int32)
        T = 4
        prob.term_topo_key = np.array([0, 1, 0, 1], np.int32)
        anti_lists, match_lists = [], []
        for c in range(n_pod_classes):
            anti_lists.append(sorted(set(rng.choice(T, rng.integers(0, 3)).tolist())) if rng.random() < 0.5 else [])
            match_lists.append(sorted(set(rng.choice(T, rng.integers(0, 3)).tolist())) if rng.random() < 0.6 else [])
        prob.anti_off = np.cumsum([0] + [len(x) for x in anti_lists]).astype(np.int32)
        prob.anti_idx = np.array([t for x in anti_lists for t in x], np.int32)
        prob.match_off = np.cumsum([0] + [len(x) for x in match_lists]).astype(np.int32)
        prob.match_idx = np.array([t for x in match_lists for t in x], np.int32)
    return prob.normalise()


def rand_scenarios(seed, prob, S=6, n_orders=3, min_n=None):
    rng = np.random.default_rng(seed + 77)
    N, P = prob.n_nodes, prob.n_pods
    lo = min_n if min_n is not None else max(1, N // 2)
    if prob.preset_node is not None and prob.gate_node is None:
        lo = max(lo, int(prob.preset_node.max()) + 1)
    orders = np.stack([np.arange(P, dtype=np.int32)] + [rng.permutation(P).astype(np.int32) for _ in range(n_orders - 1)])
    scen = np.stack([rng.integers(lo, N + 1, S), rng.integers(0, n_orders, S)], 1).astype(np.int32)
    scen[0] = (N, 0)
    return scen, orders
