"""Seeded random problems that exercise every input feature of include/simon_hip.h (tests only)."""
import numpy as np

from open_simulator_amd import capi

MiB = 1 << 20
GiB = 1 << 30


def rand_problem(seed, N=40, P=200, *, nz_differs=False, init_state=False, static_mask=False, presets=False,
                 gates=False, eph=False, scalars=0, gpu=False, anti=False, zero_pods=False, tight_pods=False,
                 odd_units=False, n_node_classes=5, n_pod_classes=6, aff=False, ipa=False, spread_hard=False,
                 spread_soft=False, static_scores=False, local=False, pins=False, anti_host=False, ports=False, static_small=False, ipa_self=False, hard_simple=False):
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
    if static_scores:   # NodeAffinity preferred / TaintToleration PreferNoSchedule / NodePreferAvoidPods tables
        prob.node_affinity_raw = (rng.integers(0, 4, (n_pod_classes, n_node_classes)) * rng.integers(1, 60)).astype(np.int64)
        prob.taint_prefer_raw = rng.integers(0, 3, (n_pod_classes, n_node_classes)).astype(np.int64)
        prob.static_add = (rng.integers(0, 2, (n_pod_classes, n_node_classes)) * 1000000).astype(np.int64)
        if seed % 2:        # all-zero rows exercise the max == 0 branches
            prob.node_affinity_raw[0] = 0
            prob.taint_prefer_raw[-1] = 0
    if static_small:    # the same tables with additions small enough for the score-table kernel's class term (no NodePreferAvoidPods)
        prob.node_affinity_raw = (rng.integers(0, 4, (n_pod_classes, n_node_classes)) * rng.integers(1, 60)).astype(np.int64)
        prob.taint_prefer_raw = rng.integers(0, 3, (n_pod_classes, n_node_classes)).astype(np.int64)
        if seed % 3 == 0:
            prob.static_add = rng.choice([0, 7, 40, 300], (n_pod_classes, n_node_classes)).astype(np.int64)
        if seed % 2:        # all-zero rows exercise the max == 0 branches
            prob.node_affinity_raw[0] = 0
            prob.taint_prefer_raw[-1] = 0
        if seed % 5 == 1:
            prob.node_affinity_raw = None
        if seed % 5 == 2:
            prob.taint_prefer_raw = None
    if pins:            # DaemonSet-style pods: node affinity admits ONE node (some beyond small scenarios' node counts)
        prob.pin_node = np.where(rng.random(P) < 0.2, rng.integers(0, N, P), -1).astype(np.int32)
        if prob.preset_node is not None:
            prob.pin_node = np.where(np.asarray(prob.preset_node) >= 0, -1, prob.pin_node).astype(np.int32)
    if local:           # Open-Local: node storage + per-class volume specs
        GiB_ = 1 << 30
        prob.local_flags = (rng.random(N) < 0.75).astype(np.int32)
        prob.local_vg_cnt = np.where(prob.local_flags > 0, rng.integers(0, capi.MAX_VG + 1, N), 0).astype(np.int32)
        prob.local_vg_cap = (rng.choice([20, 50, 100, 200], (N, capi.MAX_VG)) * GiB_).astype(np.int64)
        prob.init_vg_req = (rng.choice([0, 0, 5, 10], (N, capi.MAX_VG)) * GiB_).astype(np.int64)
        prob.local_vg_name = np.stack([rng.permutation(6)[:capi.MAX_VG] for _ in range(N)]).astype(np.int32)
        prob.local_dev_cnt = np.where(prob.local_flags > 0, rng.integers(0, capi.MAX_LDEV + 1, N), 0).astype(np.int32)
        prob.local_dev_cap = (rng.choice([10, 50, 50, 100, 200], (N, capi.MAX_LDEV)) * GiB_).astype(np.int64)
        media = rng.integers(0, 3, (N, capi.MAX_LDEV))
        prob.local_dev_media = (media << (2 * np.arange(capi.MAX_LDEV))[None, :]).sum(1).astype(np.int32)
        prob.init_dev_alloc = ((rng.random((N, capi.MAX_LDEV)) < 0.15) << np.arange(capi.MAX_LDEV)[None, :]).sum(1).astype(np.int32)
        specs = np.zeros(5, capi.LOCAL_SPEC_DTYPE)
        for sp in specs:
            n_named = int(rng.integers(0, 2))
            sp["n_lvm"] = int(rng.integers(0, capi.MAX_LVOL + 1))
            sp["lvm_size"][:] = rng.choice([1, 5, 10, 20, 40], capi.MAX_LVOL) * GiB_
            sp["lvm_vg"][:] = -1
            sp["lvm_vg"][:min(n_named, sp["n_lvm"])] = rng.integers(0, 6)
            sp["n_ssd"], sp["n_hdd"] = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            sp["ssd_size"][:] = np.sort(rng.choice([5, 10, 40, 80], capi.MAX_LVOL)) * GiB_
            sp["hdd_size"][:] = np.sort(rng.choice([5, 10, 40, 80], capi.MAX_LVOL)) * GiB_
        prob.local_specs = specs
        prob.local_spec_of = np.where(rng.random(n_pod_classes) < 0.6, rng.integers(0, len(specs), n_pod_classes), -1).astype(np.int32)
    spread_hard = spread_hard or hard_simple
    v2 = aff or ipa or spread_hard or spread_soft or ipa_self
    if anti or v2:
        _topology(prob, rng, N, n_pod_classes, anti, aff, ipa, spread_hard, spread_soft, anti_host, ipa_self, hard_simple)
    elif anti_host or ports:
        _topology_host(prob, rng, N, n_pod_classes, anti_host, ports)
    return prob.normalise()


def _csr(lists):
    off = np.cumsum([0] + [len(x) for x in lists]).astype(np.int32)
    flat = [v for x in lists for v in x]
    return off, np.array(flat if flat else [0], np.int32)


def _topology_host(prob, rng, N, Cp, anti=True, ports=False):
    """Required anti-affinity on node-level topology keys only (kubernetes.io/hostname and a second key that also gives every
    node its own domain) and / or host ports: the shape the score-table kernel's REST path takes.  Terms 0..9: selectors (classes
    match up to 4, require up to 3); terms 10..15: (hostIP, protocol, port) triples a class binds (= matches) and conflicts with."""
    prob.topo_dom = np.stack([np.arange(N, dtype=np.int32), rng.permutation(N).astype(np.int32)])
    prob.topo_n_dom = np.array([N, N], np.int32)
    prob.topo_is_hostname = np.array([1, 0], np.uint8)
    T = 16 if ports else 10
    keys = rng.integers(0, 2, T).astype(np.int32)
    keys[10:] = 0                                          # port terms live on the hostname key
    prob.term_topo_key = keys
    match, antil, portl = [], [], []
    for c in range(Cp):
        m = sorted(set(rng.choice(10, rng.integers(1, 5)).tolist())) if anti and rng.random() < 0.6 else []
        a = sorted(set(rng.choice(10, rng.integers(1, 4)).tolist())) if anti and rng.random() < 0.5 else []
        if a and rng.random() < 0.5:
            m = sorted(set(m) | {a[0]})                    # self anti-affinity: one pod of the class per node
        pl = []
        if ports and rng.random() < 0.5:
            bound = sorted(set((10 + rng.choice(6, rng.integers(1, 3))).tolist()))
            m = sorted(set(m) | set(bound))                # the class binds these triples ...
            pl = sorted(set(bound) | ({int(10 + rng.integers(0, 6))} if rng.random() < 0.5 else set()))   # ... and conflicts with them (+ a wildcard twin)
        match.append(m)
        antil.append(a)
        portl.append(pl)
    prob.match_off, prob.match_idx = _csr(match)
    prob.anti_off, prob.anti_idx = _csr(antil)
    if ports:
        prob.port_off, prob.port_idx = _csr(portl)


def _topology(prob, rng, N, Cp, anti, aff, ipa, spread_hard, spread_soft, anti_host=False, ipa_self=False, hard_simple=False):
    """Two topology keys -- hostname (domain = node) and zone (4 zones, some nodes unlabeled) -- and 8 terms over them;
    every role list of include/simon_hip.h gets random entries for the enabled features."""
    zone = rng.integers(-1, 4, N).astype(np.int32)
    if spread_hard and not anti:
        zone = np.abs(zone)            # keep hard-spread problems schedulable: every node labeled
    prob.topo_dom = np.stack([np.arange(N, dtype=np.int32), zone])
    prob.topo_n_dom = np.array([N, 4], np.int32)
    prob.topo_is_hostname = np.array([1, 0], np.uint8)
    T = 8
    prob.term_topo_key = np.array([0, 1, 0, 1, 1, 0, 1, 1], np.int32)
    words = (N + 63) // 64
    sets = np.zeros((2, words), np.uint64)
    for r in range(2):
        for j in range(N):
            if rng.random() < 0.8:
                sets[r, j // 64] |= np.uint64(1) << np.uint64(j % 64)
    prob.node_sets = sets
    tset = np.full(T, -1, np.int32)
    if spread_soft:
        tset[6] = 0
        tset[7] = 1
    prob.term_node_set = tset

    def pick(cands, p_any, kmax):
        if rng.random() >= p_any:
            return []
        return sorted(set(rng.choice(cands, rng.integers(1, kmax + 1)).tolist()))

    match, antil, affl, pref, prefw, own, ownw, hard, hskew, hself, hset, soft, sskew, flags = ([] for _ in range(14))
    for c in range(Cp):
        match.append(pick(np.arange(T), 0.7, 4))
        # anti_host (without anti): required anti-affinity on hostname-key terms only -- what the score table can fold in (term 5 is a
        # soft-spread term as well)
        antil.append(pick([0, 1, 2, 3], 0.5, 2) if anti else pick([0, 2, 5], 0.5, 2) if anti_host else [])
        a = pick([4, 5], 0.4, 2) if aff else []
        affl.append(a)
        flags.append(capi.CLASS_AFF_SELF if a and all(t in match[c] for t in a) else 0)
        pr = pick([0, 1, 2, 3, 5], 0.6, 3) if ipa else []
        pref.append(pr)
        prefw.append([int(rng.integers(1, 101)) * (1 if rng.random() < 0.6 else -1) for _ in pr])
        ow = pick([0, 1, 2, 3, 4], 0.6, 3) if ipa else []
        own.append(ow)
        ownw.append([int(rng.integers(1, 101)) * (1 if rng.random() < 0.7 else -1) for _ in ow])
        if ipa_self and not ipa:
            # preferred pod (anti-)affinity in its usual, self-referential form: whoever matches one of the terms 1, 3 (zone key) or 5
            # (hostname key; also a soft-spread term) owns it with the term's weight; preferences of the class itself on the same terms
            if c == 0:
                self_w = {t: (int(rng.integers(1, 101)) * (1 if rng.random() < 0.3 else -1) if rng.random() < 0.8 else 0) for t in (1, 3, 5)}
                prob._self_w = self_w
            self_w = prob._self_w
            if c == 0:
                prob._own_any = rng.random() < 0.4     # owners need not be matchers (affinity TO another workload): rows that count owners
            if prob._own_any:
                own[-1] = [t for t in (1, 3, 5) if self_w[t] != 0 and rng.random() < 0.4]
            else:
                own[-1] = [t for t in match[c] if t in self_w and self_w[t] != 0]
            ownw[-1] = [self_w[t] for t in own[-1]]
            pr = pick([1, 3, 5], 0.6, 3)
            pref[-1] = pr
            prefw[-1] = [int(rng.integers(1, 101)) * (1 if rng.random() < 0.4 else -1) for _ in pr]
        h = pick([1, 3], 0.5, 2) if spread_hard else []
        hard.append(h)
        hskew.append([int(rng.integers(1, 4)) for _ in h])
        hself.append([int(t in match[c]) for t in h])
        hset.append([-1 if hard_simple else int(rng.integers(-1, 2)) for _ in h])   # hard_simple: every labelled node is eligible
        so = pick([6, 7, 5], 0.7, 3) if spread_soft else []
        soft.append(so)
        sskew.append([int(rng.integers(1, 6)) for _ in so])
    # make some classes own an affinity term they do not match (exercises the non-escape branch)
    prob.match_off, prob.match_idx = _csr(match)
    prob.anti_off, prob.anti_idx = _csr(antil)
    if aff:
        prob.aff_off, prob.aff_idx = _csr(affl)
        prob.class_flags = np.array(flags, np.uint8)
    if ipa or ipa_self:
        prob.pref_off, prob.pref_idx = _csr(pref)
        prob.pref_w = _csr(prefw)[1]
        prob.own_off, prob.own_idx = _csr(own)
        prob.own_w = _csr(ownw)[1]
    if spread_hard:
        prob.spread_hard_off, prob.spread_hard_idx = _csr(hard)
        prob.spread_hard_skew = _csr(hskew)[1]
        prob.spread_hard_self = _csr(hself)[1]
        prob.spread_hard_set = np.array([v for x in hset for v in x] or [0], np.int32)
    if spread_soft:
        from open_simulator_amd.gomath import spread_log_table
        prob.spread_soft_off, prob.spread_soft_idx = _csr(soft)
        prob.spread_soft_skew = _csr(sskew)[1]
        prob.spread_log = spread_log_table(N)


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
