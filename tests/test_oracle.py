"""CPU tests: the oracle against the reference-derived known answers (no GPU, no HIP calls)."""
import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, synth
from open_simulator_amd.quantity import parse_quantity as pq, simon_raw_score

GiB = 1 << 30
MiB = 1 << 20


def kav_problem(n_pods=2):
    """SURVEY.md section 8(c) hand-derived vector: nodes A=(8000m,16GiB), B=(16000m,32GiB), pod=(1000m,2GiB)."""
    req = {"cpu": pq("1000m"), "memory": pq("2Gi")}
    raw = [simon_raw_score(req, {"cpu": pq(c), "memory": pq(m), "pods": pq("110")}) for c, m in (("8", "16Gi"), ("16", "32Gi"))]
    return capi.Problem(alloc_cpu=[8000, 16000], alloc_mem=[16 * GiB, 32 * GiB], alloc_pods=[110, 110], node_class=[0, 1],
                        req_cpu=[1000] * n_pods, req_mem=[2 * GiB] * n_pods, n_pod_classes=1, n_node_classes=2,
                        simon_raw=np.array([raw]), const_score=np.array([synth.CONST_SCORE])), raw


def test_known_answer_vector_first_pod():
    prob, raw = kav_problem()
    assert raw == [14, 6]                       # int64(100*max(1/7, 2/14)), int64(100*max(1/15, 2/30))
    best, sc = O.score_pod(prob, 2, 0)
    assert sc["feasible"].tolist() == [1, 1]
    assert sc["la"].tolist() == [87, 93]        # ((7000*100/8000)+(14Gi*100/16Gi))/2, (93+93)/2
    assert sc["ba"].tolist() == [100, 100]      # equal cpu and memory fractions
    assert sc["sn"].tolist() == [100, 0]        # (14-6)*100/8, 0
    # totals: 87+100+2*100+1000300, 93+100+0+1000300 (SURVEY prints 1 000 593 for B: an addition slip,
    # its own addends 93+100+0+0+1000300 give 1 000 493)
    assert sc["total"].tolist() == [1000687, 1000493]
    assert best == 0


def test_known_answer_vector_second_pod():
    prob, _ = kav_problem()
    res = O.run(prob, [[2, 0]], np.arange(2)[None])
    assert res.placement.tolist() == [[0, 0]]   # A then A again (LA 75 vs 93 is outweighed by 2*100 Simon)
    assert res.unscheduled.tolist() == [0]
    assert res.used_cpu.tolist() == [2000] and res.used_mem.tolist() == [4 * GiB]
    prob.init_req_cpu = np.array([1000, 0]); prob.init_req_mem = np.array([2 * GiB, 0])
    prob.init_nz_cpu = prob.init_req_cpu; prob.init_nz_mem = prob.init_req_mem; prob.init_npods = np.array([1, 0])
    _, sc = O.score_pod(prob, 2, 1)
    assert sc["la"].tolist() == [75, 93]
    assert sc["total"].tolist() == [1000300 + 75 + 100 + 200, 1000300 + 93 + 100]


def test_fit_reasons_and_zero_request_rule():
    # fit.go:230-302: every insufficient resource is reported; an all-zero request only checks the pod count
    prob = capi.Problem(alloc_cpu=[1000, 1000, 4000], alloc_mem=[GiB, 4 * GiB, GiB], alloc_pods=[110, 110, 0],
                        req_cpu=[2000, 0], req_mem=[2 * GiB, 0], n_pod_classes=1, n_node_classes=1)
    res, (nf, failed, codes) = O.run(prob, [[3, 0]], np.arange(2)[None], explain_scenario=0, max_failed=4)
    assert nf == 1 and failed.tolist() == [0]
    F = capi.FAIL_FIT
    assert codes[0].tolist() == [F | capi.FIT_CPU | capi.FIT_MEM, F | capi.FIT_CPU, F | capi.FIT_PODS | capi.FIT_MEM]
    assert res.placement[0].tolist() == [capi.UNSCHEDULED, 0]   # zero-request pod lands on the first max


def test_gpushare_example_fixture():
    """example/cluster/gpushare + example/application/gpushare restated as SoA (tests/golden/README.md).

    pai-node-00: 64 cpu / 256000Mi, 2 x 16280Mi;  pai-node-01: 64 cpu / 256000Mi, 4 x 16160Mi.
    Pods: gpu-pod-00 4c/9216Mi gpu 1024Mi x1; gpu-pod-01 8c/17408Mi; gpu-pod-02 12c/18432Mi gpu 10240Mi x2;
    6 x gpu-rs-03 8c/18432Mi (annotations sit on the ReplicaSet, not the pod template -> non-GPU pods).
    The reference's expectation for this config (simon-gpushare-config.yaml) is 0 unscheduled pods
    on the 2 existing nodes."""
    prob = gpushare_problem()
    res = O.run(prob, [[2, 0]], np.arange(9)[None])
    assert res.unscheduled.tolist() == [0]
    assert (res.placement >= 0).all()
    assert res.used_cpu.tolist() == [(4 + 8 + 12 + 6 * 8) * 1000]
    # both nodes have identical cpu/mem, so Simon raw is identical and LA/BA drive the spread;
    # determinised tie-break puts the first pod on node 0
    assert res.placement[0, 0] == 0


def gpushare_problem():
    cpu = [64000, 64000]
    mem = [256000 * MiB] * 2
    pods_cpu = [4000, 8000, 12000] + [8000] * 6
    pods_mem = [9216 * MiB, 17408 * MiB, 18432 * MiB] + [18432 * MiB] * 6
    shapes = sorted(set(zip(pods_cpu, pods_mem)), key=lambda t: (pods_cpu + [0]).index(t[0]))
    pcls = [shapes.index((c, m)) for c, m in zip(pods_cpu, pods_mem)]
    raw = []
    for c, m in shapes:
        req = {"cpu": pq(str(c // 1000)), "memory": pq(f"{m // MiB}Mi")}
        row = []
        for gm, gc in (("32560Mi", "2"), ("64640Mi", "4")):
            alloc = {"cpu": pq("64"), "memory": pq("256000Mi"), "pods": pq("110"),
                     "alibabacloud.com/gpu-mem": pq(gm), "alibabacloud.com/gpu-count": pq(gc)}
            row.append(simon_raw_score(req, alloc))
        raw.append(row)
    return capi.Problem(alloc_cpu=cpu, alloc_mem=mem, alloc_pods=[110, 110], node_class=[0, 1],
                        gpu_cnt=[2, 4], gpu_mem_total=[32560 * MiB, 64640 * MiB],
                        req_cpu=pods_cpu, req_mem=pods_mem, pod_class=pcls,
                        gpu_mem=[1024 * MiB, 0, 10240 * MiB] + [0] * 6, pod_gpu_cnt=[1, 0, 2] + [0] * 6,
                        n_pod_classes=len(shapes), n_node_classes=2, simon_raw=np.array(raw),
                        const_score=np.full(len(shapes), synth.CONST_SCORE))


def test_gpu_allocate_rules():
    """gpunodeinfo.go:232-290: 1-GPU pods take the TIGHTEST device (lowest id on ties); n-GPU pods greedily
    stack slices on the first devices that still fit."""
    G = GiB
    prob = capi.Problem(alloc_cpu=[64000], alloc_mem=[256 * G], alloc_pods=[110], gpu_cnt=[4], gpu_mem_total=[64 * G],
                        init_gpu_used=np.array([[10 * G, 4 * G, 12 * G, 0, 0, 0, 0, 0]]),
                        req_cpu=[100] * 5, req_mem=[G] * 5, gpu_mem=[4 * G, 4 * G, 6 * G, 16 * G, 16 * G],
                        pod_gpu_cnt=[1, 1, 2, 1, 1], n_pod_classes=1, n_node_classes=1)
    # idle: [6,12,4,16] -> pod0 (4G x1) tightest = dev2 (idle 4) -> idle [6,12,0,16]
    # pod1 (4G x1) -> dev0 (idle 6) -> [2,12,0,16]; pod2 (6G x2): dev0 no, dev1 twice -> [2,0,0,16]
    # pod3 (16G x1) -> dev3 -> [2,0,0,0]; pod4 (16G) -> nothing fits: unschedulable with code GPUSHARE
    res, (nf, failed, codes) = O.run(prob, [[1, 0]], np.arange(5)[None], explain_scenario=0, max_failed=2)
    assert res.placement[0].tolist() == [0, 0, 0, 0, capi.UNSCHEDULED]
    assert nf == 1 and failed.tolist() == [4] and codes[0].tolist() == [capi.FAIL_GPUSHARE]


def test_anti_affinity_both_directions():
    """interpodaffinity/filtering.go:317-401 with hostname topology: class 0 requires anti-affinity to
    itself; class 1 has none but MATCHES class 0's selector (so existing class-0 pods repel it)."""
    N = 3
    prob = capi.Problem(alloc_cpu=[8000] * N, alloc_mem=[16 * GiB] * N, alloc_pods=[110] * N,
                        topo_dom=np.arange(N, dtype=np.int32)[None], topo_n_dom=[N],
                        req_cpu=[100] * 6, req_mem=[GiB] * 6, pod_class=[0, 0, 0, 0, 1, 1],
                        n_pod_classes=2, n_node_classes=1, term_topo_key=[0],
                        anti_off=[0, 1, 1], anti_idx=[0], match_off=[0, 1, 2], match_idx=[0, 0])
    res, (nf, failed, codes) = O.run(prob, [[N, 0]], np.arange(6)[None], explain_scenario=0, max_failed=4)
    pl = res.placement[0]
    assert sorted(pl[:3].tolist()) == [0, 1, 2]          # one per host
    assert pl[3] == capi.UNSCHEDULED                     # 4th replica has no host left
    assert pl[4] == capi.UNSCHEDULED and pl[5] == capi.UNSCHEDULED
    assert codes[0].tolist() == [capi.FAIL_ANTI_INCOMING] * 3
    assert codes[1].tolist() == [capi.FAIL_ANTI_EXISTING] * 3


def test_min_plan_matches_apply_loop_rule():
    prob, scen, orders = synth.config3(n_counts=6, n_orders=2, n_pods=300, n_het=10)
    res = O.run(prob, scen, orders)
    plan = O.min_plan(prob, scen, res)
    ok = [s for s in range(len(scen)) if res.unscheduled[s] == 0]
    if ok:
        assert plan.found == 1 and plan.n_nodes == min(scen[s, 0] for s in ok)
        assert plan.scenario == min(s for s in ok if scen[s, 0] == plan.n_nodes)
        tot = int(prob.alloc_cpu[:plan.n_nodes].sum())
        assert plan.cpu_pct == int(float(res.used_cpu[plan.scenario]) / float(tot) * 100)
    strict = O.min_plan(prob, scen, res, max_cpu=1, max_mem=1)
    assert strict.found == 0


def test_generator_twins_are_byte_identical():
    import ctypes as C
    lib = O.load()
    for cfg, (n_het, n_total, P) in {2: (128, 128, 1000), 3: (488, 1512, 10000)}.items():
        seed = synth.SEED + cfg
        cpu, mem, pods, cls = synth.gen_nodes(seed, n_het, n_total)
        c2, m2 = np.zeros(n_total, np.int64), np.zeros(n_total, np.int64)
        p2, k2 = np.zeros(n_total, np.int32), np.zeros(n_total, np.int32)
        lib.simon_oracle_gen_nodes(seed, n_het, n_total, capi._ptr(c2, C.c_int64), capi._ptr(m2, C.c_int64),
                                   capi._ptr(p2, C.c_int32), capi._ptr(k2, C.c_int32))
        assert (cpu == c2).all() and (mem == m2).all() and (pods == p2).all() and (cls == k2).all()
        pc, pm = synth.gen_pods(seed, P)
        q1, q2 = np.zeros(P, np.int64), np.zeros(P, np.int64)
        lib.simon_oracle_gen_pods(seed, P, capi._ptr(q1, C.c_int64), capi._ptr(q2, C.c_int64))
        assert (pc == q1).all() and (pm == q2).all()


def test_simon_raw_host_vs_oracle():
    """The host's Quantity float path (quantity.py) against the oracle's independent C restatement."""
    import ctypes as C
    lib = O.load()
    rng = np.random.default_rng(5)
    for _ in range(300):
        cpu_m = int(rng.choice([0, 50, 100, 250, 333, 1000, 1500, 7000]))
        mem_mi = int(rng.choice([0, 64, 100, 1000, 4096, 9216]))
        ncpu = int(rng.choice([1, 2, 4, 8, 64, 96]))
        nmem = int(rng.choice([1, 2, 16, 250, 512])) 
        req = {}
        if cpu_m: req["cpu"] = pq(f"{cpu_m}m")
        if mem_mi: req["memory"] = pq(f"{mem_mi}Mi")
        alloc = {"cpu": pq(str(ncpu)), "memory": pq(f"{nmem}Gi"), "pods": pq("110")}
        host = simon_raw_score(req, alloc)
        names = list(alloc)
        pr = (O.OQuantity * 3)(*[O.OQuantity(req.get(n, pq(0)).value, req.get(n, pq(0)).scale) for n in names])
        na = (O.OQuantity * 3)(*[O.OQuantity(alloc[n].value, alloc[n].scale) for n in names])
        assert lib.simon_oracle_simon_raw(pr, na, 3, 1 if req else 0) == host


@pytest.mark.parametrize("seed", range(4))
def test_oracle_invariants_random(seed):
    """Domain invariants on random problems: no node over-committed by scheduled pods, used == sum of placed."""
    prob = randprob.rand_problem(seed, N=30, P=150, eph=True, scalars=2, static_mask=True, tight_pods=True)
    scen, orders = randprob.rand_scenarios(seed, prob)
    res = O.run(prob, scen, orders)
    for s in range(len(scen)):
        n = scen[s, 0]
        pl = res.placement[s]
        placed = pl >= 0
        assert (pl[placed] < n).all()
        cpu = np.bincount(pl[placed], weights=prob.req_cpu[placed], minlength=n)
        assert (cpu <= prob.alloc_cpu[:n]).all()
        cnt = np.bincount(pl[placed], minlength=n)
        assert (cnt <= prob.alloc_pods[:n]).all()
        assert res.used_cpu[s] == int(prob.req_cpu[placed].sum())
        assert res.unscheduled[s] == int((pl == capi.UNSCHEDULED).sum())
