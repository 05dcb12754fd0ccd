"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle, bit-exact."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_gpu(prob, scen, orders, env=None, want_placement=True):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders, want_placement)
            st = ctx.stats()
            return res, st.kernel_variant
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def assert_same(a: capi.BatchResult, b: capi.BatchResult):
    assert a.unscheduled.tolist() == b.unscheduled.tolist()
    if a.used_vg is not None and b.used_vg is not None:
        assert a.used_vg.tolist() == b.used_vg.tolist()
    assert a.used_cpu.tolist() == b.used_cpu.tolist()
    assert a.used_mem.tolist() == b.used_mem.tolist()
    if a.placement is not None and b.placement is not None:
        bad = np.argwhere(a.placement != b.placement)
        assert len(bad) == 0, f"{len(bad)} placements differ, first (scenario,pod)={bad[0].tolist()}: " \
                              f"{a.placement[tuple(bad[0])]} vs {b.placement[tuple(bad[0])]}"


def test_library_loads_and_sees_device():
    lib = capi.load_library()
    assert lib.simon_hip_version() == capi.ABI_VERSION
    assert lib.simon_hip_device_count() >= 1


def test_known_answer_vector_on_gpu():
    from test_oracle import kav_problem
    prob, _ = kav_problem()
    res, variant = run_gpu(prob, [[2, 0]], np.arange(2)[None])
    assert variant == capi.KERNEL_NARROW_CACHE
    assert res.placement.tolist() == [[0, 0]] and res.unscheduled.tolist() == [0]
    for env in ({"SIMON_NO_CACHE": "1"}, {"SIMON_NARROW_V1": "1"}, {"SIMON_FORCE_WIDE": "1"}):
        res, _ = run_gpu(prob, [[2, 0]], np.arange(2)[None], env=env)
        assert res.placement.tolist() == [[0, 0]] and res.unscheduled.tolist() == [0]


@pytest.mark.parametrize("homogeneous", [False, True])
def test_config2_single_scenario(homogeneous):
    prob, scen, orders = synth.config2(homogeneous)
    ref = O.run(prob, scen, orders)
    res, variant = run_gpu(prob, scen, orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_CACHE": "1"})
    assert variant == capi.KERNEL_NARROW_FAST
    assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NARROW_V1": "1"})
    assert variant == capi.KERNEL_NARROW
    assert_same(res, ref)


VARIANT_OF_ENV = {"": capi.KERNEL_NARROW_CACHE, "SIMON_NO_CACHE": capi.KERNEL_NARROW_FAST, "SIMON_NARROW_V1": capi.KERNEL_NARROW}


@pytest.mark.parametrize("env", [{}, {"SIMON_CACHE_BANDS": "1"}, {"SIMON_NO_CACHE": "1"}, {"SIMON_NARROW_V1": "1", "SIMON_RCP_DIV": "0"},
                                 {"SIMON_NARROW_V1": "1", "SIMON_RCP_DIV": "1"}])
def test_config3_full_size_subset(env):
    """10k pods x 488..1511 nodes: 4 orders x 6 node counts at full size, every placement compared."""
    prob, scen, orders = synth.config3()
    pick = [0, 1, 2, 3, 4 * 7 + 1, 4 * 100, 4 * 300 + 2, 4 * 511 + 3, 4 * 512, 4 * 800 + 1, 4 * 1023 + 2, 4 * 1023 + 3]
    sub = scen[pick]
    ref = O.run(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders, env=env)
    want = capi.KERNEL_NARROW if "SIMON_NARROW_V1" in env else capi.KERNEL_NARROW_FAST if "SIMON_NO_CACHE" in env \
        else capi.KERNEL_NARROW_CACHE
    assert variant == want
    assert_same(res, ref)


def test_config3_homogeneous_ties():
    """All nodes identical: almost every pod ties on score, the first-max rule decides everything."""
    prob, scen, orders = synth.config3(n_counts=64, n_orders=4, n_pods=4000, n_het=100, homogeneous=True)
    sub = scen[::9]
    ref = O.run(prob, sub, orders)
    assert_same(run_gpu(prob, sub, orders)[0], ref)
    assert_same(run_gpu(prob, sub, orders, env={"SIMON_NO_CACHE": "1"})[0], ref)


@pytest.mark.parametrize("wg", ["64", "128", "256", "512"])
@pytest.mark.parametrize("v1", ["0", "1"])
def test_workgroup_shapes(wg, v1):
    prob, scen, orders = synth.config3(n_counts=40, n_orders=3, n_pods=1500, n_het=90)
    sub = scen[::5]
    assert_same(run_gpu(prob, sub, orders, env={"SIMON_WG": wg, "SIMON_NARROW_V1": v1, "SIMON_NO_CACHE": "1"})[0],
                O.run(prob, sub, orders))


def test_replica_runs_hit_the_score_cache():
    """Pods grouped by request signature (what a Deployment's replicas look like): the fast kernel re-evaluates
    only the node touched by the previous assume; results must still equal the oracle's."""
    prob, scen, orders = synth.config3(n_counts=16, n_orders=4, n_pods=3000, n_het=60)
    grouped = np.argsort(prob.pod_class, kind="stable").astype(np.int32)
    orders = np.stack([grouped, grouped[::-1].copy(), orders[1], orders[2]])
    ref = O.run(prob, scen, orders)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_CACHE": "1"})
    assert variant == capi.KERNEL_NARROW_FAST
    assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)


NARROW_FEATURES = [
    dict(), dict(nz_differs=True), dict(init_state=True, nz_differs=True), dict(static_mask=True),
    dict(presets=True), dict(gates=True, presets=True), dict(zero_pods=True, nz_differs=True), dict(tight_pods=True),
    dict(odd_units=True), dict(static_mask=True, init_state=True, presets=True, gates=True, tight_pods=True, zero_pods=True,
                               nz_differs=True),
    dict(pins=True), dict(pins=True, static_mask=True, presets=True, gates=True, tight_pods=True, nz_differs=True),
]


@pytest.mark.parametrize("idx", range(len(NARROW_FEATURES)))
@pytest.mark.parametrize("force_wide", ["0", "1"])
def test_random_cpu_mem_features(idx, force_wide):
    feat = NARROW_FEATURES[idx]
    for seed in range(3):
        prob = randprob.rand_problem(100 * idx + seed, N=70 + 37 * seed, P=400, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=8)
        ref = O.run(prob, scen, orders)
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_FORCE_WIDE": force_wide})
        if force_wide == "1":
            assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)
        if force_wide == "0":
            if "odd_units" not in feat:                                 # odd_units (gcd 1) exceeds 31 bits -> WIDE
                assert variant == capi.KERNEL_NARROW_CACHE
            res, variant = run_gpu(prob, scen, orders, env={"SIMON_NARROW_V1": "1"})
            assert variant in (capi.KERNEL_NARROW, capi.KERNEL_WIDE)
            assert_same(res, ref)
            res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_CACHE": "1"})
            assert variant in (capi.KERNEL_NARROW_FAST, capi.KERNEL_NARROW, capi.KERNEL_WIDE)
            assert_same(res, ref)


WIDE_FEATURES = [
    dict(eph=True), dict(scalars=2), dict(gpu=True), dict(anti=True), dict(gpu=True, init_state=True),
    dict(eph=True, scalars=3, gpu=True, anti=True, static_mask=True, init_state=True, presets=True, gates=True,
         nz_differs=True, zero_pods=True, tight_pods=True),
    dict(gpu=True, anti=True, static_mask=True, pins=True, presets=True),
]


@pytest.mark.parametrize("idx", range(len(WIDE_FEATURES)))
def test_random_wide_features(idx):
    feat = WIDE_FEATURES[idx]
    for seed in range(3):
        prob = randprob.rand_problem(1000 + 100 * idx + seed, N=50 + 41 * seed, P=350, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=6)
        ref = O.run(prob, scen, orders)
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_REST": "1"})
        assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)
        res, variant = run_gpu(prob, scen, orders)          # GPU share, ephemeral storage, extended resources: generation 6 may take it
        if set(feat) <= {"gpu"}:
            assert variant == capi.KERNEL_NARROW_CACHE
        assert_same(res, ref)


REST_FEATURES = [
    dict(gpu=True), dict(anti_host=True), dict(gpu=True, anti_host=True), dict(gpu=True, anti_host=True, init_state=True, static_mask=True),
    dict(gpu=True, anti_host=True, presets=True, gates=True, pins=True, tight_pods=True, zero_pods=True, nz_differs=True, static_mask=True),
    dict(anti_host=True, pins=True, presets=True, tight_pods=True),
    dict(eph=True), dict(scalars=2), dict(eph=True, scalars=4, zero_pods=True, tight_pods=True),
    dict(ports=True), dict(ports=True, anti_host=True, gpu=True, presets=True, pins=True, static_mask=True),
    dict(anti=True), dict(anti=True, gpu=True, presets=True, pins=True, gates=True, static_mask=True, tight_pods=True),
    dict(static_small=True), dict(static_small=True, static_mask=True, gpu=True, anti_host=True, tight_pods=True, pins=True),
    dict(aff=True), dict(aff=True, anti=True, gpu=True, presets=True, pins=True, static_mask=True, tight_pods=True),
    dict(eph=True, scalars=3, gpu=True, anti_host=True, static_mask=True, zero_pods=True, tight_pods=True, gates=True, pins=True, nz_differs=True),
]


@pytest.mark.parametrize("idx", range(len(REST_FEATURES)))
def test_random_rest_features(idx, monkeypatch):
    """Open-Gpu-Share, required anti-affinity on node-level topology keys, ephemeral storage and extended resources on the score-table
    kernel (generation 6): position masks per block, per-class best + NormalizeScore over the classes that kept a node."""
    feat = REST_FEATURES[idx]
    monkeypatch.setenv("SIMON_NO_FOLD", "1")               # (anti-affinity / ports alone would be folded into the table: the test below)
    monkeypatch.setenv("SIMON_NO_GPU_FOLD", "1")           # (GPU share alone, with few signatures, likewise: tests/test_gpu_round4.py)
    for seed in range(4):
        N = [37, 150, 700, 1500][seed]
        prob = randprob.rand_problem(7000 + 100 * idx + seed, N=N, P=500 + 300 * seed, n_pod_classes=6 + 5 * seed, n_node_classes=3 + 2 * seed, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=5, min_n=1 if seed == 0 else None)
        ref = O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.run_loaded(True)
            res, st = ctx.fetch(True), ctx.stats()
        if feat.get("scalars", 0) <= 2:                    # more extended resources: more than 32 distinct requests may appear
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == (6 if set(feat) - {"static_small"} else 4)
        assert_same(res, ref)


FOLD_FEATURES = [dict(anti_host=True), dict(ports=True), dict(anti_host=True, ports=True),
                 dict(anti_host=True, pins=True, presets=True, tight_pods=True), dict(ports=True, anti_host=True, presets=True, gates=True, static_mask=True),
                 dict(anti_host=True, static_small=True, nz_differs=True, init_state=True, zero_pods=True)]


@pytest.mark.parametrize("idx", range(len(FOLD_FEATURES)))
def test_anti_affinity_and_ports_folded_into_the_score_table(idx, monkeypatch):
    """Required anti-affinity and host ports on node-level topology keys WITHOUT GPU share: a landing pod zeroes the node's byte of every
    signature it excludes there (monotone infeasibility: simon_hip.hip fold_supported) -- generations 4 / 5, no position masks; the same
    problems through the position masks (SIMON_NO_FOLD=1: generation 6)."""
    feat = FOLD_FEATURES[idx]
    for seed in range(4):
        N = [37, 150, 700, 1500][seed]
        prob = randprob.rand_problem(7900 + 100 * idx + seed, N=N, P=500 + 300 * seed, n_pod_classes=6 + 5 * seed, n_node_classes=3 + 2 * seed, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=5, min_n=1 if seed == 0 else None)
        ref = O.run_threaded(prob, scen, orders)
        for no_fold in (False, True):
            if no_fold:
                monkeypatch.setenv("SIMON_NO_FOLD", "1")
            else:
                monkeypatch.delenv("SIMON_NO_FOLD", raising=False)
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                ctx.load_scenarios(scen, orders)
                ctx.run_loaded(True)
                res, st = ctx.fetch(True), ctx.stats()
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == (6 if no_fold else st.kernel_generation)
            assert (st.kernel_generation in (4, 5)) != no_fold, (st.kernel_generation, no_fold)
            assert_same(res, ref)


V2_FEATURES = [
    dict(local=True), dict(local=True, init_state=True, presets=True, gates=True, tight_pods=True, static_mask=True),
    dict(aff=True), dict(ipa=True), dict(spread_hard=True), dict(spread_soft=True), dict(static_scores=True),
    dict(aff=True, anti=True, ipa=True), dict(spread_hard=True, spread_soft=True, static_scores=True, static_mask=True),
    dict(anti=True, aff=True, ipa=True, spread_hard=True, spread_soft=True, static_scores=True, gpu=True, eph=True,
         scalars=2, presets=True, gates=True, tight_pods=True, static_mask=True, nz_differs=True, init_state=True,
         zero_pods=True, local=True),
    dict(pins=True, spread_hard=True, spread_soft=True, ipa=True, static_mask=True, static_scores=True),
]


@pytest.mark.parametrize("idx", range(len(V2_FEATURES)))
@pytest.mark.parametrize("wg", ["", "64", "1024"])
def test_random_v2_features(idx, wg):
    """ABI v2 plugins (required affinity, InterPodAffinity scores, PodTopologySpread filter + score, NodeAffinity /
    TaintToleration preferred, NodePreferAvoidPods) on random inputs, one-wave and multi-wave workgroups."""
    feat = V2_FEATURES[idx]
    for seed in range(3):
        prob = randprob.rand_problem(3000 + 100 * idx + seed, N=50 + 41 * seed, P=350, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=6)
        ref = O.run(prob, scen, orders)
        env = {"SIMON_WG": wg} if wg else {}
        if set(feat) <= {"aff", "anti"}:                    # required (anti-)affinity alone fits the score-table kernel: keep the
            env["SIMON_NO_REST"] = "1"                      # all-feature kernel's coverage of it here
        if set(feat) <= {"spread_soft"}:                    # soft spread constraints alone likewise (generation 7, tests/test_gpu_round3.py)
            env["SIMON_NO_SPREAD"] = "1"
        res, variant = run_gpu(prob, scen, orders, env=env or None)
        assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)


@pytest.mark.parametrize("wg", ["64", "256", "512"])
def test_random_v2_features_many_nodes(wg):
    """Every ABI v2 plugin at once on a pool large enough that a lane owns several nodes and several load batches
    (700 nodes: 11 nodes per lane at 64 threads), with gated / preset pods and prefix scenarios."""
    feat = dict(anti=True, aff=True, ipa=True, spread_hard=True, spread_soft=True, static_scores=True, gpu=True, eph=True,
                presets=True, gates=True, static_mask=True, nz_differs=True, init_state=True, local=True)
    for seed in (4240, 4254):                     # ~80 % of the pods schedulable: long assume histories
        prob = randprob.rand_problem(seed, N=700, P=1200, n_pod_classes=14, n_node_classes=9, **feat)
        scen, orders = randprob.rand_scenarios(5, prob, S=4, min_n=400)
        ref = O.run(prob, scen, orders)
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_WG": wg})
        assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)
        res, _ = run_gpu(prob, scen, orders, env={"SIMON_WG": wg, "SIMON_WIDE_NO_TABLE": "1"})     # full per-node evaluation path
        assert_same(res, ref)


def test_v2_hand_cases_on_gpu():
    """The hand-derived cases of tests/test_oracle_v2.py through the HIP path."""
    import test_oracle_v2 as T
    probs = []
    orig = T.zero_pods_problem

    def capture(*a, **kw):
        pr = orig(*a, **kw)
        probs.append(pr)
        return pr
    T.zero_pods_problem = capture
    try:
        for name in ("test_spread_soft_upstream_vector_hostname", "test_spread_soft_two_constraints_zone_and_hostname",
                     "test_spread_soft_ignored_nodes_and_maxskew_offset", "test_spread_soft_counts_only_nodes_of_the_term_node_set",
                     "test_spread_hard_filter_skew_and_missing_label", "test_spread_hard_minimum_only_over_registered_domains",
                     "test_required_affinity_first_pod_escape_then_colocation",
                     "test_interpod_affinity_score_incoming_and_symmetric_terms",
                     "test_node_affinity_and_taint_prefer_normalisation_and_static_add"):
            getattr(T, name)()
    finally:
        T.zero_pods_problem = orig
    assert len(probs) >= 9
    for pr in probs:
        n, P = pr.n_nodes, pr.n_pods
        scen, orders = [[n, 0]], np.arange(P, dtype=np.int32)[None]
        assert_same(run_gpu(pr, scen, orders)[0], O.run(pr, scen, orders))


def test_explain_v2_codes_match_oracle():
    prob = randprob.rand_problem(3777, N=30, P=300, anti=True, aff=True, spread_hard=True, spread_soft=True, ipa=True,
                                 static_mask=True, tight_pods=True, gpu=True)
    scen, orders = randprob.rand_scenarios(7, prob, S=2)
    ref, (nf, failed, codes) = O.run(prob, scen[:1], orders, explain_scenario=0, max_failed=24)
    assert nf > 0
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        n, f2, c2 = ctx.explain(int(scen[0, 0]), orders[scen[0, 1]], max_failed=24)
    assert n == nf and f2.tolist() == failed.tolist()
    assert (c2 == codes).all()


def test_explain_on_a_narrow_problem():
    """simon_explain stages the all-feature kernel lazily when the batch itself runs on a NARROW kernel."""
    prob = randprob.rand_problem(11, N=20, P=400, tight_pods=True)
    scen, orders = randprob.rand_scenarios(3, prob, S=2)
    ref, (nf, failed, codes) = O.run(prob, scen[:1], orders, explain_scenario=0, max_failed=16)
    assert nf > 0
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders)
        assert ctx.stats().kernel_variant == capi.KERNEL_NARROW_CACHE
        n, f2, c2 = ctx.explain(int(scen[0, 0]), orders[scen[0, 1]], max_failed=16)
    assert n == nf and f2.tolist() == failed.tolist() and (c2 == codes).all()


def test_explain_on_a_generation_6_problem(monkeypatch):
    """GPU-share / anti-affinity failure codes of a batch that ran on the score-table kernel's REST path: simon_explain and
    simon_explain_loaded stage the all-feature kernel lazily, the group API shards the same batch over two contexts."""
    monkeypatch.setenv("SIMON_NO_GPU_FOLD", "1")           # (with few signatures GPU share + hostname anti-affinity would both be folded into the table)
    prob = randprob.rand_problem(7421, N=40, P=500, gpu=True, anti_host=True, tight_pods=True, static_mask=True)
    scen, orders = randprob.rand_scenarios(3, prob, S=4)
    ref, (nf, failed, codes) = O.run(prob, scen[:1], orders, explain_scenario=0, max_failed=32)
    assert nf > 0 and ((codes & 0xF000) == capi.FAIL_GPUSHARE).any()
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders)
        assert ctx.stats().kernel_generation == 6
        assert_same(res, O.run(prob, scen, orders))
        n, f2, c2 = ctx.explain(int(scen[0, 0]), orders[scen[0, 1]], max_failed=32)
        assert n == nf and f2.tolist() == failed.tolist() and (c2 == codes).all()
        n, f2, c2 = ctx.explain_loaded(0, max_failed=32)
        assert n == nf and f2.tolist() == failed.tolist() and (c2 == codes).all()
    with capi.Group([0, 0]) as g:
        g.load_problem(prob)
        assert_same(g.run_batch(scen, orders), O.run(prob, scen, orders))


def test_k8s_fixtures_on_gpu():
    """YAML-level cases (reference example/ inputs and random clusters, tests/golden/k8s_*.json): the HIP path must
    reproduce the placements of the independent object-level scheduler (tests/pyref_sched.py), no oracle in the loop."""
    import test_host_mirror as H
    assert len(H.K8S_FIXTURES) >= 8
    for name in H.K8S_FIXTURES:
        d, prob = H.load_k8s_fixture(name)
        P, n = d["n_pods"], d["n_nodes"]
        res, variant = run_gpu(prob, [[n, 0]], np.arange(P, dtype=np.int32)[None])
        assert res.placement[0].tolist() == d["object_level_placement"], name


def test_k8s_simulate_and_sweep_through_the_hip_engine():
    """simulate() / sweep() with the product engine against the object-level scheduler and against the oracle engine."""
    import pyref_sched
    import test_host_mirror as H
    from open_simulator_amd import flatten as fl, simulate as sim
    import randk8s
    for seed in (2, 9, 21):
        nodes, pods, services, rs = H._random_case(seed, gpu=(seed % 3 == 0), n_nodes=20, n_workloads=16, local=True)
        flat = fl.flatten(nodes, pods, services, rs, [], storage_classes=randk8s.STORAGE_CLASSES)
        res, _ = run_gpu(flat.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
        ref = pyref_sched.Scheduler(nodes, services, rs, [], randk8s.STORAGE_CLASSES).run(pods)
        assert [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()] == ref
    for seed in (5, 14):          # ImageLocality as a static score (nodes list the image the pods run)
        nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=9, n_workloads=10)
        nodes = H._with_images([nodes[j] for j in H.k8s.canonical_node_order(nodes)], seed)
        cluster = {k: [] for k in H.k8s.KINDS}
        cluster["Node"], cluster["Service"] = nodes, services
        pods, _ = sim.build_stream(cluster, [sim.AppResource("app", H.k8s.group_resources(workloads))], nodes, len(nodes))
        flat = fl.flatten(nodes, pods, services, [], [], image_total=len(nodes))
        res, _ = run_gpu(flat.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
        ref = pyref_sched.Scheduler(nodes, services, [], []).run(pods)
        assert [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()] == ref
    # the add-nodes search of test_host_mirror.test_sweep_finds_the_minimum_node_count on the GPU
    H_test = H.test_sweep_finds_the_minimum_node_count
    orig = H.OracleEngine
    H.OracleEngine = sim.HipEngine
    try:
        H_test()
        H.test_unscheduled_reason_text()
    finally:
        H.OracleEngine = orig


def test_k8s_sweep_at_scale_matches_oracle():
    """The general path at a size where every code path is busy (400 nodes incl. 100 new, ~1 200 pods from 60 random
    workloads + a DaemonSet, 8 cluster sizes): `simulate.sweep` on the HIP engine, every placement against the oracle."""
    import randk8s
    from open_simulator_amd import k8s, simulate as sim, workloads as wl
    nodes, workloads, services = randk8s.rand_cluster(1, n_nodes=300, n_workloads=60, max_replicas=30)
    for j, n in enumerate(nodes):
        n["metadata"]["labels"][randk8s.ZONE] = f"z{j % 3}"
    ds = {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "agent", "namespace": "kube-system"},
          "spec": {"selector": {"matchLabels": {"app": "agent"}},
                   "template": {"metadata": {"labels": {"app": "agent"}},
                                "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "100m", "memory": "128Mi"}}}],
                                         "tolerations": [{"operator": "Exists"}]}}}}
    cluster = k8s.group_resources(nodes + services + [ds])
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "z0"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "40"}, "capacity": {"cpu": "32", "memory": "64Gi"}}}

    class Recording(sim.HipEngine):
        def run(self, prob, scen, orders, want_placement=True, **kw):
            self.args = (prob, scen, orders)
            self.out = super().run(prob, scen, orders, want_placement, **kw)
            return self.out

    eng = Recording()
    sim.sweep(cluster, apps, template, [0, 14, 28, 42, 57, 71, 85, 100], engine=eng)
    prob, scen, orders = eng.args
    assert prob.n_pods > 1000 and len(scen) == 8
    assert_same(eng.out, O.run(prob, scen, orders))


def _random_ranks(rng, scen, N):
    ranks = np.zeros((len(scen), N), np.int32)
    for s, (n, _) in enumerate(np.asarray(scen).tolist()):
        ranks[s, :n] = rng.permutation(n)
    return ranks


@pytest.mark.parametrize("feat", [dict(), dict(static_mask=True, presets=True, gates=True, pins=True),
                                  dict(gpu=True, anti=True, static_mask=True), dict(gpu=True, anti_host=True, ports=True, presets=True, pins=True),
                                  dict(ipa=True, spread_soft=True, spread_hard=True, aff=True, static_scores=True, static_mask=True),
                                  dict(spread_soft=True), dict(spread_soft=True, static_mask=True, presets=True, gates=True, pins=True)])
def test_per_scenario_node_ranks(feat):
    """simon_set_node_ranks: the tie-break of selectHost follows the scenario's own canonical node order.  Homogeneous
    pools (every score ties) make the ranks decide almost every placement."""
    rng = np.random.default_rng(5)
    for seed, homogeneous in ((0, True), (1, False)):
        prob = randprob.rand_problem(6000 + seed, N=60 + 50 * seed, P=300, **feat)
        if homogeneous:
            prob.alloc_cpu[:] = prob.alloc_cpu[0]; prob.alloc_mem[:] = prob.alloc_mem[0]; prob.alloc_pods[:] = prob.alloc_pods[0]
            prob.node_class[:] = 0
        scen, orders = randprob.rand_scenarios(seed, prob, S=6)
        ranks = _random_ranks(rng, scen, prob.n_nodes)
        ref = O.run(prob, scen, orders, node_ranks=ranks)
        assert (ref.placement != O.run(prob, scen, orders).placement).any()          # the ranks matter on this input
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            # cpu+memory problems keep the score-table kernel (per-scenario class lists in rank order); the rest: all-feature kernel
            narrow = set(feat) <= {"static_mask", "presets", "gates", "pins", "gpu", "anti", "anti_host", "ports"} or \
                set(feat) <= {"spread_soft", "static_mask", "presets", "gates", "pins"}      # (generation 7 keeps the rank of every position)
            assert ctx.stats().kernel_variant == (capi.KERNEL_NARROW_CACHE if narrow else capi.KERNEL_WIDE)
            assert_same(ctx.fetch(True), ref)
            ctx.set_node_ranks(None)                                                  # back to pool order
            ctx.run_loaded(True)
            assert_same(ctx.fetch(True), O.run(prob, scen, orders))


def test_k8s_sweep_over_several_zones_on_gpu():
    """tests/test_host_mirror.py::test_sweep_over_several_zones_... with the product engine: one batch with per-scenario
    nodeTree ranks against one Simulate() per size."""
    import test_host_mirror as H
    from open_simulator_amd import simulate as sim
    orig = H.OracleEngine
    H.OracleEngine = sim.HipEngine
    try:
        H.test_sweep_over_several_zones_runs_every_size_on_its_own()
    finally:
        H.OracleEngine = orig


def test_config5_gpushare_style():
    """BASELINE config 5 shape (GPU share + required anti-affinity + taints): small pool, every placement compared;
    then ONE scenario at full size (50k pods x 5k nodes)."""
    prob, scen, orders = synth.config5(n_pods=5000, n_nodes=500, n_scen=16, n_orders=2, n_groups=10, group_size=50)
    ref = O.run(prob, scen, orders)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_REST": "1"})
    assert variant == capi.KERNEL_WIDE
    assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders)
    assert variant == capi.KERNEL_NARROW_CACHE              # generation 6: the score table + per-node filters
    assert_same(res, ref)
    assert ref.unscheduled.max() > 0 and ref.unscheduled.min() == 0        # the sweep crosses the feasibility boundary
    prob, scen, orders = synth.config5()
    sub = scen[[len(scen) // 2 + 1]]
    assert_same(run_gpu(prob, sub, orders)[0], O.run(prob, sub, orders))


def test_gpushare_example_and_hand_cases_on_gpu():
    from test_oracle import gpushare_problem
    prob = gpushare_problem()
    assert_same(run_gpu(prob, [[2, 0]], np.arange(9)[None])[0], O.run(prob, [[2, 0]], np.arange(9)[None]))


def test_explain_codes_match_oracle():
    prob = randprob.rand_problem(7, N=24, P=300, eph=True, scalars=2, gpu=True, anti=True, static_mask=True,
                                 tight_pods=True, pins=True)
    scen, orders = randprob.rand_scenarios(7, prob, S=2)
    ref, (nf, failed, codes) = O.run(prob, scen[:1], orders, explain_scenario=0, max_failed=16)
    assert nf > 0
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        n, f2, c2 = ctx.explain(int(scen[0, 0]), orders[scen[0, 1]], max_failed=16)
    assert n == nf and f2.tolist() == failed.tolist()
    assert (c2 == codes).all()


def test_min_plan_matches_oracle():
    prob, scen, orders = synth.config3(n_counts=24, n_orders=2, n_pods=2500, n_het=110)
    ref = O.run(prob, scen, orders, want_placement=False)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders, want_placement=False)
        assert_same(res, ref)
        for caps in ((100, 100), (60, 100), (100, 50), (1, 1)):
            a, b = ctx.min_plan(*caps), O.min_plan(prob, scen, ref, *caps)
            assert a.as_dict() == b.as_dict(), caps


def test_min_plan_with_the_vg_cap_matches_oracle():
    """MaxVG (pkg/apply/apply.go:712-771) on the device: Open-Local problem, node-count sweep, caps from loose to strict."""
    prob = randprob.rand_problem(7, N=120, P=30, local=True, tight_pods=False)
    scen = np.array([[n, 0] for n in range(16, 121, 8)], np.int32)
    orders = np.arange(prob.n_pods, dtype=np.int32)[None]
    ref = O.run(prob, scen, orders, want_placement=False)
    seen = set()
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders, want_placement=False)
        assert_same(res, ref)
        for vg_cap in (100, 30, 8, 7, 6, 5, 4, 0):
            (a, va), (b, vb) = ctx.min_plan_vg(100, 100, vg_cap), O.min_plan_vg(prob, scen, ref, 100, 100, vg_cap)
            assert a.as_dict() == b.as_dict() and va == vb, vg_cap
            seen.add((a.found, a.n_nodes))
        assert ctx.min_plan(100, 100).as_dict() == O.min_plan(prob, scen, ref).as_dict()
    assert len(seen) >= 2, "the caps never changed the answer: the test does not exercise MaxVG"


def test_size_independent_properties_full_batch():
    """Whole config-3 batch (4096 scenarios): properties that hold without the oracle."""
    prob, scen, orders = synth.config3()
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        res = ctx.fetch(want_placement=False)
        total_cpu, total_mem = int(prob.req_cpu.sum()), int(prob.req_mem.sum())
        full = res.unscheduled == 0
        assert full.any()
        assert (res.used_cpu[full] == total_cpu).all() and (res.used_mem[full] == total_mem).all()
        assert (res.used_cpu <= total_cpu).all()
        for s in (0, 1, 2047, 4095):
            pl = ctx.fetch_placement(s)
            n = scen[s, 0]
            placed = pl >= 0
            assert (pl[placed] < n).all() and int((pl == capi.UNSCHEDULED).sum()) == res.unscheduled[s]
            assert (np.bincount(pl[placed], weights=prob.req_cpu[placed], minlength=n) <= prob.alloc_cpu[:n]).all()
            assert (np.bincount(pl[placed], weights=prob.req_mem[placed], minlength=n) <= prob.alloc_mem[:n]).all()
            assert res.used_cpu[s] == int(prob.req_cpu[placed].sum())
        # idempotence: a second run of the loaded batch gives identical results
        ctx.run_loaded(want_placement=False)
        again = ctx.fetch(want_placement=False)
        assert_same(again, res)


def test_golden_fixtures_on_gpu():
    """HIP path against the committed fixtures (tests/golden/), no oracle in the loop."""
    import golden_util as G
    for name in G.SMALL:
        d, prob, scen, orders = G.load(name)
        for env in ({}, {"SIMON_NO_CACHE": "1"}, {"SIMON_FORCE_WIDE": "1"}):
            G.check_against(d, run_gpu(prob, scen, orders, env=env)[0])
    dg = G.digests()
    for hom in (False, True):
        prob, scen, orders = synth.config2(hom)
        r, _ = run_gpu(prob, scen, orders)
        want = dg[f"config2_{'homogeneous' if hom else 'heterogeneous'}"]
        assert r.unscheduled.tolist() == want["unscheduled"] and G.sha(r.placement) == want["placement_sha256"]
    prob, scen, orders = synth.config3()
    want = dg["config3_subset"]
    r, variant = run_gpu(prob, scen[want["pick"]], orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert r.unscheduled.tolist() == want["unscheduled"] and r.used_cpu.tolist() == want["used_cpu"]
    assert [G.sha(row) for row in r.placement] == want["placement_sha256"]


def test_randomised_feature_sweep():
    """A slice of tests/fuzz_gpu.py (random feature subsets, sizes, node ranks, workgroup sizes) in the regular suite."""
    import fuzz_gpu
    old = sys.argv
    sys.argv = ["fuzz_gpu.py", "80", "5000"]
    try:
        assert fuzz_gpu.main() == 0
    finally:
        sys.argv = old


def test_errors_are_reported_not_thrown():
    prob, scen, orders = synth.config2()
    with capi.Context(0) as ctx:
        with pytest.raises(capi.SimonError):
            ctx.problem = prob
            ctx.load_scenarios(scen, orders)        # nothing loaded yet
        ctx.load_problem(prob)
        with pytest.raises(capi.SimonError):
            ctx.run_batch([[prob.n_nodes + 1, 0]], orders)
        with pytest.raises(capi.SimonError):
            ctx.run_batch([[5, 3]], orders)
        # node ranks must be a permutation of 0..n-1 per scenario; they need loaded scenarios
        ctx.load_scenarios([[6, 0], [9, 0]], orders)
        ranks = np.zeros((2, prob.n_nodes), np.int32)
        ranks[0, :6] = np.arange(6)[::-1]
        ranks[1, :9] = [0, 1, 2, 3, 4, 5, 6, 7, 7]             # 7 twice, 8 missing
        with pytest.raises(capi.SimonError, match="permutation"):
            ctx.set_node_ranks(ranks)
        ranks[1, :9] = np.arange(9)
        ctx.set_node_ranks(ranks)
        ctx.run_loaded(True)
    # a pinned pod must name a pool node and cannot also be bound
    bad = randprob.rand_problem(1, N=10, P=20)
    bad.pin_node = np.full(20, -1, np.int32)
    bad.pin_node[3] = 10
    with capi.Context(0) as ctx:
        with pytest.raises(capi.SimonError, match="pin_node"):
            ctx.load_problem(bad)


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py's world-size > 1 path (sharding, per-step plan all-gather, max-over-ranks timing) on a single-GPU box:
    two ranks on device 0 over gloo.  The driver's real multi-GPU runs use one rank per GPU over RCCL."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, SIMON_BENCH_BACKEND="gloo", SIMON_BENCH_SHARE_DEVICE="1", SIMON_BENCH_DETAIL=str(tmp_path / "bench_detail.json"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--counts", "32", "--pods", "2000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]                    # the LAST line is the record the driver parses
    assert len(line) <= 4096
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["scenarios_per_gpu"] == 32 * 4 and d["value"] > 0
    assert "8 pod orders = 256 scenarios" in d["config"]["workload"]
    assert d["config"]["plan"][0] >= 488 or d["config"]["plan"][0] == -1
