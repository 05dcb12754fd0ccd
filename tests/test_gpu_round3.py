"""GPU parity tests, round 3: the limits of the score-table kernel hit exactly (64 internal node classes), EVERY benchmarked
config-5 scenario, and the round's ABI additions."""
import os

import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, synth
from test_gpu_parity import assert_same, run_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("coarse", ["0", "1"])
@pytest.mark.parametrize("n_classes", [63, 64, 65, 100, 128])
def test_exactly_64_internal_node_classes(n_classes, coarse, monkeypatch):
    """with exactly 64 internal node classes no lane `Cn` exists, so the class-segment sentinel s_seg[Cn]
    was never written and class_of_pos read stale LDS.  16 caller classes x 4 allocatable shapes, every pair present."""
    monkeypatch.setenv("SIMON_TABLE_COARSE", coarse)
    rng = np.random.default_rng(640 + n_classes)
    N, P = 900, 1500
    # (round 4: 65 .. 128 classes -- two per lane in the instantiations without REST rows / SPREAD walks -- stay on the score table too)
    prob = randprob.rand_problem(6400 + n_classes, N=N, P=P, n_node_classes=32, n_pod_classes=12, tight_pods=True)
    shapes_c = np.array([4000, 8000, 16000, 32000])
    shapes_m = np.array([8, 16, 64, 128]) << 30
    pair = np.concatenate([np.arange(n_classes), rng.integers(0, n_classes, N - n_classes)])
    rng.shuffle(pair)
    prob.node_class = (pair // 4).astype(np.int32)
    prob.alloc_cpu = shapes_c[pair % 4].astype(np.int64)
    prob.alloc_mem = shapes_m[pair % 4].astype(np.int64)
    assert len({(a, b, c) for a, b, c in zip(prob.node_class.tolist(), prob.alloc_cpu.tolist(), prob.alloc_mem.tolist())}) == n_classes
    scen, orders = randprob.rand_scenarios(64, prob, S=6)
    scen[0, 0] = N                                   # one scenario holds every class
    ref = O.run_threaded(prob, scen, orders)
    for _ in range(3):                               # stale LDS differs from launch to launch
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.run_loaded(True)
            st = ctx.stats()
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation in (4, 5)
            assert_same(ctx.fetch(True), ref)


def test_config5_every_benchmarked_scenario():
    """BASELINE config 5 at full size (50 000 pods x 2 500..5 000 nodes; GPU share + anti-affinity + taints): ALL 256
    benchmarked scenarios on generation 6 of the score-table kernel, every placement row against the oracle (threaded:
    one scenario costs the oracle about a second per host thread)."""
    prob, scen, orders = synth.config5()
    assert len(scen) == 256
    ref = O.run_threaded(prob, scen, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        st = ctx.stats()
        assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 6
        res = ctx.fetch(want_placement=False)
        assert res.unscheduled.tolist() == ref.unscheduled.tolist()
        assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()
        bad_rows = []
        for s in range(len(scen)):
            row = ctx.fetch_placement(s)
            bad = np.flatnonzero(row != ref.placement[s])
            if len(bad):
                bad_rows.append((s, len(bad), int(bad[0])))
        assert not bad_rows, f"{len(bad_rows)} scenarios differ, first (scenario, count, pod) = {bad_rows[0]}"


@pytest.mark.parametrize("n_sigs", [129, 200, 300, 384, 385, 700, 1023])
def test_config3_beyond_128_signatures_stays_on_the_score_table(n_sigs, monkeypatch):
    """129 request signatures used to fall to generation 2 (8x slower).  The score-table kernel now keeps two
    signatures per lane in registers and refreshes further groups of 128 from TableCold::sigs (K <= 384 in one round trip per cycle, up to 1 023
    with one more per group since round 4; beyond 256 the host
    prefers generation 2 where that is eligible -- SIMON_FORCE_TABLE keeps the table for this test)."""
    monkeypatch.setenv("SIMON_FORCE_TABLE", "1")
    prob, scen, orders = synth.config3(n_counts=40, n_orders=3, n_pods=5000, n_sigs=n_sigs)
    sub = scen[::4]
    ref = O.run_threaded(prob, sub, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(sub, orders)
        ctx.run_loaded(True)
        st = ctx.stats()
        assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation in (4, 5), "must stay on the score-table kernel"
        assert_same(ctx.fetch(True), ref)


@pytest.mark.parametrize("n_sigs", [300, 385])
def test_many_signatures_fall_back_correctly(n_sigs):
    """Without the knob: from 257 signatures on a plain cpu+memory problem takes generation 2 (faster there)."""
    prob, scen, orders = synth.config3(n_counts=12, n_orders=2, n_pods=4000, n_sigs=n_sigs)
    ref = O.run_threaded(prob, scen, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders)
        assert ctx.stats().kernel_generation == 2
    assert_same(res, ref)


SPREAD_FEATURES = [dict(), dict(static_mask=True), dict(presets=True, gates=True), dict(pins=True, tight_pods=True), dict(nz_differs=True, init_state=True),
                   dict(zero_pods=True, odd_units=True), dict(static_small=True)]


@pytest.mark.parametrize("idx", range(len(SPREAD_FEATURES)))
def test_soft_spread_constraints_on_the_score_table_kernel(idx):
    """Generation 7: pods with soft PodTopologySpread constraints (hostname-like + zone-like keys, node sets, several constraints per
    class, unlabeled nodes = IgnoredNodes) stay on the score-table kernel; every placement against the oracle, and the same problems on
    the all-feature kernel (SIMON_NO_SPREAD=1)."""
    feat = SPREAD_FEATURES[idx]
    for seed, (N, P) in enumerate([(40, 300), (200, 900), (700, 1500), (1300, 2500)]):
        prob = randprob.rand_problem(7000 + 10 * idx + seed, N=N, P=P, spread_soft=True, n_node_classes=4, n_pod_classes=9, **feat)
        scen, orders = randprob.rand_scenarios(70 + seed, prob, S=5)
        ref = O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
        if "odd_units" not in feat:          # gcd-1 quantities leave the 31-bit fast path altogether: all-feature kernel, parity only
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 7, (st.kernel_variant, st.kernel_generation)
        assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_SPREAD": "1"})
    assert variant == capi.KERNEL_WIDE
    assert_same(res, ref)


def test_service_workload_full_size_subset():
    """config 3's pool with every pod selected by a Service (synth.config_service: 60 services, system-default soft spread constraints):
    a subset of the 4 096 benchmarked scenarios at full size (10 000 pods x 488..1 511 nodes), every placement, on generation 7; a
    smaller instance additionally against the all-feature kernel."""
    prob, scen, orders = synth.config_service()
    pick = np.unique(np.linspace(0, len(scen) - 1, 24).astype(int))
    ref = O.run_threaded(prob, scen[pick], orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(True)
        st = ctx.stats()
        assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 7
        res = ctx.fetch(False)
        assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist() and res.used_cpu[pick].tolist() == ref.used_cpu.tolist()
        for j, s_ in enumerate(pick.tolist()):
            row = ctx.fetch_placement(s_)
            bad = np.flatnonzero(row != ref.placement[j])
            assert len(bad) == 0, f"scenario {s_}: {len(bad)} placements differ, first pod {bad[0]}"
    prob, scen, orders = synth.config_service(n_counts=24, n_orders=2, n_pods=2500, n_het=200, n_services=25)
    ref = O.run_threaded(prob, scen, orders)
    assert_same(run_gpu(prob, scen, orders)[0], ref)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_SPREAD": "1"})
    assert variant == capi.KERNEL_WIDE
    assert_same(res, ref)


def test_k8s_deployments_behind_services_sweep_on_generation_7():
    """The object-level path of SURVEY.md 8(f) N1: Deployments / StatefulSets selected by Services (system-default soft spread
    constraints, plugin.go:39-50), some with their own ScheduleAnyway constraints, node selectors, required anti-affinity to their own
    replicas on the hostname key or the preferred kind (hostname + zone), on a cluster with three zones,
    unlabeled and tainted nodes -- `simulate.sweep` over eight cluster sizes in one batch (per-scenario nodeTree ranks) stays on the
    score-table kernel; every placement against the oracle."""
    import randk8s
    from open_simulator_amd import k8s, simulate as sim
    rng = np.random.default_rng(77)
    nodes = []
    for j in range(180):
        shape = [("8", "16Gi"), ("16", "32Gi"), ("32", "64Gi")][int(rng.integers(0, 3))]
        labels = {randk8s.HOST: f"node-{j}", "disk": ["ssd", "hdd"][j % 2]}
        if j % 11:
            labels[randk8s.ZONE] = f"z{j % 3}"
        node = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": f"node-{j}", "labels": labels},
                "status": {"allocatable": {"cpu": shape[0], "memory": shape[1], "pods": "60"}, "capacity": {"cpu": shape[0], "memory": shape[1]}}}
        if j % 13 == 5:
            node["spec"] = {"taints": [{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}]}
        nodes.append(node)
    workloads, services = [], []
    for w in range(40):
        app = f"app{w}"
        spec = {"containers": [{"name": "c", "image": "x", "resources": {"requests": {
            "cpu": str(rng.choice(["100m", "250m", "500m", "1"])), "memory": str(rng.choice(["128Mi", "512Mi", "1Gi", "2Gi"]))}}}]}
        r = w % 5
        if r == 1:
            spec["topologySpreadConstraints"] = [{"maxSkew": 2, "topologyKey": randk8s.ZONE, "whenUnsatisfiable": "ScheduleAnyway",
                                                  "labelSelector": {"matchLabels": {"app": app}}}]
        elif r == 2:            # zone before hostname: the general walk of spread_select
            spec["topologySpreadConstraints"] = [
                {"maxSkew": 1, "topologyKey": randk8s.ZONE, "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": {"matchLabels": {"app": app}}},
                {"maxSkew": 2, "topologyKey": randk8s.HOST, "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": {"matchLabels": {"app": app}}}]
        paa = {}
        if w % 6 == 5:          # one replica per node: required anti-affinity to the workload's own pods on the hostname key (folded into the table)
            paa["requiredDuringSchedulingIgnoredDuringExecution"] = [{"labelSelector": {"matchLabels": {"app": app}}, "topologyKey": randk8s.HOST}]
        if w % 10 == 9 or w in (13, 33):   # the chart default: prefer not to sit next to your own replicas (hostname 100, zone 50), with a Service
            paa["preferredDuringSchedulingIgnoredDuringExecution"] = [      # (9, 19, 29, 39: its constraints count on the zoned nodes only -- a harmless node set) or without
                {"weight": 100, "podAffinityTerm": {"labelSelector": {"matchLabels": {"app": app}}, "topologyKey": randk8s.HOST}},
                {"weight": 50, "podAffinityTerm": {"labelSelector": {"matchLabels": {"app": app}}, "topologyKey": randk8s.ZONE}}]
        if paa:
            spec["affinity"] = {"podAntiAffinity": paa}
        if w % 7 == 3:
            spec["nodeSelector"] = {"disk": "ssd"}
        if w % 9 == 4:
            spec["tolerations"] = [{"key": "dedicated", "operator": "Exists"}]
        kind = "StatefulSet" if w % 4 == 0 else "Deployment"
        workloads.append({"apiVersion": "apps/v1", "kind": kind, "metadata": {"name": app, "namespace": "default"},
                          "spec": {"replicas": int(rng.integers(5, 60)), "selector": {"matchLabels": {"app": app}},
                                   "template": {"metadata": {"labels": {"app": app}}, "spec": spec}}})
        if r != 3:
            services.append({"apiVersion": "v1", "kind": "Service", "metadata": {"name": f"svc-{w}", "namespace": "default"},
                             "spec": {"selector": {"app": app}}})
    cluster = k8s.group_resources(nodes + services)
    apps = [sim.AppResource("shop", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "z1"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "60"}, "capacity": {"cpu": "32", "memory": "64Gi"}}}

    class Recording(sim.HipEngine):
        def run(self, prob, scen, orders, want_placement=True, **kw):
            self.args, self.kw = (prob, scen, orders), kw
            self.out = super().run(prob, scen, orders, want_placement, **kw)
            return self.out

    eng = Recording()
    sim.sweep(cluster, apps, template, [0, 6, 12, 18, 24, 30, 36, 42], engine=eng)
    prob, scen, orders = eng.args
    assert prob.n_pods > 1000 and len(scen) == 8 and eng.kw.get("node_ranks") is not None
    assert eng.last_stats.kernel_variant == capi.KERNEL_NARROW_CACHE and eng.last_stats.kernel_generation == 7
    assert_same(eng.out, O.run(prob, scen, orders, node_ranks=eng.kw["node_ranks"]))


@pytest.mark.parametrize("feat", [dict(anti_host=True), dict(anti_host=True, static_mask=True, presets=True, gates=True, pins=True),
                                  dict(anti_host=True, nz_differs=True, init_state=True, tight_pods=True)])
def test_soft_spread_constraints_with_required_hostname_anti_affinity(feat):
    """Generation 7 + the fold: Service-style soft spread constraints together with required anti-affinity on the hostname key (a term
    may be both) stay on the score-table kernel; every placement against the oracle, with and without per-scenario node ranks."""
    rng = np.random.default_rng(11)
    on7 = 0
    for seed, (N, P) in enumerate([(40, 300), (200, 900), (700, 1500), (1300, 2500)]):
        prob = randprob.rand_problem(7300 + seed, N=N, P=P, spread_soft=True, n_node_classes=4, n_pod_classes=9, **feat)
        scen, orders = randprob.rand_scenarios(170 + seed, prob, S=5)
        ref = O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 7, (st.kernel_variant, st.kernel_generation)
            assert_same(res, ref)
            ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
            for s_, (n, _) in enumerate(np.asarray(scen).tolist()):
                ranks[s_, :n] = rng.permutation(n)
            ctx.load_scenarios(scen, orders)
            ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            assert ctx.stats().kernel_generation == 7
            assert_same(ctx.fetch(True), O.run(prob, scen, orders, node_ranks=ranks))
        on7 += 1
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_FOLD": "1"})          # without the fold: the walks over the position-mask rows (round 6) ...
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)
    res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_FOLD": "1", "SIMON_NO_RS": "1"})   # ... and without those: the all-feature kernel
    assert variant == capi.KERNEL_WIDE
    assert_same(res, ref)


@pytest.mark.parametrize("feat", [dict(ipa_self=True), dict(ipa_self=True, spread_soft=True), dict(ipa_self=True, spread_soft=True, anti_host=True),
                                  dict(ipa_self=True, spread_soft=True, static_mask=True, presets=True, gates=True, pins=True, nz_differs=True)])
def test_preferred_pod_affinity_in_self_referential_form_on_generation_7(feat):
    """InterPodAffinity preferred terms whose owners are the pods they select (the usual chart default: prefer not to sit next to your own
    replicas): the raw score is a function of (class, count) and joins the spread score in spread_select's table (simon_hip.hip:
    spread_supported).  Random problems -- not every draw has the shape (a class with two per-node counters keeps the all-feature
    kernel) -- every placement against the oracle on whichever kernel runs, and the same problems with SIMON_NO_IPA_FOLD=1."""
    on7 = 0
    for seed, (N, P) in enumerate([(40, 300), (200, 900), (700, 1500), (1300, 2500), (90, 600), (400, 1200)]):
        prob = randprob.rand_problem(7600 + seed, N=N, P=P, n_node_classes=4, n_pod_classes=5 + seed, **feat)
        scen, orders = randprob.rand_scenarios(270 + seed, prob, S=5)
        ref = O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
        on7 += st.kernel_generation == 7
        assert_same(res, ref)
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_IPA_FOLD": "1"})
        assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)
    assert on7 >= 2, on7


def test_k8s_service_and_preferred_self_anti_affinity_share_their_counter_rows():
    """A Deployment behind a Service whose pods also PREFER not to sit next to their own replicas (hostname 100, zone 50): the Service's
    default spread constraints and the anti-affinity terms select the same pods on the same keys -- on a fully zoned cluster their
    counters are the same, the terms share one row each (simon_hip.hip: sp_rep) and the preferred score joins spread_select's table.
    Object level, `simulate.sweep`, every placement against the oracle."""
    import randk8s
    from open_simulator_amd import k8s, simulate as sim
    rng = np.random.default_rng(5)
    nodes = []
    for j in range(90):
        shape = [("8", "16Gi"), ("16", "32Gi")][j % 2]
        nodes.append({"apiVersion": "v1", "kind": "Node", "metadata": {"name": f"node-{j}", "labels": {randk8s.HOST: f"node-{j}", randk8s.ZONE: f"z{j % 3}",
                                                                                                         "disk": ["ssd", "hdd"][(j // 3) % 2]}},
                      "status": {"allocatable": {"cpu": shape[0], "memory": shape[1], "pods": "40"}, "capacity": {"cpu": shape[0], "memory": shape[1]}}})
    workloads, services = [], []
    # three Deployments with the SAME labels behind one Service, one anywhere, one on the ssd, one on the hdd nodes: their default constraints are terms with
    # different node sets whose hostname-key counters are one row -- counted on every node (each class reads inside its own set only)
    for name, sel in (("twin-any", None), ("twin-ssd", {"disk": "ssd"}), ("twin-hdd", {"disk": "hdd"})):
        spec = {"containers": [{"name": "c", "image": "x", "resources": {"requests": {"cpu": "250m", "memory": "256Mi"}}}]}
        if sel:
            spec["nodeSelector"] = sel
        workloads.append({"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": name, "namespace": "default"},
                          "spec": {"replicas": 30, "selector": {"matchLabels": {"app": "twin"}}, "template": {"metadata": {"labels": {"app": "twin"}}, "spec": spec}}})
    services.append({"apiVersion": "v1", "kind": "Service", "metadata": {"name": "svc-twin", "namespace": "default"}, "spec": {"selector": {"app": "twin"}}})
    for w in range(14):
        app = f"app{w}"
        spec = {"containers": [{"name": "c", "image": "x", "resources": {"requests": {
            "cpu": str(rng.choice(["100m", "250m", "500m"])), "memory": str(rng.choice(["128Mi", "512Mi", "1Gi"]))}}}]}
        if w % 3 != 2:
            spec["affinity"] = {"podAntiAffinity": {"preferredDuringSchedulingIgnoredDuringExecution": [
                {"weight": 100, "podAffinityTerm": {"labelSelector": {"matchLabels": {"app": app}}, "topologyKey": randk8s.HOST}},
                {"weight": int(rng.choice([20, 50])), "podAffinityTerm": {"labelSelector": {"matchLabels": {"app": app}}, "topologyKey": randk8s.ZONE}}]}}
        if w % 5 == 4:          # a hard zone constraint: a per-class verdict of spread_select
            spec["topologySpreadConstraints"] = [{"maxSkew": int(1 + w % 2), "topologyKey": randk8s.ZONE, "whenUnsatisfiable": "DoNotSchedule",
                                                  "labelSelector": {"matchLabels": {"app": app}}}]
        workloads.append({"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": app, "namespace": "default"},
                          "spec": {"replicas": int(rng.integers(10, 50)), "selector": {"matchLabels": {"app": app}},
                                   "template": {"metadata": {"labels": {"app": app}}, "spec": spec}}})
        if w % 4 != 3:
            services.append({"apiVersion": "v1", "kind": "Service", "metadata": {"name": f"svc-{w}", "namespace": "default"}, "spec": {"selector": {"app": app}}})
    cluster = k8s.group_resources(nodes + services)
    apps = [sim.AppResource("shop", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {randk8s.ZONE: "z1"}},
                "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "40"}, "capacity": {"cpu": "16", "memory": "32Gi"}}}

    class Recording(sim.HipEngine):
        def run(self, prob, scen, orders, want_placement=True, **kw):
            self.args, self.kw = (prob, scen, orders), kw
            self.out = super().run(prob, scen, orders, want_placement, **kw)
            return self.out

    eng = Recording()
    sim.sweep(cluster, apps, template, [0, 4, 8, 12, 16, 20], engine=eng)
    prob, scen, orders = eng.args
    assert prob.pref_off is not None and len(scen) == 6
    assert eng.last_stats.kernel_variant == capi.KERNEL_NARROW_CACHE and eng.last_stats.kernel_generation == 7
    ranks = eng.kw.get("node_ranks")
    assert_same(eng.out, O.run(prob, scen, orders, node_ranks=ranks) if ranks is not None else O.run(prob, scen, orders))


@pytest.mark.parametrize("feat", [dict(hard_simple=True), dict(hard_simple=True, spread_soft=True), dict(hard_simple=True, spread_soft=True, ipa_self=True, anti_host=True),
                                  dict(hard_simple=True, static_mask=True, presets=True, gates=True, tight_pods=True)])   # (pinned pods with a hard constraint: all-feature kernel)
def test_hard_zone_spread_constraints_as_class_verdicts_on_generation_7(feat):
    """DoNotSchedule constraints on a zone-like key whose eligible nodes are all the labelled ones: the node classes are split by zone, so
    the filter is a per-class verdict inside spread_select (count + self - minimum over the registered zones <= maxSkew; classes without
    the label are out) and the Simon normalisation runs over the classes that are left.  Every placement against the oracle, and the same
    problems with SIMON_NO_HARD_FOLD=1 (all-feature kernel)."""
    on7 = 0
    for seed, (N, P) in enumerate([(40, 300), (200, 900), (700, 1500), (1300, 2500), (90, 600), (400, 1200)]):
        prob = randprob.rand_problem(7800 + seed, N=N, P=P, n_node_classes=4, n_pod_classes=5 + seed, **feat)
        scen, orders = randprob.rand_scenarios(370 + seed, prob, S=5)
        ref = O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders)
            st = ctx.stats()
        on7 += st.kernel_generation == 7
        assert_same(res, ref)
        res, variant = run_gpu(prob, scen, orders, env={"SIMON_NO_HARD_FOLD": "1"})
        assert variant == capi.KERNEL_WIDE
        assert_same(res, ref)
    assert on7 >= 3, on7


def test_k8s_typical_cluster_sweep_on_generation_7():
    """profiles/e2e_sweep.py --typical at a size the oracle checks in seconds: Deployments / StatefulSets behind Services, preferred and
    required anti-affinity to their own replicas, hard zone constraints, tolerations, node selectors -- the whole sweep on the score-table
    kernel, every placement against the oracle."""
    import randk8s
    from open_simulator_amd import k8s, simulate as sim
    from open_simulator_amd import synth as _synth
    nodes, workloads, services = _synth.typical_cluster_objects(3, 360, 70, 40)
    for j, n in enumerate(nodes):
        n["metadata"]["labels"][randk8s.ZONE] = f"z{j % 3}"
    cluster = k8s.group_resources(nodes + services)
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "z0"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "110"}, "capacity": {"cpu": "32", "memory": "64Gi"}}}

    class Recording(sim.HipEngine):
        def run(self, prob, scen, orders, want_placement=True, **kw):
            self.args, self.kw = (prob, scen, orders), kw
            self.out = super().run(prob, scen, orders, want_placement, **kw)
            return self.out

    eng = Recording()
    sim.sweep(cluster, apps, template, [0, 20, 40, 60, 80, 100, 120], engine=eng)
    prob, scen, orders = eng.args
    assert prob.n_pods > 1000 and len(scen) == 7
    assert eng.last_stats.kernel_variant == capi.KERNEL_NARROW_CACHE and eng.last_stats.kernel_generation == 7
    ranks = eng.kw.get("node_ranks")
    assert_same(eng.out, O.run(prob, scen, orders, node_ranks=ranks) if ranks is not None else O.run(prob, scen, orders))
