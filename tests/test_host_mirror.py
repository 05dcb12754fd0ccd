"""CPU tests of the host-side mirror (YAML objects -> SoA -> engine -> SimulateResult).

The checker here is tests/pyref_sched.py, an object-level restatement of the reference scheduler that shares no code
with the SoA path; the engine under test on CPU is the C oracle (injected), on the GPU box the HIP library
(tests/test_gpu_parity.py::test_k8s_*)."""
import json
import os

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
import pyref_sched
import randk8s
from open_simulator_amd import capi, fiterror, flatten as fl, k8s, simulate as sim, workloads as wl

K8S_FIXTURES = sorted(f[:-5] for f in os.listdir(G.GOLDEN) if f.startswith("k8s_") and f.endswith(".json"))
REF_EXAMPLE = "/root/reference/example"


class OracleEngine:
    """Test-only engine with the HipEngine interface."""

    def run(self, prob, scen, orders, want_placement=True, node_ranks=None, want_gpu_slices=False):
        return O.run(prob, scen, orders, want_placement, node_ranks=node_ranks, want_gpu_slices=want_gpu_slices)

    def explain(self, prob, n_nodes, order, max_failed):
        _, (nf, failed, codes) = O.run(prob, [[n_nodes, 0]], np.asarray(order)[None], explain_scenario=0, max_failed=max_failed)
        return nf, failed, codes, O.LAST_LOCAL_DETAIL[0]


def load_k8s_fixture(name):
    with open(os.path.join(G.GOLDEN, name + ".json")) as f:
        d = json.load(f)
    pd = dict(d["problem"])
    u64 = ("static_mask", "node_sets")
    kw = {k: (np.asarray(v, np.uint64) if k in u64 else np.asarray(v)) for k, v in pd.items()
          if k not in ("n_pod_classes", "n_node_classes", "local_specs")}
    if "local_specs" in pd:                                  # structured array, stored field by field
        fields = pd["local_specs"]
        specs = np.zeros(len(fields["n_lvm"]), capi.LOCAL_SPEC_DTYPE)
        for name, v in fields.items():
            specs[name] = np.asarray(v)
        kw["local_specs"] = specs
    prob = capi.Problem(n_pod_classes=pd["n_pod_classes"], n_node_classes=pd["n_node_classes"], **kw).normalise()
    return d, prob


@pytest.mark.parametrize("name", K8S_FIXTURES)
def test_oracle_reproduces_object_level_placements(name):
    d, prob = load_k8s_fixture(name)
    P, n = d["n_pods"], d["n_nodes"]
    res = O.run(prob, [[n, 0]], np.arange(P, dtype=np.int32)[None])
    assert res.placement[0].tolist() == d["object_level_placement"]


def _random_case(seed, **kw):
    nodes, workloads, services = randk8s.rand_cluster(seed, **kw)
    rs = [w for w in workloads if w["kind"] == "ReplicaSet"] if seed % 2 else []
    nodes = [nodes[j] for j in k8s.canonical_node_order(nodes)]
    cluster = {k: [] for k in k8s.KINDS}
    cluster["Node"], cluster["Service"] = nodes, services
    if seed % 3 != 1:       # DaemonSets: per-node pods pinned by matchFields (one with its own node selector terms and a toleration)
        spec_a = {"containers": [{"name": "a", "image": "busybox", "resources": {"requests": {"cpu": "100m", "memory": "64Mi"}}}]}
        spec_b = {"containers": [{"name": "b", "image": "busybox", "resources": {"requests": {"cpu": "200m"}}}],
                  "tolerations": [{"key": "dedicated", "operator": "Exists"}],
                  "affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [
                      {"matchExpressions": [{"key": "disk", "operator": "In", "values": ["ssd"]}]},
                      {"matchExpressions": [{"key": "rank", "operator": "Exists"}]}]}}}}
        if seed % 4 == 3:   # a DaemonSet with a DoNotSchedule constraint: its pods look only at their own node
            spec_a = dict(spec_a, topologySpreadConstraints=[{"maxSkew": 1, "topologyKey": randk8s.ZONE, "whenUnsatisfiable": "DoNotSchedule",
                                                              "labelSelector": {"matchLabels": {"app": "ds-0"}}}])
        cluster["DaemonSet"] = [
            {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": f"ds-{i}", "namespace": "kube-system"},
             "spec": {"selector": {"matchLabels": {"app": f"ds-{i}"}}, "template": {"metadata": {"labels": {"app": f"ds-{i}"}}, "spec": sp}}}
            for i, sp in enumerate([spec_a, spec_b][:1 + seed % 2])]
    pods, _ = sim.build_stream(cluster, [sim.AppResource("app", k8s.group_resources(workloads))], nodes, len(nodes))
    return nodes, pods, services, rs


@pytest.mark.parametrize("seed", range(24))
def test_flatten_plus_oracle_equals_object_level_scheduler(seed):
    nodes, pods, services, rs = _random_case(seed, gpu=(seed % 3 == 0), local=(seed % 2 == 0))
    flat = fl.flatten(nodes, pods, services, rs, [], storage_classes=randk8s.STORAGE_CLASSES)
    res = O.run(flat.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
    ref = pyref_sched.Scheduler(nodes, services, rs, [], randk8s.STORAGE_CLASSES).run(pods)
    got = [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()]
    assert got == ref


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLE), reason="reference tree not present (GPU box)")
def test_committed_example_fixtures_match_the_reference_inputs():
    """The committed k8s_example_* fixtures are what flatten() produces from the reference's example/ YAML today."""
    cluster = k8s.group_resources(k8s.load_objects(os.path.join(REF_EXAMPLE, "cluster/demo_1")))
    apps = [sim.AppResource("simple", k8s.group_resources(k8s.load_objects(os.path.join(REF_EXAMPLE, "application/simple"))))]
    nodes = cluster["Node"]
    pods, _ = sim.build_stream(cluster, apps, nodes, len(nodes))
    flat = fl.flatten(nodes, pods, cluster["Service"], cluster["ReplicaSet"], cluster["StatefulSet"])
    d, prob = load_k8s_fixture("k8s_example_simple")
    assert d["pod_names"] == ["/".join(r) for r in flat.pod_refs]
    for f in ("alloc_cpu", "alloc_mem", "req_cpu", "req_mem", "nz_cpu", "nz_mem", "pod_class", "simon_raw", "static_mask"):
        a, b = getattr(flat.problem, f), getattr(prob, f)
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), f
    # core_test.go-style expectation on the workload expansion: per-workload pod counts of example/application/simple
    kinds = {}
    for p in pods:
        k = p["metadata"]["annotations"].get(wl.ANNO_WORKLOAD_KIND, "Pod")
        kinds[k] = kinds.get(k, 0) + 1
    assert sum(kinds.values()) == len(pods) == 37


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLE), reason="reference tree not present (GPU box)")
def test_reference_gpushare_config_end_to_end():
    """`simon apply -f example/simon-gpushare-config.yaml` through the mirror: the cluster alone holds every pod (the
    expectation the reference documents for this config), so the add-nodes search answers 0 new nodes."""
    cfg = sim.load_config(os.path.join(REF_EXAMPLE, "simon-gpushare-config.yaml"), base_dir=os.path.dirname(REF_EXAMPLE))
    assert len(cfg["cluster"]["Node"]) == 2 and [a.name for a in cfg["apps"]] == ["pai_gpu"] and cfg["new_node"] is not None
    sw = sim.sweep(cfg["cluster"], cfg["apps"], cfg["new_node"], range(0, 3), engine=OracleEngine())
    assert sw.unscheduled == [0, 0, 0] and sw.best == 0
    placed = sum(len(s["pods"]) for s in sw.result.node_status)
    assert placed == 9 and not sw.result.unscheduled_pods
    # what the mirror hangs on pods for its own use never leaves it; result nodes are objects of their own (top level, metadata, status)
    assert not any(k.startswith("_") for s in sw.result.node_status for p in s["pods"] for k in p)
    by_name = {n["metadata"]["name"]: n for n in cfg["cluster"]["Node"]}
    for s in sw.result.node_status:
        src = by_name.get(s["node"]["metadata"]["name"])
        if src is not None:
            assert s["node"] is not src and s["node"]["metadata"] is not src["metadata"] and s["node"]["status"] is not src["status"]
    # gpu-pod-00 and gpu-pod-02 carry the gpu-mem annotation (gpu-pod-01 has none; the ReplicaSet's sit on the RS object)
    gpu_pods = [p for s in sw.result.node_status for p in s["pods"] if "alibabacloud.com/gpu-mem" in (p["metadata"].get("annotations") or {})]
    assert len(gpu_pods) == 2


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLE), reason="reference tree not present (GPU box)")
def test_reference_open_local_example_end_to_end():
    """example/application/open_local on example/cluster/demo_1 with example/newnode/demo_1: each nginx-lvm pod needs
    10Gi + 40Gi of LVM and one 100Gi HDD device; worker-1 and the new-node template own ONE hdd device each, the
    masters are tainted or carry no local storage -> one pod per storage node, 3 new nodes for the 4 replicas."""
    root = os.path.dirname(REF_EXAMPLE)
    cluster = k8s.group_resources(k8s.load_objects(os.path.join(REF_EXAMPLE, "cluster/demo_1")))
    sim.attach_local_storage(cluster["Node"], os.path.join(REF_EXAMPLE, "cluster/demo_1"))
    yoda = [{"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "yoda-lvm-default"}, "parameters": {"volumeType": "LVM"}},
            {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "yoda-device-hdd"},
             "parameters": {"volumeType": "Device", "mediaType": "hdd"}}]
    app = sim.AppResource("open_local", k8s.group_resources(k8s.load_objects(os.path.join(REF_EXAMPLE, "application/open_local")) + yoda))
    tmpl = k8s.group_resources(k8s.load_objects(os.path.join(REF_EXAMPLE, "newnode/demo_1")))["Node"]
    sim.attach_local_storage(tmpl, os.path.join(REF_EXAMPLE, "newnode/demo_1"))
    sw = sim.sweep(cluster, [app], tmpl[0], range(0, 5), engine=OracleEngine())
    assert sw.unscheduled == [3, 2, 1, 0, 0] and sw.best == 3
    hosts = {s["node"]["metadata"]["name"]: [p["metadata"]["name"] for p in s["pods"] if p["metadata"]["name"].startswith("nginx-lvm")]
             for s in sw.result.node_status}
    assert hosts["worker-1"] == ["nginx-lvm-0"] and [hosts[f"simon-{i:05d}"] for i in range(3)] == [["nginx-lvm-1"], ["nginx-lvm-2"], ["nginx-lvm-3"]]
    # MaxVG (pkg/apply/apply.go:747-771): 4 x 50Gi requested over master-1 + worker-1 (200Gi each) + k x 500Gi
    assert sw.vg_pct == [int(50 * (1 + min(k, 3)) / (400 + 500 * k) * 100) for k in range(5)]
    capped = sim.sweep(cluster, [app], tmpl[0], range(0, 7), engine=OracleEngine(), max_vg=9)
    assert capped.vg_pct[3] == 10 and capped.vg_pct[4] == 8 and capped.best == 4
    one = sim.simulate(cluster, [app], engine=OracleEngine())
    assert [u["pod"]["metadata"]["name"] for u in one.unscheduled_pods] == ["nginx-lvm-1", "nginx-lvm-2", "nginx-lvm-3"]
    # master-1 is tainted, master-2/3 have no storage annotation (Unschedulable without a reason), worker-1's one device is taken:
    # ProcessDevicePVC's hdd branch reports the SSD counts (algo/common.go:428-434) -- 0 requested, 0 free -- and the node's device total
    assert one.unscheduled_pods[0]["reason"].endswith(
        "0/4 nodes are available: 1 Insufficient Device storage, requested 0, available 0, capacity 1, 1 node(s) had taint {node-role.kubernetes.io/master: }, "
        "that the pod didn't tolerate.")
    del root


def _with_images(nodes, seed):
    """some nodes list the image the random pods run ("busybox" -> "busybox:latest") and an unrelated one"""
    rng = np.random.default_rng(seed)
    for n in nodes:
        if rng.random() < 0.5:
            n.setdefault("status", {})["images"] = [
                {"names": ["registry.local/other:1.0", "other@sha256:abc"], "sizeBytes": 500 << 20},
                {"names": ["busybox:latest", "docker.io/library/busybox@sha256:123"], "sizeBytes": int(rng.integers(30, 2500)) << 20}]
    return nodes


@pytest.mark.parametrize("seed", [2, 5, 9, 14])
def test_image_locality_single_size(seed):
    """ImageLocality (imagelocality/image_locality.go:53-113) for ONE cluster size: a static per (class, node) score next to
    NodePreferAvoidPods in static_add -- against the object-level scheduler, which keeps its own imageStates."""
    nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=9, n_workloads=10)
    nodes = _with_images([nodes[j] for j in k8s.canonical_node_order(nodes)], seed)
    cluster = {k: [] for k in k8s.KINDS}
    cluster["Node"], cluster["Service"] = nodes, services
    pods, _ = sim.build_stream(cluster, [sim.AppResource("app", k8s.group_resources(workloads))], nodes, len(nodes))
    flat = fl.flatten(nodes, pods, services, [], [], image_total=len(nodes))
    assert flat.problem.static_add is not None and (flat.problem.static_add % 10000 != 0).any()      # the plugin scores somewhere
    res = O.run(flat.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
    ref = pyref_sched.Scheduler(nodes, services, [], []).run(pods)
    assert [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()] == ref
    without = fl.flatten([dict(n, status={k: v for k, v in n["status"].items() if k != "images"}) for n in nodes], pods, services, [], [])
    res0 = O.run(without.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
    assert res0.placement.tolist() != res.placement.tolist()                                            # ... and it matters


def test_image_locality_in_a_sweep_runs_size_by_size():
    """ImageLocality scores 0 everywhere unless a node lists an image some pod runs; then the score depends on the cluster
    size: one batch over several sizes cannot carry it, sweep() evaluates every size as its own problem."""
    nodes, workloads, services = randk8s.rand_cluster(4, n_nodes=6, n_workloads=5)
    for n in nodes:
        n["metadata"]["labels"].pop(randk8s.ZONE, None)
    cluster = k8s.group_resources(nodes + services)
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {}},
                "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "20"}, "capacity": {"cpu": "16", "memory": "32Gi"}}}
    nodes[2].setdefault("status", {})["images"] = [{"names": ["registry.local/other:1.0", "other@sha256:abc"], "sizeBytes": 500 << 20}]
    sim.sweep(cluster, apps, template, [0, 1, 2], engine=OracleEngine())      # unrelated image: still a constant
    nodes[2]["status"]["images"].append({"names": ["busybox:latest"], "sizeBytes": 300 << 20})      # the pods run "busybox"
    with pytest.raises(fl.Unsupported, match="ImageLocality"):              # one batch over several sizes: not expressible
        pool = cluster["Node"] + wl.new_fake_nodes(template, 2)
        fl.flatten(pool, sim.build_stream(cluster, apps, pool, len(nodes))[0], services, [], [])
    sw = sim.sweep(cluster, apps, template, [0, 1, 2], engine=OracleEngine())   # ... so sweep() runs size by size
    for k, uns in zip([0, 1, 2], sw.unscheduled):
        assert len(sim.simulate(cluster, apps, engine=OracleEngine(), new_nodes=wl.new_fake_nodes(template, k)).unscheduled_pods) == uns


def _prio_cluster():
    node = lambda name: {"apiVersion": "v1", "kind": "Node", "metadata": {"name": name, "labels": {"kubernetes.io/hostname": name}},
                         "status": {"allocatable": {"cpu": "2", "memory": "4Gi", "pods": "10"}, "capacity": {"cpu": "2", "memory": "4Gi"}}}
    pod = lambda name, cpu, prio=None: {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default"},
                                        "spec": dict({"containers": [{"name": "c", "image": "busybox", "resources": {"requests": {"cpu": cpu, "memory": "1Gi"}}}]},
                                                     **({"priority": prio} if prio is not None else {}))}
    return node, pod


def test_pods_of_unequal_priority_run_on_the_engine_and_only_risky_scenarios_are_flagged():
    """DefaultPreemption (V/scheduler.go:479, defaultpreemption/default_preemption.go:578-592) acts only for a pod that FAILED and only on pods
    of lower priority.  The mirror no longer refuses clusters with differing spec.priority (every kubeconfig-dumped cluster has them):
    the engine flags the scenarios in which such a pair exists, everything else is exact -- checked against the object-level scheduler,
    which looks for victims the way selectVictimsOnNode does."""
    node, pod = _prio_cluster()
    cluster = k8s.group_resources([node("n0")])
    template = node("tmpl")
    # low, low, HIGH: on one node (2 cpu) the third pod fails while lower-priority pods sit there -> flagged; with a second node all fit
    apps = [sim.AppResource("a", k8s.group_resources([pod("low0", "900m", 0), pod("low1", "900m", 0), pod("vip", "900m", 1000)]))]
    with pytest.raises(sim.NeedsReference, match="DefaultPreemption"):
        sim.simulate(cluster, apps, engine=OracleEngine())
    assert isinstance(sim.NeedsReference("x"), fl.Unsupported)            # hosts that catch Unsupported fall back to the Go path as before
    res = sim.simulate(cluster, apps, engine=OracleEngine(), new_nodes=wl.new_fake_nodes(template, 1))
    assert not res.unscheduled_pods
    sw = sim.sweep(cluster, apps, template, [0, 1, 2], engine=OracleEngine())
    assert sw.unscheduled == [1, 0, 0] and sw.needs_reference == [True, False, False] and sw.best == 1
    # HIGH first: the pod that fails has the LOWEST priority -> nothing it could evict -> exact, not flagged
    apps2 = [sim.AppResource("a", k8s.group_resources([pod("vip", "900m", 1000), pod("mid", "900m", 10), pod("low", "900m", 0)]))]
    res2 = sim.simulate(cluster, apps2, engine=OracleEngine())
    assert [u["pod"]["metadata"]["name"] for u in res2.unscheduled_pods] == ["low"]
    assert sim.sweep(cluster, apps2, template, [0, 1], engine=OracleEngine()).needs_reference == [False, False]
    # equal priorities among the failing and the placed pods: no victim either (strictly lower is required)
    apps3 = [sim.AppResource("a", k8s.group_resources([pod("a", "900m", 5), pod("b", "900m", 5), pod("c", "900m", 5), pod("d", "100m", 7)]))]
    assert sim.sweep(cluster, apps3, template, [0], engine=OracleEngine()).needs_reference == [False]
    # the object-level restatement agrees on random clusters whose pods carry random priorities
    for seed in (3, 6, 14, 29):
        nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=4, n_workloads=8, max_replicas=6)
        for i, w in enumerate(workloads):
            spec = w["spec"]["template"]["spec"] if "template" in w.get("spec", {}) else w["spec"]
            spec["priority"] = [0, 0, 100, 1000][(seed + i) % 4]
        cl = k8s.group_resources(nodes + services)
        ap = [sim.AppResource("app", k8s.group_resources(workloads))]
        pods, _ = sim.build_stream(cl, ap, cl["Node"], len(cl["Node"]))
        order = k8s.canonical_node_order(cl["Node"])
        sch = pyref_sched.Scheduler([cl["Node"][j] for j in order], cl.get("Service", []), cl.get("ReplicaSet", []), cl.get("StatefulSet", []))
        want = sch.run(pods)
        flagged = False
        try:
            got = sim.simulate(cl, ap, engine=OracleEngine())
            assert len(got.unscheduled_pods) == sum(1 for w_ in want if w_ is None)
        except sim.NeedsReference:
            flagged = True
        assert flagged == sch.preempt_risk, seed


def test_a_zero_quantity_scalar_entry_disables_the_all_zero_shortcut():
    """fit.go:244-249 returns early only for len(podRequest.ScalarResources) == 0, and Resource.Add (V/framework/types.go:320-322) creates the map
    entry even for a quantity of 0: a BestEffort pod that lists `example.com/foo: "0"` is checked against cpu / memory like any other pod --
    "Insufficient cpu" on a node that bound pods over-committed -- while the same pod without the entry passes.  """
    node, pod = _prio_cluster()
    hog = pod("hog", "3", None)
    hog["spec"]["nodeName"] = "n0"                                     # bound by Spec.NodeName: no filter, the node ends up over-committed (3 of 2 cpu)
    be = lambda name, res: {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default"},
                            "spec": {"containers": [{"name": "c", "image": "busybox", "resources": {"requests": res}}]}}
    cluster = k8s.group_resources([node("n0"), hog])
    for res, fails in (({}, False), ({"example.com/foo": "0"}, True), ({"hugepages-2Mi": "0"}, True), ({"storage": "0"}, False)):
        apps = [sim.AppResource("a", k8s.group_resources([be("besteffort", res)]))]
        out = sim.simulate(cluster, apps, engine=OracleEngine())
        assert bool(out.unscheduled_pods) == fails, res
        if fails:
            assert "Insufficient cpu" in out.unscheduled_pods[0]["reason"] and "Insufficient example" not in out.unscheduled_pods[0]["reason"]
        pods, _ = sim.build_stream(cluster, apps, cluster["Node"], 1)
        assert (pyref_sched.Scheduler(cluster["Node"]).run(pods)[-1] is None) == fails, res
    # the zero entry itself is compared too: a node over-committed on THAT resource fails it with the resource's own reason
    n1 = node("n0")
    n1["status"]["allocatable"]["example.com/foo"] = "1"
    hog2 = be("hog2", {"example.com/foo": "2"})
    hog2["spec"]["nodeName"] = "n0"
    cl2 = k8s.group_resources([n1, hog2])
    out = sim.simulate(cl2, [sim.AppResource("a", k8s.group_resources([be("z", {"example.com/foo": "0"}), be("plain", {})]))], engine=OracleEngine())
    assert [u["pod"]["metadata"]["name"] for u in out.unscheduled_pods] == ["z"] and "Insufficient example.com/foo" in out.unscheduled_pods[0]["reason"]


def test_sweep_over_several_zones_runs_every_size_on_its_own(seeds=(11, 31, 47)):
    """With nodes in several zones the nodeTree order of one cluster size is not a prefix of the next (the clones join
    one zone's round-robin list).  sweep() still runs ONE batch -- the pool stays "cluster nodes, then clones" and every
    scenario brings its canonical ranks (simon_set_node_ranks) -- and must answer like one Simulate() per size."""
    eng = OracleEngine()
    for seed in seeds:
        nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=7, n_workloads=12, max_replicas=8)
        assert len({k8s.zone_key(n) for n in nodes}) > 1
        cluster = k8s.group_resources(nodes + services)
        apps = [sim.AppResource("app", k8s.group_resources(workloads))]
        template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "z1"}},
                    "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "20"}, "capacity": {"cpu": "16", "memory": "32Gi"}}}
        counts = [0, 1, 2, 4]
        with pytest.raises(fl.Unsupported):
            sim._check_prefix_order(cluster["Node"] + wl.new_fake_nodes(template, 4), [len(nodes) + k for k in counts])
        sw = sim.sweep(cluster, apps, template, counts, engine=eng)
        for k, uns in zip(counts, sw.unscheduled):
            one = sim.simulate(cluster, apps, engine=eng, new_nodes=wl.new_fake_nodes(template, k))
            assert len(one.unscheduled_pods) == uns, (seed, k)
            if sw.best == k:            # the plan's placements, node by node
                by_node = lambda res: {st["node"]["metadata"]["name"]: sorted(p["metadata"]["name"] for p in st["pods"]) for st in res.node_status}
                assert by_node(sw.result) == by_node(one), seed


def _problem_arrays(prob):
    out = {}
    for f in capi.Problem.__dataclass_fields__:
        v = getattr(prob, f)
        if f != "_keep" and v is not None:
            out[f] = np.asarray(v)
    return out


@pytest.mark.parametrize("seed", [3, 8, 21])
def test_template_interning_equals_per_pod_evaluation(seed):
    """Ingest: flatten() evaluates each workload template once (`_tmpl` tokens from workloads.py) and
    each pod class once per node signature.  The result must be the arrays a pod-by-pod evaluation gives: same stream
    with the tokens stripped (every pod its own template), with DaemonSet pods (matchFields pins) and Open-Local in."""
    nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=14, n_workloads=12, gpu=(seed % 3 == 0), local=True)
    ds = {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "agent", "namespace": "kube-system"},
          "spec": {"selector": {"matchLabels": {"app": "agent"}},
                   "template": {"metadata": {"labels": {"app": "agent"}},
                                "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "100m"}}}],
                                         "nodeSelector": {"disk": "ssd"}}}}}
    nodes = [nodes[j] for j in k8s.canonical_node_order(nodes)]
    cluster = k8s.group_resources(nodes + services + [ds])
    pods, gates = sim.build_stream(cluster, [sim.AppResource("app", k8s.group_resources(workloads))], nodes, len(nodes))
    assert any("_tmpl" in p for p in pods) and any("_daemon_node" in p for p in pods)
    os.environ["SIMON_CHECK_TEMPLATES"] = "1"
    try:
        a = fl.flatten(nodes, pods, services, [], [], gates, storage_classes=randk8s.STORAGE_CLASSES)
    finally:
        del os.environ["SIMON_CHECK_TEMPLATES"]
    stripped = [{k: v for k, v in p.items() if k != "_tmpl"} for p in pods]
    b = fl.flatten(nodes, stripped, services, [], [], gates, storage_classes=randk8s.STORAGE_CLASSES)
    xa, xb = _problem_arrays(a.problem), _problem_arrays(b.problem)
    assert xa.keys() == xb.keys()
    for k in xa:
        assert xa[k].dtype == xb[k].dtype and xa[k].shape == xb[k].shape and (xa[k] == xb[k]).all(), k
    # a replica that was edited after expansion breaks the token promise: the debug check names it
    seen = set()
    later = [p for p in pods if "_tmpl" in p and (p["_tmpl"] in seen or seen.add(p["_tmpl"]))]      # non-first replicas
    later[-1]["metadata"]["labels"]["edited"] = "yes"
    os.environ["SIMON_CHECK_TEMPLATES"] = "1"
    try:
        with pytest.raises(AssertionError, match="differs from its template"):
            fl.flatten(nodes, pods, services, [], [], gates, storage_classes=randk8s.STORAGE_CLASSES)
    finally:
        del os.environ["SIMON_CHECK_TEMPLATES"]


def test_sweep_equals_one_simulate_per_cluster_size():
    """The batched add-nodes search (gated DaemonSet pods, prefix node pools) against one Simulate() per size."""
    nodes, workloads, services = randk8s.rand_cluster(5, n_nodes=6, n_workloads=14, max_replicas=8)
    for n in nodes:
        n["metadata"]["labels"].pop(randk8s.ZONE, None)          # single zone key: nodeTree order == insertion order
    ds = {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "agent", "namespace": "kube-system"},
          "spec": {"selector": {"matchLabels": {"app": "agent"}},
                   "template": {"metadata": {"labels": {"app": "agent"}},
                                "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "200m", "memory": "128Mi"}}}],
                                         "tolerations": [{"operator": "Exists"}]}}}}
    cluster = k8s.group_resources(nodes + services + [ds])
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd"}},
                "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "20"}, "capacity": {"cpu": "16", "memory": "32Gi"}}}
    eng = OracleEngine()
    counts = [0, 1, 2, 3, 5]
    sw = sim.sweep(cluster, apps, template, counts, engine=eng)
    for k, uns in zip(counts, sw.unscheduled):
        one = sim.simulate(cluster, apps, engine=eng, new_nodes=wl.new_fake_nodes(template, k))
        assert len(one.unscheduled_pods) == uns, k
        if k == sw.best:                                         # same placements, pod for pod, node for node
            names = lambda r: [(s["node"]["metadata"]["name"], [p["metadata"]["name"] for p in s["pods"]]) for s in r.node_status]
            assert names(one) == names(sw.result)
    if sw.best is not None:
        assert sw.unscheduled[counts.index(sw.best)] == 0 and sw.result is not None
        n_ds = sum(1 for s in sw.result.node_status for p in s["pods"] if p["metadata"]["name"].startswith("agent-"))
        assert n_ds == len(nodes) + sw.best                      # one DaemonSet pod per node of the chosen size


def test_sweep_finds_the_minimum_node_count():
    """Applier.Run's loop on a case that needs new nodes: 40 x (1 cpu, 1Gi) + a DaemonSet on 3 x 8-cpu nodes, 16-cpu template."""
    mk = lambda name: {"apiVersion": "v1", "kind": "Node", "metadata": {"name": name, "labels": {k8s.LABEL_HOSTNAME: name}},
                       "status": {"allocatable": {"cpu": "8", "memory": "16Gi", "pods": "110"}}}
    deploy = {"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "web", "namespace": "default"},
              "spec": {"replicas": 40, "selector": {"matchLabels": {"app": "web"}},
                       "template": {"metadata": {"labels": {"app": "web"}},
                                    "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": "1", "memory": "1Gi"}}}]}}}}
    ds = {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "agent", "namespace": "kube-system"},
          "spec": {"selector": {"matchLabels": {"app": "agent"}},
                   "template": {"metadata": {"labels": {"app": "agent"}},
                                "spec": {"containers": [{"name": "a", "resources": {"requests": {"cpu": "500m", "memory": "128Mi"}}}]}}}}
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"role": "worker"}},
                "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "110"}}}
    cluster = k8s.group_resources([mk("a"), mk("b"), mk("c"), ds])
    apps = [sim.AppResource("web", k8s.group_resources([deploy]))]
    eng = OracleEngine()
    sw = sim.sweep(cluster, apps, template, range(0, 5), engine=eng)
    # capacity: 3 x floor(8 - 0.5) = 21 pods on the cluster, 15 per new node -> 2 new nodes hold the remaining 19
    assert sw.unscheduled == [19, 4, 0, 0, 0] and sw.best == 2
    assert sw.cpu_pct[2] == int((40 * 1000 + 5 * 500) / (3 * 8000 + 2 * 16000) * 100)
    one = sim.simulate(cluster, apps, engine=eng, new_nodes=wl.new_fake_nodes(template, 2))
    names = lambda r: [(s["node"]["metadata"]["name"], [p["metadata"]["name"] for p in s["pods"]]) for s in r.node_status]
    assert names(one) == names(sw.result) and not one.unscheduled_pods
    assert sum(1 for s in sw.result.node_status for p in s["pods"] if p["metadata"]["name"].startswith("agent-")) == 5
    assert all(p["spec"]["nodeName"] == s["node"]["metadata"]["name"] and p["status"]["phase"] == "Running"
               for s in sw.result.node_status for p in s["pods"])
    # occupancy caps (satisfyResourceSetting): a 60 % CPU cap needs a larger cluster than pure feasibility
    capped = sim.sweep(cluster, apps, template, range(0, 5), engine=eng, max_cpu=60)
    assert capped.best == 3 and capped.cpu_pct[3] <= 60 < capped.cpu_pct[2]


def test_unscheduled_reason_text():
    node = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "n0", "labels": {}},
            "status": {"allocatable": {"cpu": "1", "memory": "1Gi", "pods": "10"}}}
    tainted = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "n1", "labels": {}},
               "spec": {"taints": [{"key": "dedicated", "value": "infra", "effect": "NoSchedule"}]},
               "status": {"allocatable": {"cpu": "8", "memory": "8Gi", "pods": "10"}}}
    pod = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "big", "namespace": "default"},
           "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": "2", "memory": "2Gi"}}}]}}
    cluster = k8s.group_resources([node, tainted, pod])
    res = sim.simulate(cluster, [], engine=OracleEngine())
    assert len(res.unscheduled_pods) == 1
    assert res.unscheduled_pods[0]["reason"] == (
        "failed to schedule pod (default/big): Unschedulable: 0/2 nodes are available: 1 Insufficient cpu, 1 Insufficient memory, "
        "1 node(s) had taint {dedicated: infra}, that the pod didn't tolerate.")


def test_node_ports_conflicts_and_reason():
    """HostPortInfo.CheckConflict (V/framework/types.go:784-812): same (protocol, port) conflicts on the same IP, and a
    0.0.0.0 binding conflicts with every IP; UDP does not conflict with TCP."""
    mk = lambda name: {"apiVersion": "v1", "kind": "Node", "metadata": {"name": name, "labels": {}},
                       "status": {"allocatable": {"cpu": "8", "memory": "16Gi", "pods": "10"}}}

    def pod(name, ports):
        return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default"},
                "spec": {"containers": [{"name": "c", "ports": ports, "resources": {"requests": {"cpu": "100m", "memory": "64Mi"}}}]}}
    pods = [pod("a", [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.1"}]),          # n0 (tie: first maximum)
            pod("b", [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.2"}]),          # n1 (LeastAllocated: emptier node)
            pod("c", [{"containerPort": 80, "hostPort": 8080}]),                                # 0.0.0.0 conflicts with ANY IP: nowhere
            pod("d", [{"containerPort": 80, "hostPort": 8080, "protocol": "UDP"}]),             # UDP never conflicts with TCP: n0
            pod("e", [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.3"}]),          # no wildcard placed: n1 (emptier)
            pod("f", [{"containerPort": 80, "hostPort": 8080, "hostIP": "10.0.0.1"}])]          # n0 holds the same IP: n1
    cluster = k8s.group_resources([mk("n0"), mk("n1")] + pods)
    res = sim.simulate(cluster, [], engine=OracleEngine())
    where = {p["metadata"]["name"]: s["node"]["metadata"]["name"] for s in res.node_status for p in s["pods"]}
    ref = pyref_sched.Scheduler([mk("n0"), mk("n1")]).run([wl.make_valid_pod(p) for p in pods])
    assert [where.get(n) for n in "abcdef"] == ref == ["n0", "n1", None, "n0", "n1", "n1"]
    assert res.unscheduled_pods[0]["reason"] == ("failed to schedule pod (default/c): Unschedulable: 0/2 nodes are available: "
                                                 "2 node(s) didn't have free ports for the requested pod ports.")


def test_k8s_helpers():
    labels = {"app": "web", "tier": "fe"}
    assert k8s.label_selector_matches({}, labels) and not k8s.label_selector_matches(None, labels)
    assert k8s.label_selector_matches({"matchExpressions": [{"key": "tier", "operator": "NotIn", "values": ["be"]}]}, labels)
    assert not k8s.label_selector_matches({"matchExpressions": [{"key": "zone", "operator": "Exists"}]}, labels)
    node = {"metadata": {"name": "n1", "labels": {"rank": "7", "disk": "ssd"}}}
    assert k8s.node_selector_term_matches({"matchExpressions": [{"key": "rank", "operator": "Gt", "values": ["4"]}]}, node)
    assert not k8s.node_selector_term_matches({}, node)                                       # empty term matches nothing
    assert k8s.node_selector_term_matches({"matchFields": [{"key": "metadata.name", "operator": "In", "values": ["n1"]}]}, node)
    taint = {"key": "k", "value": "v", "effect": "NoSchedule"}
    assert k8s.toleration_tolerates({"operator": "Exists"}, taint)
    assert not k8s.toleration_tolerates({"key": "k", "value": "x"}, taint)
    assert not k8s.toleration_tolerates({"key": "k", "operator": "Exists", "effect": "NoExecute"}, taint)
    pod = {"metadata": {}, "spec": {"containers": [{"resources": {"requests": {"cpu": "100m"}}}, {"resources": {"requests": {"cpu": "1.5", "memory": "1Gi"}}}],
                                    "initContainers": [{"resources": {"requests": {"cpu": "3"}}}], "overhead": {"cpu": "10m"}}}
    assert k8s.pod_request(pod) == {"cpu": 3010, "memory": 1 << 30}
    assert k8s.pod_nonzero_request(pod) == (3010, (200 << 20) + (1 << 30))                    # first container: default memory
    # nodeTree.list(): zones round-robin in first-appearance order
    mk = lambda name, zone: {"metadata": {"name": name, "labels": ({k8s.LABEL_ZONE: zone} if zone else {})}}
    nodes = [mk("a", "z1"), mk("b", "z1"), mk("c", "z2"), mk("d", None), mk("e", "z1")]
    assert [nodes[j]["metadata"]["name"] for j in k8s.canonical_node_order(nodes)] == ["a", "c", "d", "b", "e"]


def test_fit_error_strings_of_the_new_codes():
    msg = fiterror.fit_error([capi.FAIL_AFFINITY, capi.FAIL_SPREAD, capi.FAIL_SPREAD_LABEL, capi.FAIL_SPREAD])
    assert msg == ("0/4 nodes are available: 1 node(s) didn't match pod affinity rules, 1 node(s) didn't match pod affinity/anti-affinity, "
                   "1 node(s) didn't match pod topology spread constraints (missing required label), "
                   "2 node(s) didn't match pod topology spread constraints.")


def test_daemonset_with_hard_spread_constraint_looks_only_at_its_own_node():
    """4 nodes in zones za x 3 and zb x 1, a DaemonSet with a zone DoNotSchedule maxSkew=1 constraint.
    A DaemonSet pod's affinity names ONE node, so only that node's domain registers (filtering.go:236-251) and every pod
    fits; the shared class view registered every zone and left two pods out."""
    nodes = [{"apiVersion": "v1", "kind": "Node", "metadata": {"name": f"n{i}", "labels": {"kubernetes.io/hostname": f"n{i}", randk8s.ZONE: z}},
              "status": {"allocatable": {"cpu": "4", "memory": "8Gi", "pods": "110"}}} for i, z in enumerate(["za", "za", "za", "zb"])]
    ds = {"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "ds", "namespace": "default"},
          "spec": {"selector": {"matchLabels": {"app": "ds"}}, "template": {"metadata": {"labels": {"app": "ds"}}, "spec": {
              "containers": [{"name": "c", "image": "busybox", "resources": {"requests": {"cpu": "100m", "memory": "64Mi"}}}],
              "topologySpreadConstraints": [{"maxSkew": 1, "topologyKey": randk8s.ZONE, "whenUnsatisfiable": "DoNotSchedule",
                                             "labelSelector": {"matchLabels": {"app": "ds"}}}]}}}}
    cluster = {k: [] for k in k8s.KINDS}
    cluster["Node"], cluster["DaemonSet"] = nodes, [ds]
    pods, _ = sim.build_stream(cluster, [], nodes, len(nodes))
    assert len(pods) == 4 and all("_class_affinity" not in p and p["_daemon_node"] for p in pods)
    order = k8s.canonical_node_order(nodes)
    nodes_c = [nodes[j] for j in order]
    flat = fl.flatten(nodes_c, pods, [], [], [])
    res = O.run(flat.problem, [[4, 0]], np.arange(4, dtype=np.int32)[None])
    ref = pyref_sched.Scheduler(nodes_c, [], [], [], randk8s.STORAGE_CLASSES).run(pods)
    got = [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()]
    assert got == ref and None not in got and res.unscheduled.tolist() == [0]
    # the same DaemonSet with a ScheduleAnyway constraint keeps the shared class + pin_node
    ds["spec"]["template"]["spec"]["topologySpreadConstraints"][0]["whenUnsatisfiable"] = "ScheduleAnyway"
    pods, _ = sim.build_stream(cluster, [], nodes, len(nodes))
    assert all("_class_affinity" in p for p in pods)
    flat = fl.flatten(nodes_c, pods, [], [], [])
    assert flat.problem.pin_node is not None and flat.problem.n_pod_classes == 1
    res = O.run(flat.problem, [[4, 0]], np.arange(4, dtype=np.int32)[None])
    assert [flat.node_names[j] for j in res.placement[0].tolist()] == pyref_sched.Scheduler(nodes_c, [], [], [], randk8s.STORAGE_CLASSES).run(pods)


def test_named_zero_extended_resource_with_no_other_request_travels_as_an_entry():
    """fit.go:244-249 tests len(ScalarResources) == 0; until round 6 the engine tested the values and the mirror refused the one input where
    they differ.  Now the entries travel (ABI v6): bit 7 for a resource nobody requests a quantity of."""
    nodes = [{"apiVersion": "v1", "kind": "Node", "metadata": {"name": "n0", "labels": {"kubernetes.io/hostname": "n0"}},
              "status": {"allocatable": {"cpu": "4", "memory": "8Gi", "pods": "110", "example.com/foo": "4"}}}]
    pod = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p", "namespace": "default"},
           "spec": {"containers": [{"name": "c", "image": "busybox", "resources": {"requests": {"example.com/foo": "0"}}}]}}
    flat = fl.flatten(nodes, [pod], [], [], [])
    assert flat.problem.scalar_entries.tolist() == [0x80] and flat.problem.scalar_req is None


def test_service_list_edited_in_place_between_two_flatten_calls_is_read_afresh():
    """Round-4 advisor finding: the (namespace, first selector pair) index of the Services was validated by list identity and length
    only, so a selector edited IN PLACE between two flatten() calls kept serving the old index -- default spread selectors from stale
    selectors, no error.  The index is now built once per flatten() call."""
    from open_simulator_amd import flatten as fl, workloads as wl
    nodes = [{"metadata": {"name": f"n{j}", "labels": {"kubernetes.io/hostname": f"n{j}"}},
              "status": {"allocatable": {"cpu": "8", "memory": "16Gi", "pods": "110"}}} for j in range(3)]
    pod = wl.make_valid_pod({"kind": "Pod", "metadata": {"name": "p", "namespace": "default", "labels": {"app": "a"}},
                             "spec": {"containers": [{"name": "c", "image": "busybox", "resources": {"requests": {"cpu": "1", "memory": "1Gi"}}}]}})
    services = [{"kind": "Service", "metadata": {"name": "s", "namespace": "default"}, "spec": {"selector": {"app": "b"}}}]
    f1 = fl.flatten(nodes, [pod], services, [], [])
    assert f1.problem.spread_soft_off is None or int(f1.problem.spread_soft_off[-1]) == 0        # nobody selects the pod
    services[0]["spec"]["selector"] = {"app": "a"}                                               # same list object, same length
    f2 = fl.flatten(nodes, [pod], services, [], [])
    assert f2.problem.spread_soft_off is not None and int(f2.problem.spread_soft_off[-1]) == 2   # the system defaults: hostname + zone
    services[0]["spec"]["selector"] = {"app": "b"}
    f3 = fl.flatten(nodes, [pod], services, [], [])
    assert f3.problem.spread_soft_off is None or int(f3.problem.spread_soft_off[-1]) == 0


def test_term_index_finds_every_match(monkeypatch):
    """flatten tries only the terms a class can match (indexed by a matchLabels pair of their selector); SIMON_CHECK_MATCH_INDEX makes it
    compare with the classes x terms evaluation: random clusters with every selector shape randk8s draws."""
    import randk8s
    from open_simulator_amd import flatten as fl, simulate as sim
    monkeypatch.setenv("SIMON_CHECK_MATCH_INDEX", "1")
    for seed in range(6):
        nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=14, n_workloads=30, max_replicas=4, gpu=seed % 2 == 0, local=seed % 3 == 0)
        nodes = [nodes[j] for j in k8s.canonical_node_order(nodes)]
        cluster = {k: [] for k in k8s.KINDS}
        cluster["Node"], cluster["Service"] = nodes, services
        pods, _ = sim.build_stream(cluster, [sim.AppResource("app", k8s.group_resources(workloads))], nodes, len(nodes))
        try:
            fl.flatten(nodes, pods, services, [], [], storage_classes=randk8s.STORAGE_CLASSES)
        except fl.Unsupported:
            pass


@pytest.mark.parametrize("seed", range(4))
def test_typical_cluster_against_the_object_level_scheduler(seed):
    """profiles/e2e_sweep.py --typical (Deployments / StatefulSets behind Services, preferred and required anti-affinity to their own
    replicas, hard zone constraints, tolerations, node selectors -- the shapes the score-table kernel keeps since round 3) at a small
    size: flatten + oracle against tests/pyref_sched.py, the restatement that works on the objects the way the Go code does."""
    from open_simulator_amd import synth as _synth
    nodes, workloads, services = _synth.typical_cluster_objects(100 + seed, 45, 16, 12)
    for j, n in enumerate(nodes):
        n["metadata"]["labels"][randk8s.ZONE] = f"z{j % 3}"
        n["status"]["allocatable"]["pods"] = "20"
    nodes = [nodes[j] for j in k8s.canonical_node_order(nodes)]
    cluster = {k: [] for k in k8s.KINDS}
    cluster["Node"], cluster["Service"] = nodes, services
    pods, _ = sim.build_stream(cluster, [sim.AppResource("app", k8s.group_resources(workloads))], nodes, len(nodes))
    flat = fl.flatten(nodes, pods, services, [], [])
    assert flat.problem.pref_off is not None and flat.problem.spread_soft_off is not None
    res = O.run(flat.problem, [[len(nodes), 0]], np.arange(len(pods), dtype=np.int32)[None])
    ref = pyref_sched.Scheduler(nodes, services, [], []).run(pods)
    assert [None if j < 0 else flat.node_names[j] for j in res.placement[0].tolist()] == ref
