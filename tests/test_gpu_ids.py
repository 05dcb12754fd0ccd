"""GPU device ids (ABI v4): what GpuSharePlugin.Reserve books -- the alibabacloud.com/gpu-index annotation of a placed GPU pod
(pkg/type/open-gpu-share/utils/pod.go:117-127), simon/node-gpu-share of its node (pkg/simulator/plugin/open-gpu-share.go:160-186) --
and pods that ARRIVE with a gpu-index annotation (GpuNodeInfo.AllocateGpuId, gpunodeinfo.go:247-253).
CPU part: oracle (hand vectors), the object-level restatement, the host mirror.  GPU part (-m gpu): HIP vs oracle, both engines."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, flatten as fl, k8s, simulate as sim, synth
from test_host_mirror import OracleEngine
from test_oracle import gpushare_problem

GiB = 1 << 30
MiB = 1 << 20
REF_EXAMPLE = "/root/reference/example"


def ids(x):
    return capi.gpu_ids_of(int(x))


def test_pack_and_unpack():
    assert capi.pack_gpu_index([2]) == 0x3 and capi.pack_gpu_index([0, 0, 1]) == 0x211
    assert ids(0x0000000000000102) == [0, 0, 1] and ids(0) == []
    with pytest.raises(ValueError):
        capi.pack_gpu_index([8])
    with pytest.raises(ValueError):
        capi.pack_gpu_index([0] * 9)


def alloc_rules_problem(gpu_index=None):
    G = GiB
    return capi.Problem(alloc_cpu=[64000], alloc_mem=[256 * G], alloc_pods=[110], gpu_cnt=[4], gpu_mem_total=[64 * G],
                        init_gpu_used=np.array([[10 * G, 4 * G, 12 * G, 0, 0, 0, 0, 0]]),
                        req_cpu=[100] * 5, req_mem=[G] * 5, gpu_mem=[4 * G, 4 * G, 6 * G, 16 * G, 16 * G],
                        pod_gpu_cnt=[1, 1, 2, 1, 1], gpu_index=gpu_index, n_pod_classes=1, n_node_classes=1)


def test_oracle_emits_the_ids_of_the_hand_vector():
    """tests/test_oracle.py::test_gpu_allocate_rules with the devices spelled out: idle [6,12,4,16] -> 4G x1 takes the tightest
    (dev 2), the next 4G dev 0, 6G x2 stacks twice on dev 1 ("1-1"), 16G dev 3, the last 16G finds nothing."""
    res = O.run(alloc_rules_problem(), [[1, 0]], np.arange(5)[None], want_gpu_slices=True)
    assert res.placement[0].tolist() == [0, 0, 0, 0, capi.UNSCHEDULED]
    assert [ids(x) for x in res.gpu_slices[0]] == [[2], [0], [1, 1], [3], []]


def test_oracle_honours_an_arriving_gpu_index():
    """gpunodeinfo.go:247-253: a valid gpu-index on the incoming pod is returned as is -- whatever is idle.  Pod 0 names device 2 twice
    ("2-2": over-commits it, 12 + 8 > 16), pod 1 names a device the node lacks (id 6 of 4: skipped at addOrUpdatePod, the pod is still
    placed), pod 3 (16G, no annotation) then finds dev 3, pod 4 nothing."""
    gi = [capi.pack_gpu_index([2, 2]), capi.pack_gpu_index([6]), 0, 0, 0]
    res = O.run(alloc_rules_problem(gi), [[1, 0]], np.arange(5)[None], want_gpu_slices=True)
    assert res.placement[0].tolist() == [0, 0, 0, 0, capi.UNSCHEDULED]
    # pod 2 (6G x2, no annotation): idle is now [6, 12, -4, 16] -> dev 0 once (6 -> 0), dev 1 once
    assert [ids(x) for x in res.gpu_slices[0]] == [[2, 2], [], [0, 1], [3], []]
    # a node whose TOTAL gpu-mem is below the request fails the filter before any id is looked at (open-gpu-share.go:64-67)
    small = capi.Problem(alloc_cpu=[64000], alloc_mem=[256 * GiB], alloc_pods=[110], gpu_cnt=[2], gpu_mem_total=[8 * GiB],
                         req_cpu=[100], req_mem=[GiB], gpu_mem=[16 * GiB], pod_gpu_cnt=[1], gpu_index=[capi.pack_gpu_index([0])],
                         n_pod_classes=1, n_node_classes=1)
    assert O.run(small, [[1, 0]], np.arange(1)[None]).placement[0].tolist() == [capi.UNSCHEDULED]


def test_gpushare_example_ids():
    """example/application/gpushare on example/cluster/gpushare: gpu-pod-00 (1024Mi x1) one device, gpu-pod-02 (10240Mi x2) two
    different devices of its node (a 16 GiB device holds one 10 GiB slice)."""
    res = O.run(gpushare_problem(), [[2, 0]], np.arange(9)[None], want_gpu_slices=True)
    got = [ids(x) for x in res.gpu_slices[0]]
    assert len(got[0]) == 1 and got[2] == [got[2][0], got[2][0] + 1] and all(g == [] for i, g in enumerate(got) if i not in (0, 2))


def _gpu_cluster():
    def node(name, cnt):
        return {"apiVersion": "v1", "kind": "Node", "metadata": {"name": name, "labels": {"kubernetes.io/hostname": name, k8s.GPU_MODEL: "V100"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "110", k8s.GPU_COUNT: str(cnt), k8s.GPU_MEM: f"{cnt * 16}Gi"},
                           "capacity": {"cpu": "32", "memory": "64Gi", k8s.GPU_COUNT: str(cnt), k8s.GPU_MEM: f"{cnt * 16}Gi"}}}

    def pod(name, mem, cnt, index=None):
        ann = {k8s.GPU_MEM: mem, k8s.GPU_COUNT: str(cnt)}
        if index:
            ann[k8s.GPU_INDEX] = index
        return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "ml", "annotations": ann},
                "spec": {"containers": [{"name": "c", "image": "x", "resources": {"requests": {"cpu": "1", "memory": "1Gi"}}}]}}
    nodes = [node("g0", 2), node("g1", 4)]
    pods = [pod("a", "8Gi", 1), pod("b", "6Gi", 2), pod("c", "4Gi", 1, "1-1"), pod("d", "16Gi", 1), pod("e", "2Gi", 3)]
    return nodes, pods


def test_object_level_scheduler_and_oracle_agree_on_ids():
    """tests/pyref_sched.py (objects, no SoA) and flatten -> oracle book the same devices, incl. a pod arriving with gpu-index."""
    import pyref_sched
    nodes, pods = _gpu_cluster()
    flat = fl.flatten(nodes, pods, [], [], [])
    assert flat.problem.gpu_index is not None and flat.problem.gpu_index.tolist() == [0, 0, capi.pack_gpu_index([1, 1]), 0, 0]
    res = O.run(flat.problem, [[2, 0]], np.arange(len(pods))[None], want_gpu_slices=True)
    sched = pyref_sched.Scheduler(nodes, [], [], [], [])
    where = sched.run([dict(p) for p in pods])
    idx = {n: j for j, n in enumerate(flat.node_names)}
    assert [(-1 if w is None else idx[w]) for w in where] == res.placement[0].tolist()
    for i, w in enumerate(where):
        if w is None:
            continue
        ni = next(x for x in sched.infos if x.node["metadata"]["name"] == w)
        assert [d for d in sched.gpu_ids[i] if d < ni.gpu_cnt] == ids(res.gpu_slices[0][i]), pods[i]["metadata"]["name"]


def test_simulate_result_carries_the_annotations():
    """NodeStatus.Pods of GPU pods carry alibabacloud.com/gpu-index / assume-time, their nodes simon/node-gpu-share; the
    node's allocatable gpu-count is NOT touched (open-gpu-share.go:177-182 sets a copy of the Quantity and drops it)."""
    nodes, pods = _gpu_cluster()
    cluster = k8s.group_resources(nodes + pods)
    res = sim.simulate(cluster, [], engine=OracleEngine())
    assert not res.unscheduled_pods
    by = {p["metadata"]["name"]: (s["node"]["metadata"]["name"], p) for s in res.node_status for p in s["pods"]}
    for name, (node, p) in by.items():
        a = p["metadata"]["annotations"]
        assert k8s.GPU_INDEX in a and int(a[k8s.GPU_ASSUME_TIME]) > 0, name
    assert by["c"][1]["metadata"]["annotations"][k8s.GPU_INDEX] == "1-1"              # the pod's own string survives
    assert len(by["b"][1]["metadata"]["annotations"][k8s.GPU_INDEX].split("-")) == 2
    assert len(by["e"][1]["metadata"]["annotations"][k8s.GPU_INDEX].split("-")) == 3
    for s in res.node_status:
        n = s["node"]
        info = json.loads(n["metadata"]["annotations"][k8s.ANNO_NODE_GPU_SHARE])
        cnt = int(n["status"]["capacity"][k8s.GPU_COUNT])
        assert info["GpuCount"] == cnt and info["GpuModel"] == "V100" and info["GpuTotalMemory"] == f"{cnt * 16}Gi"
        assert sorted(info["DevsBrief"]) == [str(d) for d in range(cnt)]
        used = 0
        for d, brief in info["DevsBrief"].items():
            assert brief["GpuTotalMemory"] == "16Gi"
            used += fl.parse_quantity(brief["GpuUsedMemory"]).int_value()
        booked = sum(fl.parse_quantity(p["metadata"]["annotations"][k8s.GPU_MEM]).int_value() * len(p["metadata"]["annotations"][k8s.GPU_INDEX].split("-"))
                     for p in s["pods"])
        assert used == booked
        full = sum(1 for b in info["DevsBrief"].values() if fl.parse_quantity(b["GpuUsedMemory"]).int_value() >= 16 * GiB)
        assert info["GpuAllocatable"] == cnt - full
        assert n["status"]["allocatable"][k8s.GPU_COUNT] == n["status"]["capacity"][k8s.GPU_COUNT]    # as the input node had it
        assert info["NumPods"] == sum(len(b["PodList"] or []) for b in info["DevsBrief"].values())


def test_gpu_mem_without_gpu_count_is_unschedulable():
    """gpunodeinfo.go:238-240: reqGpuNum <= 0 finds no device -- the mirror used to default the count to 1 (rounds 1-2)."""
    nodes, pods = _gpu_cluster()
    del pods[0]["metadata"]["annotations"][k8s.GPU_COUNT]
    res = sim.simulate(k8s.group_resources(nodes + pods[:1]), [], engine=OracleEngine())
    assert len(res.unscheduled_pods) == 1 and "2 Node:g0" not in res.unscheduled_pods[0]["reason"]
    assert "1 Node:g0, 1 Node:g1" in res.unscheduled_pods[0]["reason"]


# ---- GPU ------------------------------------------------------------------------------------------------------------------
def _hip(prob, scen, orders, env=None, monkeypatch=None):
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(True, want_gpu_slices=True)
        return ctx.fetch(True, want_gpu_slices=True), ctx.stats(), [ctx.fetch_gpu_slices(s) for s in range(len(scen))]


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["score_table", "all_feature"])
def test_hip_books_the_devices_the_oracle_books(engine, monkeypatch):
    env = {"SIMON_NO_REST": "1"} if engine == "all_feature" else {}
    for prob, scen, orders in [(gpushare_problem(), [[2, 0]], np.arange(9)[None]), (alloc_rules_problem(), [[1, 0]], np.arange(5)[None])]:
        ref = O.run(prob, scen, orders, want_gpu_slices=True)
        res, st, rows = _hip(prob, scen, orders, env, monkeypatch)
        assert (res.placement == ref.placement).all() and (res.gpu_slices == ref.gpu_slices).all()
        assert all((rows[s] == ref.gpu_slices[s]).all() for s in range(len(scen)))
    for seed in range(6):                                    # random problems: multi-device requests, anti-affinity next to them
        prob = randprob.rand_problem(900 + seed, N=60 + 30 * seed, P=700, gpu=True, anti_host=seed % 2 == 0, static_mask=seed % 3 == 0)
        scen, orders = randprob.rand_scenarios(seed, prob, S=5)
        ref = O.run(prob, scen, orders, want_gpu_slices=True)
        res, st, rows = _hip(prob, scen, orders, env, monkeypatch)
        assert st.kernel_variant == (capi.KERNEL_WIDE if engine == "all_feature" else capi.KERNEL_NARROW_CACHE)
        assert (res.placement == ref.placement).all()
        assert (res.gpu_slices == ref.gpu_slices).all(), seed
        assert ref.gpu_slices.any()


@pytest.mark.gpu
def test_hip_honours_an_arriving_gpu_index():
    """Pods with a gpu-index annotation take the all-feature kernel (their filter ignores idle memory): the hand vector and random
    problems in which a third of the GPU pods arrive with ids."""
    gi = [capi.pack_gpu_index([2, 2]), capi.pack_gpu_index([6]), 0, 0, 0]
    prob = alloc_rules_problem(gi)
    ref = O.run(prob, [[1, 0]], np.arange(5)[None], want_gpu_slices=True)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch([[1, 0]], np.arange(5)[None], True, True)
        assert ctx.stats().kernel_variant == capi.KERNEL_WIDE
    assert (res.placement == ref.placement).all() and (res.gpu_slices == ref.gpu_slices).all()
    rng = np.random.default_rng(3)
    for seed in range(4):
        prob = randprob.rand_problem(950 + seed, N=80, P=600, gpu=True)
        g = np.zeros(prob.n_pods, np.uint32)
        for p in np.flatnonzero(prob.gpu_mem > 0):
            if rng.random() < 0.33:
                g[p] = capi.pack_gpu_index(rng.integers(0, 8, int(rng.integers(1, 4))).tolist())
        prob.gpu_index = g
        scen, orders = randprob.rand_scenarios(seed, prob, S=4)
        ref = O.run(prob, scen, orders, want_gpu_slices=True)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders, True, True)
        assert (res.placement == ref.placement).all() and (res.gpu_slices == ref.gpu_slices).all(), seed


@pytest.mark.gpu
def test_config5_subset_ids_and_group_fetch():
    """BASELINE config 5 at full size, 6 of the 256 scenarios: every GPU pod's devices against the oracle (generation 6); the same rows
    through a two-member device group; a run that did not record them says so."""
    prob, scen, orders = synth.config5()
    pick = np.unique(np.linspace(0, len(scen) - 1, 6).astype(int))
    sub = scen[pick]
    ref = O.run_threaded(prob, sub, orders, want_gpu_slices=True)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(sub, orders)
        ctx.run_loaded(True, want_gpu_slices=True)
        assert ctx.stats().kernel_generation == 6
        for s in range(len(sub)):
            assert (ctx.fetch_placement(s) == ref.placement[s]).all()
            row = ctx.fetch_gpu_slices(s)
            bad = np.flatnonzero(row != ref.gpu_slices[s])
            assert len(bad) == 0, (s, len(bad), int(bad[0]))
        assert (ref.gpu_slices != 0).sum() > 1000
        ctx.run_loaded(True)                                   # without the flag: the rows are not there, and the call says so
        with pytest.raises(capi.SimonError, match="did not record GPU devices"):
            ctx.fetch_gpu_slices(0)
    with capi.Group([0, 0]) as g:
        g.load_problem(prob)
        g.load_scenarios(sub, orders)
        g.run_loaded(True, want_gpu_slices=True)
        for s in (0, 3, 5):
            assert (g.fetch_gpu_slices(s) == ref.gpu_slices[s]).all()


@pytest.mark.parametrize("text,ids", [("1", [1]), ("0-0-1", [0, 0, 1]), ("+2", [2]), ("00-1", [0, 1]), (" 1", None), ("1_0", None),
                                      ("١", None), ("1-", None), ("", None), ("99999999999999999999", None)])
def test_gpu_index_annotation_parses_like_strconv_atoi(text, ids):
    """GpuIdStrToIntList (pkg/type/open-gpu-share/utils/pod.go:100-115) splits on '-' and runs strconv.Atoi on each part: one optional
    sign + ASCII digits, int64 range.  Forms Python's int() would take (' 1', '1_0', Unicode digits) are INVALID in the reference --
    it logs a warning and allocates normally (gpunodeinfo.go:247-253)."""
    assert fl.gpu_index_annotation({"metadata": {"annotations": {k8s.GPU_INDEX: text}}}) == ids
