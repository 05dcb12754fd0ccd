"""pkg/simulator/core_test.go:32-362 ("simple") restated as SoA input (tests only).

4 nodes (8 cpu / 16Gi / 110 pods: pkg/test/node.go:15-37): master-1 (taint node-role.kubernetes.io/master:NoSchedule),
master-2, master-3, worker-1.  Pods in the order Simulate() feeds them: the cluster's static pods (preset NodeName),
its Deployment and DaemonSet pods (pkg/simulator/core.go:85-95), then app "simple" (kind order fixed as SURVEY H2
prescribes: Pods, Deployments, ReplicaSets, StatefulSets, Jobs, DaemonSets).  The static filters (taints,
nodeSelector, required node affinity, DaemonSet matchFields pinning) are evaluated by hand into static_mask.
The reference asserts only counts: 0 unscheduled pods (core_test.go:346) and per-workload pod counts (:364-591).
Not modelled (next-tier N1): the StatefulSet's PREFERRED pod anti-affinity (a score, cannot change feasibility).
"""
import numpy as np

from open_simulator_amd import capi, synth
from open_simulator_amd.quantity import parse_quantity as pq, simon_raw_score

MiB, GiB = 1 << 20, 1 << 30
M1, M2, M3, W1 = 1, 2, 4, 8          # node bit of master-1..3, worker-1


def problem():
    pods = []    # (name, cpu "", mem "", mask, preset)

    def add(name, cpu, mem, mask, preset=-1, count=1):
        for i in range(count):
            pods.append((f"{name}-{i}" if count > 1 else name, cpu, mem, mask, preset))

    # cluster static pods, Spec.NodeName = master-1 (core_test.go:140-153)
    add("etcd-master-1", "", "", M1, 0)
    add("kube-apiserver-master-1", "250m", "", M1, 0)
    add("kube-controller-manager-master-1", "200m", "", M1, 0)
    add("kube-scheduler-master-1", "100m", "", M1, 0)
    # metrics-server: node affinity master Exists, no toleration -> master-2/3; its required anti-affinity uses a zone
    # key no node carries, so it can never match (interpodaffinity/filtering.go:133-148)
    add("metrics-server", "1", "500Mi", M2 | M3)
    # DaemonSets: one pod per eligible node, pinned by matchFields metadata.name (pkg/utils/utils.go:337-366)
    for bit in (M1, M2, M3):
        add("kube-proxy-master", "", "", bit)
    add("kube-proxy-worker", "", "", W1)
    for bit in (M1, M2, M3):
        add("coredns", "100m", "70Mi", bit)
    # app "simple"
    add("single-pod", "100m", "100Mi", M1 | M2 | M3)                       # nodeSelector master + toleration
    add("busybox-deploy", "1500m", "1Gi", M1 | M2 | M3 | W1, count=4)      # tolerates the master taint
    add("calico-kube-controllers", "", "", M1 | M2 | M3 | W1, count=2)     # tolerates everything, zero requests
    add("busybox-sts", "1", "512Mi", M1 | M2 | M3 | W1, count=4)
    add("pi", "100m", "100Mi", M2 | M3 | W1)                               # no toleration
    add("busybox-ds", "500m", "512Mi", W1)                                 # node affinity master DoesNotExist

    alloc = {"cpu": pq("8"), "memory": pq("16Gi"), "pods": pq("110")}
    classes, pod_class = {}, []
    req_cpu, req_mem, nz_cpu, nz_mem, preset = [], [], [], [], []
    for name, cpu, mem, mask, pre in pods:
        key = (cpu, mem, mask)
        if key not in classes:
            classes[key] = len(classes)
        pod_class.append(classes[key])
        c = pq(cpu).milli_value() if cpu else 0
        m = pq(mem).int_value() if mem else 0
        req_cpu.append(c); req_mem.append(m)
        nz_cpu.append(c if cpu else 100)                 # V/util/non_zero.go:35-38 defaults
        nz_mem.append(m if mem else 200 * MiB)
        preset.append(pre)
    Cp = len(classes)
    mask = np.zeros((Cp, 1), np.uint64)
    raw = np.zeros((Cp, 1), np.int64)
    for (cpu, mem, m), c in classes.items():
        mask[c, 0] = m
        req = {}
        if cpu: req["cpu"] = pq(cpu)
        if mem: req["memory"] = pq(mem)
        raw[c, 0] = simon_raw_score(req, alloc)
    prob = capi.Problem(alloc_cpu=[8000] * 4, alloc_mem=[16 * GiB] * 4, alloc_pods=[110] * 4, node_class=[0] * 4,
                        req_cpu=req_cpu, req_mem=req_mem, nz_cpu=nz_cpu, nz_mem=nz_mem, pod_class=pod_class,
                        preset_node=preset, n_pod_classes=Cp, n_node_classes=1, static_mask=mask,
                        static_reason=np.where((mask >> np.arange(4, dtype=np.uint64)[None, :]) & np.uint64(1), 0, 3).astype(np.uint8),
                        simon_raw=raw, const_score=np.full(Cp, synth.CONST_SCORE))
    names = [p[0] for p in pods]
    return prob.normalise(), names


EXPECTED_COUNTS = {"busybox-deploy": 4, "busybox-sts": 4, "calico-kube-controllers": 2, "coredns": 3,
                   "kube-proxy-master": 3, "kube-proxy-worker": 1, "busybox-ds": 1, "pi": 1, "single-pod": 1,
                   "metrics-server": 1}
