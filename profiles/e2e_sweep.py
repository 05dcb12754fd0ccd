#!/usr/bin/env python3
"""End-to-end `simon apply`-style sweep at config-5 scale through the Python mirror on the HIP engine:
random Kubernetes objects (tests/randk8s.py: affinity, spread constraints incl. system defaults, taints, host ports, ...)
-> expand -> flatten -> ONE scenario batch on the GPU -> SimulateResult of the best plan.  Prints one JSON line with the
wall-clock of every stage.  Needs a GPU:  python profiles/e2e_sweep.py [--nodes 2500 --new-nodes 2500 --workloads 400]
Parity of this path is covered by tests/test_gpu_parity.py::test_k8s_sweep_at_scale_matches_oracle (the oracle is test
infrastructure and is not touched from here)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
import randk8s  # noqa: E402
from open_simulator_amd import capi, flatten as fl, k8s, simulate as sim, synth, workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2500)
    ap.add_argument("--new-nodes", type=int, default=2500)
    ap.add_argument("--workloads", type=int, default=400)
    ap.add_argument("--max-replicas", type=int, default=250)
    ap.add_argument("--counts", type=int, default=64)
    ap.add_argument("--daemonsets", type=int, default=1)
    ap.add_argument("--own-zone", action="store_true", help="new nodes form a zone of their own: nodeTree order differs per cluster size "
                    "(per-scenario node ranks, simon_set_node_ranks)")
    ap.add_argument("--uniform-pods", action="store_true", help="every node allows 110 pods (fewer node classes: the <= 64 class path)")
    ap.add_argument("--typical", action="store_true", help="a cluster of the usual workloads instead of the adversarial random mix: Deployments behind "
                    "Services, preferred (some required) anti-affinity to their own replicas, some hard zone constraints, tolerations, node selectors")
    a = ap.parse_args()
    if a.typical:
        nodes, workloads, services = synth.typical_cluster_objects(1, a.nodes, a.workloads, a.max_replicas)
    else:
        nodes, workloads, services = randk8s.rand_cluster(1, n_nodes=a.nodes, n_workloads=a.workloads, max_replicas=a.max_replicas)
    # prefix pools need ONE zone round-robin order: keep zone labels only on a prefix-stable pattern (all nodes zoned, by index)
    for j, n in enumerate(nodes):
        n["metadata"]["labels"][randk8s.ZONE] = f"z{j % 3}"
        if a.uniform_pods or a.typical:
            n["status"]["allocatable"]["pods"] = "110"
    ds = [{"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": f"agent{i}", "namespace": "kube-system"},
           "spec": {"selector": {"matchLabels": {"app": f"agent{i}"}},
                    "template": {"metadata": {"labels": {"app": f"agent{i}"}},
                                 "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "100m", "memory": "128Mi"}}}],
                                          "tolerations": [{"operator": "Exists"}]}}}} for i in range(a.daemonsets)]
    cluster = k8s.group_resources(nodes + services + ds)
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "znew" if a.own_zone else "z0"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "40"}, "capacity": {"cpu": "32", "memory": "64Gi"}}}
    counts = np.unique(np.linspace(0, a.new_nodes, a.counts).astype(int)).tolist()

    class TimedEngine(sim.HipEngine):
        def run(self, prob, scen, orders, want_placement=True, node_ranks=None):
            self.problem, self.scen, self.orders, self.ranked = prob, scen, orders, node_ranks is not None
            t0 = time.perf_counter()
            with capi.Context(self.device_id) as ctx:
                ctx.load_problem(prob)
                ctx.load_scenarios(scen, orders)
                if node_ranks is not None:
                    ctx.set_node_ranks(node_ranks)
                ctx.run_loaded(want_placement)
                out = ctx.fetch(want_placement)
                self.stats = ctx.stats()
            self.engine_s = time.perf_counter() - t0
            self.out = out
            return out

    eng = TimedEngine()
    t0 = time.perf_counter()
    try:
        sw = sim.sweep(cluster, apps, template, counts, engine=eng)
    except fl.Unsupported as e:
        print(json.dumps({"unsupported": str(e)}))
        return
    total = time.perf_counter() - t0
    prob = eng.problem
    res = {"nodes": a.nodes, "new_nodes_max": a.new_nodes, "pods": int(prob.n_pods), "scenarios": len(counts),
           "pod_classes": int(prob.n_pod_classes), "node_classes": int(prob.n_node_classes),
           "total_s": round(total, 2), "engine_s": round(eng.engine_s, 2), "kernel_ms": round(eng.stats.kernel_ms, 1),
           "kernel_variant": int(eng.stats.kernel_variant), "workgroup": int(eng.stats.workgroup_size),
           "kernel_generation": int(eng.stats.kernel_generation), "host_s": round(total - eng.engine_s, 2), "node_ranks": bool(eng.ranked), "best_new_nodes": sw.best, "unscheduled_first_last": [sw.unscheduled[0], sw.unscheduled[-1]]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
