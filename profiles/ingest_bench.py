#!/usr/bin/env python3
"""Host-side ingest timing: workload -> pod expansion (`simulate.build_stream`) and objects -> SoA
(`flatten.flatten`) on synthetic Kubernetes objects of BASELINE config-5 size.  CPU only; run from the repo root:

    python profiles/ingest_bench.py            # template / node-signature interning (the product path)
    python profiles/ingest_bench.py --per-pod  # same stream, tokens stripped: every pod evaluated on its own

The second mode is the pre-interning behaviour of the pod half (the node half cannot be switched off); the numbers of
the fully pod-by-pod, node-by-node version this replaced are recorded in profiles/README.md."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401  (registers the package alias)
import randk8s  # noqa: E402
from open_simulator_amd import flatten as fl, k8s, simulate as sim, workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2500)
    ap.add_argument("--new-nodes", type=int, default=2500)
    ap.add_argument("--workloads", type=int, default=400)
    ap.add_argument("--max-replicas", type=int, default=250)
    ap.add_argument("--daemonsets", type=int, default=2)
    ap.add_argument("--per-pod", action="store_true")
    a = ap.parse_args()
    nodes, workloads, services = randk8s.rand_cluster(1, n_nodes=a.nodes, n_workloads=a.workloads, max_replicas=a.max_replicas)
    for n in nodes:
        n["metadata"]["labels"].pop(randk8s.ZONE, None)          # one zone key: prefix pools keep nodeTree order
    ds = [{"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": f"agent{i}", "namespace": "kube-system"},
           "spec": {"selector": {"matchLabels": {"app": f"agent{i}"}},
                    "template": {"metadata": {"labels": {"app": f"agent{i}"}},
                                 "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "100m", "memory": "128Mi"}}}],
                                          "tolerations": [{"operator": "Exists"}]}}}} for i in range(a.daemonsets)]
    cluster = k8s.group_resources(nodes + services + ds)
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd"}},
                "status": {"allocatable": {"cpu": "16", "memory": "32Gi", "pods": "20"}, "capacity": {"cpu": "16", "memory": "32Gi"}}}
    pool = cluster["Node"] + wl.new_fake_nodes(template, a.new_nodes)
    t0 = time.perf_counter()
    pods, gates = sim.build_stream(cluster, apps, pool, a.nodes)
    t1 = time.perf_counter()
    if a.per_pod:
        pods = [{k: v for k, v in p.items() if k != "_tmpl"} for p in pods]
    t2 = time.perf_counter()
    flat = fl.flatten(pool, pods, services, [], [], gates)
    t3 = time.perf_counter()
    print(json.dumps({"mode": "per-pod" if a.per_pod else "interned", "nodes": len(pool), "pods": len(pods),
                      "expand_s": round(t1 - t0, 3), "flatten_s": round(t3 - t2, 3), **flat.info}))


if __name__ == "__main__":
    main()
