"""Seeded random Kubernetes object sets (nodes, workloads, services) that exercise every placement-dependent plugin
through the YAML-level path: labels / selectors, required and preferred pod (anti-)affinity, explicit and default
topology spread constraints, node affinity, taints and tolerations, GPU share annotations.  Tests only."""
import numpy as np

ZONE = "topology.kubernetes.io/zone"
HOST = "kubernetes.io/hostname"
APPS = ["web", "db", "cache", "batch"]
TIERS = ["fe", "be"]


GiB = 1 << 30
# StorageClasses of the Open-Local cases: LVM without / with a VG name, SSD and HDD devices (names the plugin recognises,
# pkg/utils/const.go:5-16)
STORAGE_CLASSES = [
    {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "open-local-lvm"}, "parameters": {"volumeType": "LVM"}},
    {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "yoda-lvm-default"},
     "parameters": {"volumeType": "LVM", "vgName": "pool1"}},
    {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "open-local-device-ssd"},
     "parameters": {"volumeType": "Device", "mediaType": "ssd"}},
    {"apiVersion": "storage.k8s.io/v1", "kind": "StorageClass", "metadata": {"name": "open-local-device-hdd"},
     "parameters": {"volumeType": "Device", "mediaType": "hdd"}},
]


def rand_cluster(seed, n_nodes=12, n_workloads=10, gpu=False, max_replicas=6, local=False):
    rng = np.random.default_rng(seed)
    nodes = []
    for j in range(n_nodes):
        shape = [("8", "16Gi"), ("16", "32Gi"), ("32", "64Gi")][int(rng.integers(0, 3))]
        labels = {HOST: f"node-{j}", "disk": ["ssd", "hdd"][int(rng.integers(0, 2))]}
        if rng.random() < 0.85:
            labels[ZONE] = f"z{int(rng.integers(0, 3))}"
        if rng.random() < 0.3:
            labels["rank"] = str(int(rng.integers(0, 10)))
        node = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": f"node-{j}", "labels": labels},
                "status": {"allocatable": {"cpu": shape[0], "memory": shape[1], "pods": str(int(rng.integers(8, 30)))},
                           "capacity": {"cpu": shape[0], "memory": shape[1]}}}
        taints = []
        if rng.random() < 0.15:
            taints.append({"key": "dedicated", "value": "infra", "effect": "NoSchedule"})
        if rng.random() < 0.25:
            taints.append({"key": "slow", "value": "true", "effect": "PreferNoSchedule"})
        if taints:
            node["spec"] = {"taints": taints}
        if gpu and rng.random() < 0.5:
            cnt = int(rng.choice([2, 4]))
            node["status"]["capacity"]["alibabacloud.com/gpu-count"] = str(cnt)
            node["status"]["capacity"]["alibabacloud.com/gpu-mem"] = f"{cnt * 16}Gi"
            node["status"]["allocatable"]["alibabacloud.com/gpu-count"] = str(cnt)
        if local and rng.random() < 0.7:       # Open-Local storage annotation (simon/node-local-storage)
            import json
            vgs = [{"name": f"pool{v}", "capacity": str(int(rng.choice([50, 100, 100, 200])) * GiB),
                    "requested": str(int(rng.choice([0, 0, 10])) * GiB)} for v in range(int(rng.integers(0, 4)))]
            devs = [{"name": f"/dev/vd{chr(98 + d)}", "device": f"/dev/vd{chr(98 + d)}", "capacity": str(int(rng.choice([50, 100, 100, 200])) * GiB),
                     "mediaType": str(rng.choice(["ssd", "hdd", "hdd"])), "isAllocated": str(rng.random() < 0.15).lower()}
                    for d in range(int(rng.integers(0, 5)))]
            node["metadata"]["annotations"] = {"simon/node-local-storage": json.dumps({"vgs": vgs, "devices": devs})}
        nodes.append(node)

    def selector(app=None, tier=None, expr=False):
        sel = {}
        if app is not None:
            sel["matchLabels"] = {"app": app}
        if tier is not None and expr:
            sel["matchExpressions"] = [{"key": "tier", "operator": "In", "values": [tier]}]
        elif tier is not None:
            sel.setdefault("matchLabels", {})["tier"] = tier
        return sel

    def pod_term(key):
        return {"labelSelector": selector(str(rng.choice(APPS)), str(rng.choice(TIERS)) if rng.random() < 0.3 else None, rng.random() < 0.5),
                "topologyKey": key}

    workloads, services = [], []
    for w in range(n_workloads):
        app, tier = str(rng.choice(APPS)), str(rng.choice(TIERS))
        ns = "default" if rng.random() < 0.8 else "team-b"
        cpu = str(rng.choice(["100m", "250m", "500m", "1", "2"]))
        mem = str(rng.choice(["128Mi", "256Mi", "1Gi", "2Gi"]))
        container = {"name": "c", "image": "busybox"}
        if rng.random() < 0.9:
            container["resources"] = {"requests": {"cpu": cpu, "memory": mem} if rng.random() < 0.8 else {"cpu": cpu}}
        if rng.random() < 0.2:       # host ports: NodePorts filter (wildcard and specific host IPs, TCP default and UDP)
            hp = {"containerPort": 80, "hostPort": int(rng.choice([8080, 9090]))}
            if rng.random() < 0.4:
                hp["hostIP"] = str(rng.choice(["10.0.0.1", "10.0.0.2"]))
            if rng.random() < 0.3:
                hp["protocol"] = "UDP"
            container["ports"] = [hp, {"containerPort": 81}]
        spec = {"containers": [container]}
        affinity = {}
        r = rng.random()
        if r < 0.25:
            affinity["podAntiAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": [
                {"labelSelector": selector(app, tier), "topologyKey": str(rng.choice([HOST, ZONE]))}]}
        elif r < 0.4:
            affinity["podAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": [pod_term(ZONE)]}
        elif r < 0.5:
            affinity["podAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": [
                {"labelSelector": selector(app), "topologyKey": ZONE}]}       # self-affine series: first-pod escape
        if rng.random() < 0.35:
            affinity.setdefault("podAffinity", {})["preferredDuringSchedulingIgnoredDuringExecution"] = [
                {"weight": int(rng.integers(1, 101)), "podAffinityTerm": pod_term(str(rng.choice([HOST, ZONE])))}]
        if rng.random() < 0.35:
            affinity.setdefault("podAntiAffinity", {})["preferredDuringSchedulingIgnoredDuringExecution"] = [
                {"weight": int(rng.integers(1, 101)), "podAffinityTerm": pod_term(str(rng.choice([HOST, ZONE])))}]
        if rng.random() < 0.25:
            affinity["nodeAffinity"] = {"preferredDuringSchedulingIgnoredDuringExecution": [
                {"weight": int(rng.integers(1, 101)), "preference": {"matchExpressions": [{"key": "disk", "operator": "In", "values": ["ssd"]}]}},
                {"weight": int(rng.integers(1, 101)), "preference": {"matchExpressions": [{"key": "rank", "operator": "Gt", "values": ["4"]}]}}]}
        if rng.random() < 0.15:
            affinity.setdefault("nodeAffinity", {})["requiredDuringSchedulingIgnoredDuringExecution"] = {"nodeSelectorTerms": [
                {"matchExpressions": [{"key": ZONE, "operator": "In", "values": ["z0", "z1"]}]},
                {"matchExpressions": [{"key": "disk", "operator": "NotIn", "values": ["hdd"]}]}]}
        if affinity:
            spec["affinity"] = affinity
        if rng.random() < 0.15:
            spec["nodeSelector"] = {"disk": str(rng.choice(["ssd", "hdd"]))}
        if rng.random() < 0.3:
            spec["tolerations"] = [{"key": "dedicated", "operator": "Exists"}] if rng.random() < 0.5 else \
                [{"key": "slow", "operator": "Equal", "value": "true", "effect": "PreferNoSchedule"}]
        r = rng.random()
        if r < 0.2:
            spec["topologySpreadConstraints"] = [{"maxSkew": int(rng.integers(1, 3)), "topologyKey": ZONE,
                                                  "whenUnsatisfiable": "DoNotSchedule", "labelSelector": selector(app)}]
        elif r < 0.4:
            spec["topologySpreadConstraints"] = [
                {"maxSkew": int(rng.integers(1, 4)), "topologyKey": str(rng.choice([HOST, ZONE])),
                 "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": selector(app, tier)},
                {"maxSkew": 1, "topologyKey": ZONE, "whenUnsatisfiable": "ScheduleAnyway", "labelSelector": selector(app)}][:int(rng.integers(1, 3))]
        md = {"labels": {"app": app, "tier": tier}}
        if gpu and rng.random() < 0.4:
            md["annotations"] = {"alibabacloud.com/gpu-mem": str(rng.choice(["2Gi", "4Gi", "8Gi"])),
                                 "alibabacloud.com/gpu-count": str(int(rng.choice([1, 1, 2])))}
        kind = str(rng.choice(["Deployment", "StatefulSet", "ReplicaSet", "Job", "Pod"]))
        if local and rng.random() < 0.35:
            kind = "StatefulSet"
        name = f"{app}-{tier}-{w}"
        if kind == "Pod":
            workloads.append({"apiVersion": "v1", "kind": "Pod", "metadata": dict(md, name=name, namespace=ns), "spec": spec})
            continue
        body = {"template": {"metadata": md, "spec": spec}}
        if kind == "Job":
            body["completions"] = int(rng.integers(1, max_replicas))
        else:
            body["replicas"] = int(rng.integers(1, max_replicas + 1))
            body["selector"] = selector(app, tier)
        if local and kind == "StatefulSet" and rng.random() < 0.8:      # volumeClaimTemplates on Open-Local StorageClasses
            body["volumeClaimTemplates"] = [
                {"metadata": {"name": f"v{k}"},
                 "spec": {"storageClassName": str(rng.choice(["open-local-lvm", "open-local-lvm", "yoda-lvm-default", "open-local-device-ssd",
                                                             "open-local-device-hdd"])),
                          "resources": {"requests": {"storage": f"{int(rng.choice([5, 10, 20, 40, 60, 120]))}Gi"}}}}
                for k in range(int(rng.integers(1, 4)))]
        workloads.append({"apiVersion": "apps/v1", "kind": kind, "metadata": {"name": name, "namespace": ns}, "spec": body})
        if rng.random() < 0.4:       # a Service selecting the app: default (system) spread constraints for its pods
            services.append({"apiVersion": "v1", "kind": "Service", "metadata": {"name": f"svc-{w}", "namespace": ns},
                             "spec": {"selector": {"app": app}}})
    return nodes, workloads, services
