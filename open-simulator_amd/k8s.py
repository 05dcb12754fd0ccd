"""Host-side Kubernetes object semantics the flattening step needs (plain dicts as parsed from YAML/JSON).

In the Go host none of this exists: the shim calls the real apimachinery / scheduler helpers.  The Python mirror
restates exactly the slices the scheduling path depends on, each function naming the reference code it follows
(paths relative to the open-simulator tree; V/ = vendor/k8s.io/kubernetes/pkg/scheduler/).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import yaml

from .quantity import parse_quantity

LABEL_HOSTNAME = "kubernetes.io/hostname"
LABEL_ZONE = "topology.kubernetes.io/zone"
LABEL_ZONE_BETA = "failure-domain.beta.kubernetes.io/zone"
LABEL_REGION = "topology.kubernetes.io/region"
LABEL_REGION_BETA = "failure-domain.beta.kubernetes.io/region"

DEFAULT_MILLI_CPU_REQUEST = 100                 # V/util/non_zero.go:35-38
DEFAULT_MEMORY_REQUEST = 200 * 1024 * 1024

GPU_MEM = "alibabacloud.com/gpu-mem"            # pkg/type/open-gpu-share/utils/const.go
GPU_COUNT = "alibabacloud.com/gpu-count"
GPU_INDEX = "alibabacloud.com/gpu-index"        # device ids of a GPU pod, written at Reserve (utils/pod.go:117-127)
GPU_ASSUME_TIME = "alibabacloud.com/assume-time"
GPU_MODEL = "alibabacloud.com/gpu-card-model"
ANNO_NODE_GPU_SHARE = "simon/node-gpu-share"      # pkg/type/const.go


# ---------------------------------------------------------------------------------------------
# YAML ingest: utils.GetYamlContentFromDirectory + simulator.GetObjectFromYamlContent
# (pkg/utils/utils.go:116-130, pkg/simulator/utils.go:233-275)
# ---------------------------------------------------------------------------------------------
KINDS = ("Node", "Pod", "DaemonSet", "StatefulSet", "Deployment", "Service", "PersistentVolumeClaim", "ReplicaSet", "Job",
         "CronJob", "StorageClass", "PodDisruptionBudget", "ConfigMap")


def parse_file_paths(path: str) -> List[str]:
    """utils.ParseFilePath (pkg/utils/utils.go:56-84): a file, or every file under a directory (walk order = sorted)."""
    if os.path.isfile(path):
        return [path]
    out = []
    for root, dirs, files in os.walk(path):
        dirs.sort()
        for f in sorted(files):
            out.append(os.path.join(root, f))
    return out


def load_objects(path: str) -> List[dict]:
    objs = []
    for fp in parse_file_paths(path):
        if os.path.splitext(fp)[1] not in (".yaml", ".yml"):
            continue
        with open(fp) as f:
            for doc in yaml.safe_load_all(f):
                if isinstance(doc, dict) and doc.get("kind") == "List":
                    objs.extend(d for d in doc.get("items") or [] if isinstance(d, dict))
                elif isinstance(doc, dict) and doc.get("kind"):
                    objs.append(doc)
    return objs


def group_resources(objs: Iterable[dict]) -> Dict[str, List[dict]]:
    """ResourceTypes (pkg/simulator/core.go:38-52) as {kind: [objects]} in file order; unknown kinds are ignored."""
    res = {k: [] for k in KINDS}
    for o in objs:
        if o.get("kind") in res:
            res[o["kind"]].append(o)
    return res


# ---------------------------------------------------------------------------------------------
# label selectors (metav1.LabelSelectorAsSelector + labels.Selector.Matches)
# ---------------------------------------------------------------------------------------------
def label_selector_matches(sel: Optional[dict], labels: Optional[dict]) -> bool:
    """nil selector matches nothing, empty selector matches everything (apimachinery meta/v1/helpers.go:33-40)."""
    if sel is None:
        return False
    labels = labels or {}
    for k, v in (sel.get("matchLabels") or {}).items():
        if labels.get(k) != v:
            return False
    for e in sel.get("matchExpressions") or []:
        key, op, vals = e["key"], e["operator"], e.get("values") or []
        if op == "In":
            if key not in labels or labels[key] not in vals:
                return False
        elif op == "NotIn":
            if key in labels and labels[key] in vals:
                return False
        elif op == "Exists":
            if key not in labels:
                return False
        elif op == "DoesNotExist":
            if key in labels:
                return False
        else:
            raise ValueError(f"unsupported label selector operator {op!r}")
    return True


def selector_is_empty(sel: Optional[dict]) -> bool:
    return sel is not None and not (sel.get("matchLabels") or sel.get("matchExpressions"))


# ---------------------------------------------------------------------------------------------
# node selector terms (vendor/k8s.io/component-helpers/scheduling/corev1/nodeaffinity/nodeaffinity.go)
# ---------------------------------------------------------------------------------------------
def _node_requirement_matches(req: dict, labels: dict) -> bool:
    key, op, vals = req["key"], req["operator"], req.get("values") or []
    if op == "In":
        return key in labels and labels[key] in vals
    if op == "NotIn":
        return not (key in labels and labels[key] in vals)
    if op == "Exists":
        return key in labels
    if op == "DoesNotExist":
        return key not in labels
    if op in ("Gt", "Lt"):                     # labels/selector.go:215-240: both sides must parse as int64
        if key not in labels or len(vals) != 1:
            return False
        try:
            lv, rv = int(labels[key]), int(vals[0])
        except ValueError:
            return False
        return lv > rv if op == "Gt" else lv < rv
    raise ValueError(f"unsupported node selector operator {op!r}")


def node_selector_term_matches(term: dict, node: dict) -> bool:
    """nodeSelectorTerm.match: AND of matchExpressions (labels) and matchFields (metadata.name); empty term -> no match."""
    exprs, fields = term.get("matchExpressions") or [], term.get("matchFields") or []
    if not exprs and not fields:
        return False
    labels = node["metadata"].get("labels") or {}
    if any(not _node_requirement_matches(r, labels) for r in exprs):
        return False
    node_fields = {"metadata.name": node["metadata"]["name"]}
    return all(_node_requirement_matches(r, node_fields) for r in fields)


def pod_matches_node_selector_and_affinity(pod: dict, node: dict) -> bool:
    """PodMatchesNodeSelectorAndAffinityTerms, V/framework/plugins/helper/node_affinity.go:27-70."""
    spec = pod["spec"]
    labels = node["metadata"].get("labels") or {}
    for k, v in (spec.get("nodeSelector") or {}).items():
        if labels.get(k) != v:
            return False
    na = ((spec.get("affinity") or {}).get("nodeAffinity") or {})
    req = na.get("requiredDuringSchedulingIgnoredDuringExecution")
    if req is not None:
        return any(node_selector_term_matches(t, node) for t in req.get("nodeSelectorTerms") or [])
    return True


def node_affinity_preferred_score(pod: dict, node: dict) -> int:
    """NodeAffinity.Score, V/framework/plugins/nodeaffinity/node_affinity.go:77-103: sum of the weights of matching terms."""
    na = ((pod["spec"].get("affinity") or {}).get("nodeAffinity") or {})
    total = 0
    for t in na.get("preferredDuringSchedulingIgnoredDuringExecution") or []:
        if t.get("weight", 0) == 0:
            continue
        if node_selector_term_matches(t.get("preference") or {}, node):
            total += int(t["weight"])
    return total


# ---------------------------------------------------------------------------------------------
# taints and tolerations (vendor/k8s.io/api/core/v1/toleration.go:37-56, v1helper.FindMatchingUntoleratedTaint)
# ---------------------------------------------------------------------------------------------
def toleration_tolerates(tol: dict, taint: dict) -> bool:
    if tol.get("effect") and tol["effect"] != taint.get("effect"):
        return False
    if tol.get("key") and tol["key"] != taint.get("key"):
        return False
    op = tol.get("operator") or "Equal"
    if op == "Equal":
        return (tol.get("value") or "") == (taint.get("value") or "")
    return op == "Exists"


def find_untolerated_taint(node: dict, pod: dict, effects=("NoSchedule", "NoExecute")) -> Optional[dict]:
    """TaintToleration.Filter, V/framework/plugins/tainttoleration/taint_toleration.go:54-71."""
    tols = pod["spec"].get("tolerations") or []
    for taint in node.get("spec", {}).get("taints") or []:
        if taint.get("effect") not in effects:
            continue
        if not any(toleration_tolerates(t, taint) for t in tols):
            return taint
    return None


def count_intolerable_prefer_no_schedule(node: dict, pod: dict) -> int:
    """TaintToleration.Score, taint_toleration.go:85-151: tolerations with empty or PreferNoSchedule effect count."""
    tols = [t for t in pod["spec"].get("tolerations") or [] if not t.get("effect") or t["effect"] == "PreferNoSchedule"]
    n = 0
    for taint in node.get("spec", {}).get("taints") or []:
        if taint.get("effect") != "PreferNoSchedule":
            continue
        if not any(toleration_tolerates(t, taint) for t in tols):
            n += 1
    return n


# ---------------------------------------------------------------------------------------------
# resource requests
# ---------------------------------------------------------------------------------------------
def _res(container: dict) -> dict:
    return (container.get("resources") or {}).get("requests") or {}


def _value(q, name: str) -> int:
    """Quantity -> int64 the way the scheduler reads it: MilliValue for cpu, Value otherwise (both ceil)."""
    qq = parse_quantity(str(q))
    return qq.milli_value() if name == "cpu" else qq.int_value()


def pod_request(pod: dict) -> Dict[str, int]:
    """computePodResourceRequest, V/framework/plugins/noderesources/fit.go:148-165: sum of containers, max with each
    init container, plus overhead.  Keys: cpu (milli), memory, ephemeral-storage, and every extended resource."""
    spec = pod["spec"]
    total: Dict[str, int] = {}
    for c in spec.get("containers") or []:
        for name, q in _res(c).items():
            total[name] = total.get(name, 0) + _value(q, name)
    for c in spec.get("initContainers") or []:
        for name, q in _res(c).items():
            total[name] = max(total.get(name, 0), _value(q, name))
    for name, q in (spec.get("overhead") or {}).items():
        total[name] = total.get(name, 0) + _value(q, name)
    return total


def is_scalar_resource_name(name: str) -> bool:
    """schedutil.IsScalarResourceName (V/util/utils.go:169-172): extended resources (a domain prefix outside kubernetes.io/, not
    `requests.`-prefixed: v1helper.IsExtendedResourceName, pkg/apis/core/v1/helper/helpers.go:37-47 -- the qualified-name syntax is
    the API server's business), hugepages-*, *kubernetes.io/* and attachable-volumes-*.  Resource.Add (V/framework/types.go:310-326)
    keeps only these next to cpu / memory / pods / ephemeral-storage: any other name a container lists is dropped."""
    native = "/" not in name or "kubernetes.io/" in name
    extended = not native and not name.startswith("requests.")
    return extended or name.startswith("hugepages-") or "kubernetes.io/" in name or name.startswith("attachable-volumes-")


def pod_nonzero_request(pod: dict) -> (int, int):
    """calculateResource's non-zero cpu / memory (V/framework/types.go:601-636, V/util/non_zero.go:41-84)."""
    spec = pod["spec"]

    def nz(c):
        r = _res(c)
        cpu = _value(r["cpu"], "cpu") if "cpu" in r else DEFAULT_MILLI_CPU_REQUEST
        mem = _value(r["memory"], "memory") if "memory" in r else DEFAULT_MEMORY_REQUEST
        return cpu, mem
    cpu = mem = 0
    for c in spec.get("containers") or []:
        a, b = nz(c)
        cpu += a
        mem += b
    for c in spec.get("initContainers") or []:
        a, b = nz(c)
        cpu, mem = max(cpu, a), max(mem, b)
    ov = spec.get("overhead") or {}
    if "cpu" in ov:
        cpu += _value(ov["cpu"], "cpu")
    if "memory" in ov:
        mem += _value(ov["memory"], "memory")
    return cpu, mem


def node_allocatable(node: dict) -> Dict[str, int]:
    """NodeInfo.SetNode -> NewResource(node.Status.Allocatable), V/framework/types.go:295-326,654-660."""
    alloc = (node.get("status") or {}).get("allocatable") or {}
    return {name: _value(q, name) for name, q in alloc.items()}


def zone_key(node: dict) -> str:
    """utilnode.GetZoneKey (vendor/k8s.io/kubernetes/pkg/util/node/node.go:184-210)."""
    labels = node["metadata"].get("labels") or {}
    zone = labels[LABEL_ZONE_BETA] if LABEL_ZONE_BETA in labels else labels.get(LABEL_ZONE, "")
    region = labels[LABEL_REGION_BETA] if LABEL_REGION_BETA in labels else labels.get(LABEL_REGION, "")
    if region == "" and zone == "":
        return ""
    return region + ":\x00:" + zone


def canonical_node_order(nodes: List[dict]) -> List[int]:
    """nodeTree.list(), V/internal/cache/node_tree.go:119-143: zones in first-appearance order, one node per zone per round."""
    zones: List[str] = []
    tree: Dict[str, List[int]] = {}
    for i, n in enumerate(nodes):
        z = zone_key(n)
        if z not in tree:
            zones.append(z)
            tree[z] = []
        tree[z].append(i)
    out: List[int] = []
    k = 0
    while len(out) < len(nodes):
        for z in zones:
            if k < len(tree[z]):
                out.append(tree[z][k])
        k += 1
    return out
