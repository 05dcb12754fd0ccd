"""Kubernetes objects -> the SoA inputs of include/simon_hip.h (the "flatten" half of the Go shim).

Everything about a (pod, node) pair that does not depend on placements is decided here, once, on the host:
  * pod request rows (computePodResourceRequest / non-zero requests), Open-Gpu-Share annotations;
  * pod classes = pods that are indistinguishable for the scheduler (replicas of one workload);
  * the static filter mask per (pod class, node): NodeUnschedulable, NodeName, TaintToleration, NodeAffinity, with the
    reason text of the first failing plugin;
  * node classes = nodes with equal allocatable and equal static scores for every pod class, and the per
    (pod class, node class) tables: SimonPlugin raw score, NodeAffinity preferred weight sum, intolerable
    PreferNoSchedule taints, NodePreferAvoidPods;
  * topology keys / domains and the interned affinity / spread terms with their per-class role lists.
Each step names the reference code it restates (V/ = vendor/k8s.io/kubernetes/pkg/scheduler/).
"""
from __future__ import annotations

import functools
import json
import re
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi, fiterror, gomath, k8s
from .quantity import ZERO, Quantity, parse_quantity, simon_raw_score

ZONE_KEY = "topology.kubernetes.io/zone"
SYSTEM_DEFAULT_SPREAD = ((k8s.LABEL_HOSTNAME, 3), (ZONE_KEY, 5))      # podtopologyspread/plugin.go:39-50
HARD_POD_AFFINITY_WEIGHT = 1                                           # V/apis/config/v1beta1/defaults.go:179
ANNO_PREFER_AVOID = "scheduler.alpha.kubernetes.io/preferAvoidPods"


class Unsupported(ValueError):
    """The workload uses a feature the engine does not model: the Go shim must route it to the Go path."""


@dataclass
class Flat:
    problem: capi.Problem
    node_names: List[str]
    pod_refs: List[Tuple[str, str]]                 # (namespace, name) per pod id
    pods: List[dict]
    static_reasons: Dict[int, str]
    scalar_names: List[str]
    pod_class_of: np.ndarray
    const_score: np.ndarray
    info: dict = field(default_factory=dict)


def _qcmp(a: Quantity, b: Quantity) -> int:
    s = min(a.scale, b.scale)
    x, y = a.value * 10 ** (a.scale - s), b.value * 10 ** (b.scale - s)
    return (x > y) - (x < y)


def pod_requests_quantities(pod: dict) -> Dict[str, Quantity]:
    """resourcehelper.PodRequestsAndLimits, vendor/k8s.io/kubectl/pkg/util/resource/resource.go:34-58 (requests half)."""
    spec = pod["spec"]
    reqs: Dict[str, Quantity] = {}
    for c in spec.get("containers") or []:
        for name, q in ((c.get("resources") or {}).get("requests") or {}).items():
            reqs[name] = reqs[name].add(parse_quantity(str(q))) if name in reqs else parse_quantity(str(q))
    for c in spec.get("initContainers") or []:
        for name, q in ((c.get("resources") or {}).get("requests") or {}).items():
            qq = parse_quantity(str(q))
            if name not in reqs or _qcmp(qq, reqs[name]) > 0:
                reqs[name] = qq
    for name, q in (spec.get("overhead") or {}).items():
        reqs[name] = reqs[name].add(parse_quantity(str(q))) if name in reqs else parse_quantity(str(q))
    return reqs


def _gpu_annotations(pod: dict) -> Tuple[int, int]:
    """GetGpuMemoryFromPodAnnotation / GetGpuCountFromPodAnnotation (pkg/type/open-gpu-share/utils/pod.go:56-97)."""
    a = pod["metadata"].get("annotations") or {}
    mem = cnt = 0
    if k8s.GPU_MEM in a:
        try:
            mem = parse_quantity(str(a[k8s.GPU_MEM])).int_value()
        except ValueError:
            mem = 0
    if k8s.GPU_COUNT in a:
        try:
            cnt = int(str(a[k8s.GPU_COUNT]))
        except ValueError:
            cnt = 0
    # no defaulting: without a gpu-count annotation reqGpuNum is 0 and AllocateGpuId finds nothing (gpunodeinfo.go:238-240) -- a pod that
    # asks for gpu-mem alone is unschedulable in the reference
    return mem, cnt if cnt >= 0 else 0


_ATOI = re.compile(r"^[+-]?[0-9]+$")


def gpu_index_annotation(pod: dict):
    """GetGpuIdFromAnnotation + GpuIdStrToIntList (pkg/type/open-gpu-share/utils/pod.go:35-53,100-115): the device ids of an
    alibabacloud.com/gpu-index annotation the pod ARRIVES with ("2", "0-0-1"), or None when it has none or an invalid one (the
    reference logs a warning and allocates normally, gpunodeinfo.go:247-253)."""
    a = pod["metadata"].get("annotations") or {}
    text = a.get(k8s.GPU_INDEX)
    if not text:
        return None
    ids = []
    for part in str(text).split("-"):
        # strconv.Atoi: one optional sign, then ASCII decimal digits only -- Python's int() also takes " 1", "1_0" and Unicode
        # digits, which the reference rejects (it then allocates normally)
        if not _ATOI.match(part):
            return None
        v = int(part)
        if not -(1 << 63) <= v < (1 << 63):           # Atoi reports ErrRange beyond int (64 bits on amd64)
            return None
        ids.append(v)
    return ids or None


def _term_namespaces(pod: dict, term: dict) -> Tuple[str, ...]:
    ns = term.get("namespaces") or []
    return tuple(sorted(ns)) if ns else (pod["metadata"]["namespace"],)      # V/framework/types.go:120-131


def _affinity_terms(pod: dict):
    """NewPodInfo (V/framework/types.go:143-176): required/preferred (anti-)affinity terms of a pod as
    (namespaces, selector-json, topologyKey[, weight])."""
    aff = pod["spec"].get("affinity") or {}
    out = {"req_aff": [], "req_anti": [], "pref_aff": [], "pref_anti": []}
    for kind, key in (("podAffinity", "aff"), ("podAntiAffinity", "anti")):
        a = aff.get(kind) or {}
        for t in a.get("requiredDuringSchedulingIgnoredDuringExecution") or []:
            out["req_" + key].append((_term_namespaces(pod, t), json.dumps(t.get("labelSelector"), sort_keys=True), t["topologyKey"]))
        for wt in a.get("preferredDuringSchedulingIgnoredDuringExecution") or []:
            t = wt["podAffinityTerm"]
            out["pref_" + key].append((_term_namespaces(pod, t), json.dumps(t.get("labelSelector"), sort_keys=True),
                                       t["topologyKey"], int(wt["weight"])))
    return out


@functools.lru_cache(maxsize=None)
def _parsed(text: str):
    """json.loads of a selector that travels as text inside the interned term tuples: parsed ONCE (the class x term matching below asked
    for it classes x terms times -- 670 000 json.loads on a 400-workload cluster; the parsed objects are only ever read)."""
    return json.loads(text)


def _matches_term(pod_ns: str, pod_labels: dict, namespaces, sel_json: str) -> bool:
    """schedutil.PodMatchesTermsNamespaceAndSelector, V/util/topologies.go:40-49."""
    return pod_ns in namespaces and k8s.label_selector_matches(_parsed(sel_json), pod_labels)


_SVC_INDEX: Dict[int, tuple] = {}


def _services_that_may_select(services, ns: str, labels: dict):
    """The Services of namespace `ns` whose selector could match `labels`, in list order (GetPodServices lists them in the lister's
    order; the merged label set does not depend on it).  A selector with pairs needs its first pair among the pod's labels: services are
    indexed by (namespace, that pair) once per list -- pod classes x services tests were 0.26 s of a 400-workload cluster's ingest."""
    key = id(services)
    hit = _SVC_INDEX.get(key)
    if hit is None or hit[0] is not services or hit[1] != len(services):
        by_pair: Dict[tuple, List[int]] = {}
        always: Dict[str, List[int]] = {}
        for i, svc in enumerate(services):
            sel = svc.get("spec", {}).get("selector")
            if sel is None:
                continue                                   # services with nil selectors match nothing (helper/spread.go:84-87)
            sns = svc["metadata"].get("namespace") or "default"
            if sel:
                k, v = next(iter(sel.items()))
                by_pair.setdefault((sns, k, v), []).append(i)
            else:
                always.setdefault(sns, []).append(i)
        _SVC_INDEX.clear()                                 # one list at a time: the index lives as long as its list is the one in use
        hit = _SVC_INDEX[key] = (services, len(services), by_pair, always)
    _, _, by_pair, always = hit
    idx = list(always.get(ns, ()))
    for k, v in labels.items():
        idx += by_pair.get((ns, k, v), ())
    return [services[i] for i in sorted(idx)]


def _default_spread_selector(pod: dict, services, replicasets, statefulsets):
    """helper.DefaultSelector (V/framework/plugins/helper/spread.go:28-95): merged Service selectors AND the selectors of
    the matching ReplicaSets / StatefulSets.  Returns a list of LabelSelector dicts to be ANDed (empty = no selector)."""
    ns, labels = pod["metadata"]["namespace"], pod["metadata"].get("labels") or {}
    parts = []
    merged = {}
    for svc in _services_that_may_select(services, ns, labels):
        sel = svc["spec"]["selector"]
        if all(labels.get(k) == v for k, v in sel.items()):
            merged.update(sel)
    if merged:
        parts.append({"matchLabels": merged})
    for coll in (replicasets, statefulsets):
        if not labels:
            break                                  # GetPodReplicaSets / GetPodStatefulSets: no labels -> no match
        for o in coll:
            if (o["metadata"].get("namespace") or "default") != ns:
                continue
            sel = o["spec"].get("selector")
            if sel is None or k8s.selector_is_empty(sel) or not k8s.label_selector_matches(sel, labels):
                continue
            parts.append(sel)
    return parts


def _spread_constraints(pod: dict, services, replicasets, statefulsets):
    """podtopologyspread: explicit constraints (filterTopologySpreadConstraints, common.go:73-91) or the system
    defaults over the default selector (buildDefaultConstraints, common.go:45-60).  Returns (hard, soft) lists of
    (selector-json, topologyKey, maxSkew)."""
    explicit = pod["spec"].get("topologySpreadConstraints") or []
    hard, soft = [], []
    if explicit:
        for c in explicit:
            item = (json.dumps([c.get("labelSelector")], sort_keys=True), c["topologyKey"], int(c["maxSkew"]))
            (hard if c.get("whenUnsatisfiable", "DoNotSchedule") == "DoNotSchedule" else soft).append(item)
        return hard, soft
    parts = _default_spread_selector(pod, services, replicasets, statefulsets)
    if parts:
        sel = json.dumps(parts, sort_keys=True)
        soft = [(sel, key, skew) for key, skew in SYSTEM_DEFAULT_SPREAD]
    return hard, soft


def _selectors_match(sel_list_json: str, labels: dict) -> bool:
    return all(k8s.label_selector_matches(sel, labels) for sel in _parsed(sel_list_json))


def _prefer_avoid(node: dict, pod: dict) -> int:
    """NodePreferAvoidPods.Score (nodepreferavoidpods/node_prefer_avoid_pods.go:58-82): 0 when the node's annotation lists
    the pod's RC / ReplicaSet controller, else 100."""
    refs = [r for r in pod["metadata"].get("ownerReferences") or [] if r.get("controller")]
    if not refs or refs[0].get("kind") not in ("ReplicationController", "ReplicaSet"):
        return 100
    anno = (node["metadata"].get("annotations") or {}).get(ANNO_PREFER_AVOID)
    if not anno:
        return 100
    try:
        avoids = json.loads(anno).get("preferAvoidPods") or []
    except ValueError:
        return 100
    for a in avoids:
        ctl = (a.get("podSignature") or {}).get("podController") or {}
        if ctl.get("kind") == refs[0]["kind"] and ctl.get("uid") == refs[0].get("uid"):
            return 0
    return 100


ANNO_NODE_LOCAL_STORAGE = "simon/node-local-storage"      # pkg/type/const.go
ANNO_POD_LOCAL_STORAGE = "simon/pod-local-storage"
LVM_SC_NAMES = ("open-local-lvm", "yoda-lvm-default")    # pkg/utils/const.go:5,12


def node_local_storage(node: dict):
    """utils.GetNodeStorage (pkg/utils/utils.go:519-531): None without the annotation, else (vgs, devices) with
    vgs = [(name, capacity, requested)], devices = [(capacity, media, allocated)]; media 1 = ssd, 2 = hdd, 0 = other."""
    anno = (node["metadata"].get("annotations") or {}).get(ANNO_NODE_LOCAL_STORAGE)
    if anno is None:
        return None
    d = json.loads(anno)
    vgs = [(v["name"], int(v.get("capacity", 0)), int(v.get("requested", 0) or 0)) for v in d.get("vgs") or []]
    devs = [(int(x.get("capacity", 0)), {"ssd": 1, "hdd": 2}.get(x.get("mediaType"), 0), str(x.get("isAllocated", "false")).lower() == "true")
            for x in d.get("devices") or []]
    return vgs, devs


def pod_local_volumes(pod: dict, storage_classes):
    """utils.GetPodLocalPVCs (pkg/utils/utils.go:580-623) + DivideLVMPVCs / DividePVCAccordingToMediaType (open-local
    algo/common.go:146-155,247-260): (lvm [(size, vgName or None)] with the VG-named ones first, ssd sizes, hdd sizes)."""
    anno = (pod["metadata"].get("annotations") or {}).get(ANNO_POD_LOCAL_STORAGE)
    if not anno:
        return None
    vols = json.loads(anno).get("volumes") or []
    if not vols:
        return None
    scs = {sc["metadata"]["name"]: sc for sc in storage_classes}
    with_vg, without_vg, ssd, hdd = [], [], [], []
    n_device_pvcs = 0
    for v in vols:
        if v.get("kind") not in ("LVM", "HDD", "SSD"):
            continue
        size, sc_name = int(v["size"]), v.get("scName", "")
        if size <= 0:
            raise Unsupported("Open-Local volume of size 0")
        params = (scs.get(sc_name) or {}).get("parameters") or {}
        if sc_name in LVM_SC_NAMES:
            (with_vg if params.get("vgName") else without_vg).append((size, params.get("vgName") or None))
        else:
            n_device_pvcs += 1
            media = params.get("mediaType")          # a PVC with another / no media type is skipped by the plugin (:247-260)
            if media == "ssd":
                ssd.append(size)
            elif media == "hdd":
                hdd.append(size)
    if n_device_pvcs and not (ssd or hdd):
        # ScoreDevice would divide by zero units (int(NaN) in Go): leave such pods to the Go path
        raise Unsupported("Open-Local device volumes whose StorageClasses carry no ssd/hdd mediaType")
    return with_vg + without_vg, sorted(ssd), sorted(hdd)


NODE_KEY = "\x00node"        # synthetic topology key whose domain is the node itself (NodePorts)


def host_ports(pod: dict):
    """getContainerPorts + HostPortInfo.sanitize (nodeports/node_ports.go:61-73, V/framework/types.go:816-823): the
    (ip, protocol, port) triples with hostPort > 0 of the pod's containers (init containers are not looked at)."""
    out = []
    for c in pod["spec"].get("containers") or []:
        for port in c.get("ports") or []:
            hp = int(port.get("hostPort") or 0)
            if hp > 0:
                out.append((port.get("hostIP") or "0.0.0.0", port.get("protocol") or "TCP", hp))
    return out


def _bitmask_rows(flags: np.ndarray, words: int) -> np.ndarray:
    """bool [R][N] -> uint64 [R][words], node j = bit j % 64 of word j / 64"""
    R, N = flags.shape
    padded = np.zeros((R, words * 64), bool)
    padded[:, :N] = flags
    return np.packbits(padded, axis=1, bitorder="little").view("<u8").astype(np.uint64).reshape(R, words)


def _bitmask(flags, words) -> np.ndarray:
    return _bitmask_rows(np.asarray(flags, bool)[None, :], words)[0]


def image_states(nodes_in_arrival_order: List[dict]) -> Dict[str, Dict[str, Tuple[int, int]]]:
    """node name -> {image name: (size, NumNodes)} as schedulerCache.addNodeImageStates leaves NodeInfo.ImageStates
    (V/internal/cache/cache.go:675-698): the summary of an image is taken when the node ARRIVES, so it counts the nodes that
    listed the image up to then; the size is the first lister's."""
    size: Dict[str, int] = {}
    count: Dict[str, int] = {}
    out = {}
    for n in nodes_in_arrival_order:
        mine = {}
        for img in (n.get("status") or {}).get("images") or []:
            for name in img.get("names") or []:
                if name not in size:
                    size[name], count[name] = int(img.get("sizeBytes") or 0), 0
                if name not in mine:
                    count[name] += 1
                    mine[name] = None
        out[n["metadata"]["name"]] = {name: (size[name], count[name]) for name in mine}
    return out


def image_locality_score(pod: dict, states: Dict[str, Tuple[int, int]], total_nodes: int) -> int:
    """ImageLocality.Score (imagelocality/image_locality.go:53-113): sum over the containers of
    int64(float64(size) * (float64(NumNodes) / float64(totalNumNodes))), clamped to [23 MB, 1000 MB x containers], scaled to 0..100."""
    mb = 1024 * 1024
    conts = pod["spec"].get("containers") or []
    if not conts:
        return 0
    ssum = 0
    for c in conts:
        name = str(c.get("image") or "")
        name = name if name.rfind(":") > name.rfind("/") else name + ":latest"          # normalizedImageName (:120-125)
        if name in states:
            sz, cnt = states[name]
            ssum += int(float(sz) * (float(cnt) / float(total_nodes)))
    hi = 1000 * mb * len(conts)
    ssum = min(max(ssum, 23 * mb), hi)
    return 100 * (ssum - 23 * mb) // (hi - 23 * mb)


def flatten(nodes: List[dict], pods: List[dict], services=(), replicasets=(), statefulsets=(),
            gates: Optional[List[int]] = None, storage_classes=(), image_total: Optional[int] = None,
            node_arrival_order: Optional[List[str]] = None) -> Flat:
    """nodes: the pool in canonical order (cluster nodes, then new-node clones).  pods: the stream in scheduling order;
    a pod with spec.nodeName is bound without filtering (V/eventhandlers.go:223-236).  gates[p] = node index the pod
    depends on (DaemonSet pods of new nodes, pkg/simulator/core.go:85-95) or -1."""
    N, P = len(nodes), len(pods)
    node_names = [n["metadata"]["name"] for n in nodes]
    node_index = {name: j for j, name in enumerate(node_names)}
    if len(node_index) != N:
        raise ValueError("duplicate node names")
    words = (N + 63) // 64
    # the (namespace, first selector pair) index of the Services is built once PER CALL: a list the caller edited in place between two
    # calls (same object, same length) must not be read through the previous call's index
    _SVC_INDEX.clear()

    # ---- pods: requests, classes ---------------------------------------------------------------------------
    # Ingest: every per-pod quantity is evaluated once per TEMPLATE.  workloads.py stamps the
    # replicas of one workload with a `_tmpl` token (identical but for metadata.name), so a 50 000-pod stream costs
    # one evaluation per workload plus integer gathers; pods without a token are their own template.
    tok_rep: Dict[int, int] = {}
    rep_list: List[int] = []                              # pod index of each distinct template
    tmpl_of = np.empty(P, np.int64)
    for i, p in enumerate(pods):
        tok = p.get("_tmpl")
        if tok is None:
            tmpl_of[i] = len(rep_list)
            rep_list.append(i)
        else:
            r = tok_rep.get(tok)
            if r is None:
                r = tok_rep[tok] = len(rep_list)
                rep_list.append(i)
            tmpl_of[i] = r
    # a DaemonSet pod is seen through its class affinity (the name requirement holds everywhere); the node it is pinned to
    # travels separately as pin_node
    tpods = [pods[i] if "_class_affinity" not in pods[i] else
             dict(pods[i], spec=dict(pods[i]["spec"], affinity=pods[i]["_class_affinity"])) for i in rep_list]
    if os.environ.get("SIMON_CHECK_TEMPLATES"):           # debug: the token promise, checked pod by pod
        for i, p in enumerate(pods):
            a, b = p, tpods[tmpl_of[i]]
            skip = ("affinity",) if "_class_affinity" in a else ()       # DaemonSet pods differ in the node their affinity names
            assert {k: v for k, v in a["spec"].items() if k not in skip} == {k: v for k, v in b["spec"].items() if k not in skip} and \
                a.get("_class_affinity") == b.get("_class_affinity") and \
                {k: v for k, v in a["metadata"].items() if k != "name"} == {k: v for k, v in b["metadata"].items() if k != "name"}, \
                f"pod {a['metadata']['name']} differs from its template"
    # (names that are neither cpu / memory / ephemeral-storage nor scalar resources never reach the scheduler's Resource: dropped here too)
    reqs = [{name: v for name, v in k8s.pod_request(p).items() if name in ("cpu", "memory", "ephemeral-storage") or k8s.is_scalar_resource_name(name)}
            for p in tpods]
    scalar_names = sorted({name for r in reqs for name, v in r.items()
                           if name not in ("cpu", "memory", "ephemeral-storage") and v != 0})
    if len(scalar_names) > capi.MAX_SCALAR:
        raise Unsupported(f"more than {capi.MAX_SCALAR} extended resources requested")
    # fitsRequest's shortcut tests len(podRequest.ScalarResources) == 0 (fit.go:244-249), and Resource.Add creates the map entry even for
    # a quantity of 0 (V/framework/types.go:320-322): a pod that names an extended resource with quantity 0 runs the resource checks in
    # Go (it can fail on a node preset pods over-committed), and the zero entry itself is compared.  The ENTRIES travel next to the
    # quantities since ABI v6 (simon_set_scalar_entries): bit k = scalar_names[k], bit 7 = a resource nobody requests a quantity of.
    ent_t = np.zeros(len(tpods), np.uint8)
    for t, r in enumerate(reqs):
        for name in r:
            if name in ("cpu", "memory", "ephemeral-storage"):
                continue
            ent_t[t] |= (1 << scalar_names.index(name)) if name in scalar_names else 0x80
    scalar_entries = ent_t[tmpl_of] if ent_t.any() else None
    req_cpu = np.array([r.get("cpu", 0) for r in reqs], np.int64)[tmpl_of]
    req_mem = np.array([r.get("memory", 0) for r in reqs], np.int64)[tmpl_of]
    req_eph = np.array([r.get("ephemeral-storage", 0) for r in reqs], np.int64)[tmpl_of]
    nz = [k8s.pod_nonzero_request(p) for p in tpods]
    nz_cpu = np.array([a for a, _ in nz], np.int64)[tmpl_of]
    nz_mem = np.array([b for _, b in nz], np.int64)[tmpl_of]
    scalar_req = np.array([[r.get(name, 0) for r in reqs] for name in scalar_names], np.int64).reshape(len(scalar_names), len(tpods))[:, tmpl_of]
    gpu = [_gpu_annotations(p) for p in tpods]
    gpu_mem = np.array([g[0] for g in gpu], np.int64)[tmpl_of]
    gpu_cnt = np.array([g[1] for g in gpu], np.int32)[tmpl_of]
    gpu_index = np.zeros(len(tpods), np.uint32)
    for t, p in enumerate(tpods):
        ids = gpu_index_annotation(p)
        if ids is None or gpu[t][0] <= 0:
            continue
        try:
            gpu_index[t] = capi.pack_gpu_index(ids)
        except ValueError as e:
            raise Unsupported(f"pod {p['metadata'].get('name')}: {e}") from None
    gpu_index = gpu_index[tmpl_of]
    preset = np.array([node_index.get(p["spec"].get("nodeName"), -1) if p["spec"].get("nodeName") else -1 for p in pods], np.int32)
    for i, p in enumerate(pods):
        if p["spec"].get("nodeName") and preset[i] < 0:
            raise Unsupported(f"pod {p['metadata']['name']} is bound to unknown node {p['spec']['nodeName']}")
    pin = np.array([node_index.get(p.get("_daemon_node"), -1) if "_class_affinity" in p else -1 for p in pods], np.int32)
    for i, p in enumerate(pods):
        if "_class_affinity" in p and (pin[i] < 0 or preset[i] >= 0):
            raise Unsupported(f"DaemonSet pod {p['metadata']['name']}: node {p.get('_daemon_node')} is not in the pool")

    class_ids: Dict[str, int] = {}
    class_rep: List[dict] = []
    rq_of_class: List[Dict[str, Quantity]] = []
    tmpl_class = np.empty(len(tpods), np.int32)
    for i, p in enumerate(tpods):
        spec, md = p["spec"], p["metadata"]
        owner = [r.get("kind") for r in md.get("ownerReferences") or [] if r.get("controller")]
        rq = pod_requests_quantities(p)
        key = json.dumps([md.get("namespace"), md.get("labels") or {}, spec.get("nodeSelector"), spec.get("affinity"),
                          spec.get("tolerations"), spec.get("topologySpreadConstraints"), owner,
                          {k: str(v) for k, v in rq.items()}, spec.get("overhead"), host_ports(p),
                          pod_local_volumes(p, storage_classes)], sort_keys=True)
        if key not in class_ids:
            class_ids[key] = len(class_rep)
            class_rep.append(p)
            rq_of_class.append(rq)
        tmpl_class[i] = class_ids[key]
    pod_class = tmpl_class[tmpl_of]          # class ids number the classes by first occurrence in the stream, as before
    Cp = len(class_rep)

    # ---- DefaultPreemption (V/scheduler.go:479, defaultpreemption/default_preemption.go) finds no victims as long as all pods share one
    # priority.  With differing spec.priority values it can act only in a scenario where some pod FAILS while a pod of lower priority
    # is placed: the priorities travel (ABI v6, simon_set_pod_priorities) and the engine flags exactly those scenarios
    # (simon_fetch_preempt_risk); every other scenario -- among them every one a capacity plan accepts -- is exact.
    prio_t = np.array([int(p["spec"].get("priority") or 0) for p in tpods], np.int32)
    priority = prio_t[tmpl_of] if len(set(prio_t.tolist())) > 1 else None

    # ---- ImageLocality (imagelocality/image_locality.go:53-113) is a constant 0 as long as no node lists an image a pod
    # runs.  Otherwise its score depends on the cluster size (spread = NumNodes / totalNumNodes): with ONE size
    # (image_total = the scenario's node count, simulate()) it is a static per (class, node) score that joins static_add;
    # a batch of different sizes is left to the Go path.
    wanted = set()
    for p in class_rep:
        for c in p["spec"].get("containers") or []:
            name = str(c.get("image") or "")
            wanted.add(name if name.rfind(":") > name.rfind("/") else name + ":latest")          # normalizedImageName (:120-125)
    img_nodes = [j for j, n in enumerate(nodes)
                 if any(wanted.intersection(img.get("names") or []) for img in (n.get("status") or {}).get("images") or [])]
    img_states = None
    if img_nodes:
        if image_total is None:
            raise Unsupported(f"node {node_names[img_nodes[0]]} lists image(s) the pods run: ImageLocality depends on the cluster size")
        by_name = {n["metadata"]["name"]: n for n in nodes}
        arrival = [by_name[x] for x in node_arrival_order] if node_arrival_order is not None else nodes
        img_states = image_states(arrival)

    # ---- static filters per (pod class, node): first failing plugin in registry order --------------------------
    # A class looks at a node only through the label keys its nodeSelector / node affinity name, the node's taints,
    # its unschedulable flag, its preferAvoidPods annotation and -- for matchFields -- whether its name is one of the
    # listed values.  Nodes are interned by that view ("node signature") and each class is evaluated once per
    # distinct signature; the (class, node) matrices are gathers.  A DaemonSet's per-node pods (one class each,
    # pinned by matchFields) cost O(signatures) instead of O(nodes) evaluations each.
    reason_ids: Dict[str, int] = {v: k for k, v in fiterror.DEFAULT_STATIC_REASONS.items()}

    def rid(text: str) -> int:
        if text not in reason_ids:
            if len(reason_ids) >= 255:
                raise Unsupported("more than 255 distinct static failure reasons")
            reason_ids[text] = len(reason_ids) + 1
        return reason_ids[text]

    def intern_col(values) -> np.ndarray:
        ids: Dict = {}
        return np.array([ids.setdefault(v, len(ids)) for v in values], np.int64)

    node_labels = [n["metadata"].get("labels") or {} for n in nodes]
    taint_sig = intern_col(json.dumps((n.get("spec") or {}).get("taints") or [], sort_keys=True) for n in nodes)
    unsched_sig = np.array([1 if (n.get("spec") or {}).get("unschedulable") else 0 for n in nodes], np.int64)
    avoid_sig = intern_col((n["metadata"].get("annotations") or {}).get(ANNO_PREFER_AVOID) or "" for n in nodes)
    fixed_sig = intern_col(zip(taint_sig.tolist(), unsched_sig.tolist(), avoid_sig.tolist()))
    group_cache: Dict[tuple, tuple] = {}

    def label_group(keys: tuple):
        """(first node of each signature, signature index per node) for the classes that read exactly `keys`."""
        g = group_cache.get(keys)
        if g is None:
            sig = fixed_sig if not keys else intern_col(zip(fixed_sig.tolist(), *[[lab.get(k) for lab in node_labels] for k in keys]))
            _, first, inv = np.unique(sig, return_index=True, return_inverse=True)
            g = group_cache[keys] = (sig, first, inv)
        return g

    def node_view_of(p: dict):
        """label keys and metadata.name values the class's node selector / affinity terms mention"""
        keys = set((p["spec"].get("nodeSelector") or {}).keys())
        names = []
        na_ = ((p["spec"].get("affinity") or {}).get("nodeAffinity") or {})
        terms_ = list((na_.get("requiredDuringSchedulingIgnoredDuringExecution") or {}).get("nodeSelectorTerms") or [])
        terms_ += [t.get("preference") or {} for t in na_.get("preferredDuringSchedulingIgnoredDuringExecution") or []]
        for t in terms_:
            keys.update(r["key"] for r in t.get("matchExpressions") or [])
            for r in t.get("matchFields") or []:
                names += [str(v) for v in r.get("values") or []]
        return tuple(sorted(keys)), names

    static_ok = np.ones((Cp, N), bool)
    static_reason = np.zeros((Cp, N), np.uint8)
    class_view: List[tuple] = []                   # per class: (first, inv, affinity_ok per signature)
    part = None                                    # node partition refined by every class that tells nodes apart
    score_cols: List[tuple] = []                   # per class: (inv, na, tt, npa per signature)
    unsched_tol = {"key": "node.kubernetes.io/unschedulable", "effect": "NoSchedule"}
    for c, p in enumerate(class_rep):
        keys, names = node_view_of(p)
        sig, first, inv = label_group(keys)
        if names:                                  # matchFields: the node's name matters only as "which listed value"
            extra = np.zeros(N, np.int64)
            for k, name in enumerate(dict.fromkeys(names)):
                if name in node_index:
                    extra[node_index[name]] = k + 1
            _, first, inv = np.unique(sig * (len(names) + 1) + extra, return_index=True, return_inverse=True)
        tolerates_unsched = any(k8s.toleration_tolerates(t, unsched_tol) for t in p["spec"].get("tolerations") or [])
        U = len(first)
        ok_u, reason_u, aff_u = np.ones(U, bool), np.zeros(U, np.uint8), np.ones(U, bool)
        na_u, tt_u, npa_u = np.zeros(U, np.int64), np.zeros(U, np.int64), np.zeros(U, np.int64)
        for u, j in enumerate(first.tolist()):
            node = nodes[j]
            aff_u[u] = k8s.pod_matches_node_selector_and_affinity(p, node)
            reason = None
            if (node.get("spec") or {}).get("unschedulable") and not tolerates_unsched:   # nodeunschedulable/node_unschedulable.go:51-66
                reason = "node(s) were unschedulable"
            else:
                taint = k8s.find_untolerated_taint(node, p)
                if taint is not None:
                    reason = fiterror.taint_reason(taint.get("key", ""), taint.get("value", "") or "")
                elif not aff_u[u]:
                    reason = "node(s) didn't match Pod's node affinity"
            if reason is not None:
                ok_u[u] = False
                reason_u[u] = rid(reason)
            na_u[u] = k8s.node_affinity_preferred_score(p, node)
            tt_u[u] = k8s.count_intolerable_prefer_no_schedule(node, p)
            npa_u[u] = _prefer_avoid(node, p)
        static_ok[c] = ok_u[inv]
        static_reason[c] = reason_u[inv]
        class_view.append((inv, aff_u))
        score_cols.append((inv, na_u, tt_u, npa_u))
    static_mask = _bitmask_rows(static_ok, words)

    def affinity_ok_of(c: int) -> np.ndarray:
        inv, aff_u = class_view[c]
        return aff_u[inv]

    # ---- nodes: allocatable, GPU capacity, static scores, classes ---------------------------------------------
    allocs = [k8s.node_allocatable(n) for n in nodes]
    alloc_cpu = np.array([a.get("cpu", 0) for a in allocs], np.int64)
    alloc_mem = np.array([a.get("memory", 0) for a in allocs], np.int64)
    alloc_eph = np.array([a.get("ephemeral-storage", 0) for a in allocs], np.int64)
    alloc_pods = np.array([a.get("pods", 0) for a in allocs], np.int32)
    scalar_alloc = np.array([[a.get(name, 0) for a in allocs] for name in scalar_names], np.int64).reshape(len(scalar_names), N)
    caps = [(n.get("status") or {}).get("capacity") or {} for n in nodes]
    node_gpu_cnt = np.array([int(parse_quantity(str(c[k8s.GPU_COUNT])).int_value()) if k8s.GPU_COUNT in c else 0 for c in caps], np.int32)
    node_gpu_mem = np.array([parse_quantity(str(c[k8s.GPU_MEM])).int_value() if k8s.GPU_MEM in c else 0 for c in caps], np.int64)
    if (node_gpu_cnt > capi.MAX_GPU_DEV).any():
        raise Unsupported(f"node with more than {capi.MAX_GPU_DEV} GPU devices")
    alloc_q = [{name: parse_quantity(str(q)) for name, q in ((n.get("status") or {}).get("allocatable") or {}).items()} for n in nodes]
    # node classes: nodes with equal allocatable quantities and equal (NodeAffinity, TaintToleration, NodePreferAvoidPods)
    # raw scores for EVERY pod class.  The partition starts from the allocatable key and is refined by each class whose
    # scores differ between signatures; classes are numbered by first occurrence in node order.
    part = intern_col(json.dumps({k: [q.value, q.scale, q.format] for k, q in sorted(a.items())}) for a in alloc_q)
    for inv, na_u, tt_u, npa_u in score_cols:
        if (na_u == na_u[0]).all() and (tt_u == tt_u[0]).all() and (npa_u == npa_u[0]).all():
            continue
        trip = intern_col(zip(na_u.tolist(), tt_u.tolist(), npa_u.tolist()))
        _, part = np.unique(part * (int(trip.max()) + 1) + trip[inv], return_inverse=True)
    img_cols: Dict[int, np.ndarray] = {}            # class -> ImageLocality score per node (only classes that score somewhere)
    if img_states is not None:
        for c, p in enumerate(class_rep):
            col = np.zeros(N, np.int64)
            for j in img_nodes:
                col[j] = image_locality_score(p, img_states[node_names[j]], image_total)
            if col.any():
                img_cols[c] = col
                _, part = np.unique(part * (int(col.max()) + 1) + col, return_inverse=True)
    _, first, inv = np.unique(part, return_index=True, return_inverse=True)
    rank = np.empty(len(first), np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    node_class = rank[inv].astype(np.int32)
    ncls_rep = np.sort(first).tolist()
    Cn = len(ncls_rep)
    rep_idx = np.array(ncls_rep, np.int64)
    na_t, tt_t, npa_t = np.zeros((Cp, Cn), np.int64), np.zeros((Cp, Cn), np.int64), np.zeros((Cp, Cn), np.int64)
    for c, (inv_, na_u, tt_u, npa_u) in enumerate(score_cols):
        at = inv_[rep_idx]
        na_t[c], tt_t[c], npa_t[c] = na_u[at], tt_u[at], npa_u[at]
    # Simon / Open-Gpu-Share raw score: only the resources the pod requests can raise the max share, so the value is
    # shared by every (class, node class) pair with the same request quantities and the same allocatable of those names
    simon_raw = np.zeros((Cp, Cn), np.int64)
    alloc_sub: Dict[tuple, tuple] = {}
    raw_rows: Dict[str, np.ndarray] = {}
    for c in range(Cp):
        rq = rq_of_class[c]
        # Quantity.AsApproximateFloat64 scales BinarySI by 2^(10*scale) and DecimalSI by 10^scale: the format is part of the value
        rq_key = json.dumps({k: [q.value, q.scale, q.format] for k, q in sorted(rq.items())})
        row = raw_rows.get(rq_key)
        if row is None:
            names = tuple(sorted(rq))
            sub = alloc_sub.get(names)
            if sub is None:
                ids = intern_col(tuple((alloc_q[j][k].value, alloc_q[j][k].scale, alloc_q[j][k].format) if k in alloc_q[j] else None for k in names)
                                 for j in ncls_rep)
                _, sfirst, sinv = np.unique(ids, return_index=True, return_inverse=True)
                sub = alloc_sub[names] = (sfirst, sinv)
            sfirst, sinv = sub
            row = raw_rows[rq_key] = np.array([simon_raw_score(rq, alloc_q[ncls_rep[d]]) for d in sfirst.tolist()], np.int64)[sinv]
        simon_raw[c] = row

    # constants the engine does not evaluate: ImageLocality 0, and the plugins that score alike on every node
    const = np.zeros(Cp, np.int64)
    prob_kw = {}
    if na_t.any():
        prob_kw["node_affinity_raw"] = na_t
    if tt_t.any():
        prob_kw["taint_prefer_raw"] = tt_t
    else:
        const += 100                                   # DefaultNormalizeScore(reverse) of all zeros
    img_t = np.zeros((Cp, Cn), np.int64)
    for c, col in img_cols.items():
        img_t[c] = col[rep_idx]
    if (npa_t != 100).any() or img_t.any():
        prob_kw["static_add"] = npa_t * 10000 + img_t            # NodePreferAvoidPods x 10000 + ImageLocality x 1, no NormalizeScore
    else:
        const += 100 * 10000

    # ---- topology keys, terms, role lists -------------------------------------------------------------------
    key_ids: Dict[str, int] = {}
    term_ids: Dict[tuple, int] = {}
    terms: List[tuple] = []                        # (kind, namespaces, selector-json, key index, node-set id)
    set_ids: Dict[bytes, int] = {}
    node_sets: List[np.ndarray] = []

    def key_id(k: str) -> int:
        if k not in key_ids:
            key_ids[k] = len(key_ids)
        return key_ids[k]

    def set_id(flags) -> int:
        m = _bitmask(flags, words)
        b = m.tobytes()
        if b not in set_ids:
            set_ids[b] = len(node_sets)
            node_sets.append(m)
        return set_ids[b]

    def term_id(kind: str, namespaces, sel_json: str, key: str, nset: int = -1) -> int:
        t = (kind, tuple(namespaces), sel_json, key_id(key), nset)
        if t not in term_ids:
            term_ids[t] = len(terms)
            terms.append(t)
        return term_ids[t]

    anti, aff, pref, own, hard, soft, flags = ([[] for _ in range(Cp)] for _ in range(7))
    # NodePorts: one term per (ip, protocol, port) triple in use anywhere, on the node-identity key
    all_ports = sorted({t for p in class_rep for t in host_ports(p)})
    port_conf = [[] for _ in range(Cp)]
    for (ip, proto, hp) in all_ports:
        term_id("port", (ip, proto, str(hp)), "", NODE_KEY)
    for c, p in enumerate(class_rep):
        for (ip, proto, hp) in host_ports(p):
            for (ip2, proto2, hp2) in all_ports:                      # HostPortInfo.CheckConflict (types.go:784-812)
                if (proto2, hp2) == (proto, hp) and (ip == "0.0.0.0" or ip2 in ("0.0.0.0", ip)):
                    port_conf[c].append(term_id("port", (ip2, proto2, str(hp2)), "", NODE_KEY))
    pref_w, own_w, hard_skew, hard_self, hard_set, soft_skew = ([[] for _ in range(Cp)] for _ in range(6))
    const_pts = np.zeros(Cp, np.int64)
    has_all_memo: Dict[tuple, np.ndarray] = {}
    for c, p in enumerate(class_rep):
        ns, labels = p["metadata"]["namespace"], p["metadata"].get("labels") or {}
        at = _affinity_terms(p)
        for (nss, sel, key) in at["req_anti"]:
            anti[c].append(term_id("sel", nss, sel, key))
        if at["req_aff"]:
            allsel = json.dumps(sorted([list(n_) + [s_] for n_, s_, _ in at["req_aff"]]), sort_keys=True)
            for (nss, sel, key) in at["req_aff"]:
                aff[c].append(term_id("all", (), allsel, key))
                own[c].append(term_id("sel", nss, sel, key))
                own_w[c].append(HARD_POD_AFFINITY_WEIGHT)
            flags[c] = all(_matches_term(ns, labels, n_, s_) for n_, s_, _ in at["req_aff"])
        else:
            flags[c] = False
        for (nss, sel, key, w) in at["pref_aff"]:
            t = term_id("sel", nss, sel, key)
            pref[c].append(t); pref_w[c].append(w)
            own[c].append(t); own_w[c].append(w)
        for (nss, sel, key, w) in at["pref_anti"]:
            t = term_id("sel", nss, sel, key)
            pref[c].append(t); pref_w[c].append(-w)
            own[c].append(t); own_w[c].append(-w)
        h, s = _spread_constraints(p, services, replicasets, statefulsets)
        if len(h) > capi.MAX_SPREAD or len(s) > capi.MAX_SPREAD:
            raise Unsupported(f"more than {capi.MAX_SPREAD} topology spread constraints of one kind")
        def has_all(cons):                            # nodes that carry every topology key of the constraints (memoised per key set)
            ks = tuple(sorted({k for _, k, _ in cons}))
            if ks not in has_all_memo:
                has_all_memo[ks] = np.array([all(k in lab for k in ks) for lab in node_labels], bool)
            return has_all_memo[ks]
        # The plugin keeps ONE counter per topology pair (TpPairToMatchNum / TopologyPairToPodCounts are keyed by
        # (key, value), filtering.go:253-268, scoring.go:129-160): constraints of one pod that share a topology key add
        # their matches into the same counter.  Such constraints therefore share a term whose selector LIST is matched
        # with multiplicity (a pod matching two of the selectors is counted twice).  The hostname key of the SOFT
        # constraints is counted per constraint in Score (scoring.go:190-192) and stays separate.
        def grouped(cons, merge_hostname):
            by_key: Dict[str, List[str]] = {}
            for sel, key, _ in cons:
                if merge_hostname or key != k8s.LABEL_HOSTNAME:
                    by_key.setdefault(key, []).append(sel)
            return by_key
        if h:
            elig = set_id(affinity_ok_of(c) & has_all(h))
            by_key = grouped(h, True)
            for (sel, key, skew) in h:
                hard[c].append(term_id("spread", (ns,), json.dumps(by_key[key]), key))
                hard_skew[c].append(skew)
                hard_self[c].append(int(_selectors_match(sel, labels)))
                hard_set[c].append(elig)
        if s:
            elig = set_id(affinity_ok_of(c) & has_all(s))
            by_key = grouped(s, False)
            seen_keys = set()
            for (sel, key, skew) in s:
                sels = by_key.get(key, [sel])
                soft[c].append(term_id("spread", (ns,), json.dumps(sels), key, elig))
                # initPreScoreState registers a pair once (scoring.go:86-96): only the first constraint of a key gets
                # the topology size, later ones see size 0
                dup = key != k8s.LABEL_HOSTNAME and key in seen_keys
                seen_keys.add(key)
                soft_skew[c].append(skew | (capi.SPREAD_DUP_KEY if dup else 0))
        else:
            const_pts[c] = 100 * 2                       # no constraints: every node scores MaxNodeScore (scoring.go:242-246)
    const += const_pts

    def class_matches(c: int, t: tuple) -> bool:
        p = class_rep[c]
        ns, labels = p["metadata"]["namespace"], p["metadata"].get("labels") or {}
        kind, nss, sel = t[0], t[1], t[2]
        if kind == "port":                                          # HostPortInfo.Add is a set: one count per distinct triple
            return int((nss[0], nss[1], int(nss[2])) in set(host_ports(p)))
        if kind == "sel":
            return _matches_term(ns, labels, nss, sel)
        if kind == "all":
            return all(_matches_term(ns, labels, tuple(x[:-1]), x[-1]) for x in _parsed(sel))
        if ns not in nss:                                           # countPodsMatchSelector: same namespace (common.go:93-105)
            return 0
        return sum(1 for s_ in _parsed(sel) if _selectors_match(s_, labels))   # multiplicity, see `grouped` above
    # Which terms can a class match at all?  A selector with matchLabels needs one of its pairs among the pod's labels: terms are indexed
    # by such a pair, the rest (match-everything selectors, expressions only) are always tried -- classes x terms tests become
    # classes x (a handful) on clusters of many workloads.
    def required_pair(sel):
        for kv in ((sel or {}).get("matchLabels") or {}).items():
            return kv
        return None
    by_pair: Dict[tuple, List[int]] = {}
    always: List[int] = []
    port_terms: List[int] = []
    for ti, t in enumerate(terms):
        kind, sel = t[0], t[2]
        if kind == "port":
            port_terms.append(ti)
            continue
        if kind == "sel":
            pairs = [required_pair(_parsed(sel))]
        elif kind == "all":
            xs = _parsed(sel)
            pairs = [required_pair(_parsed(xs[0][-1]))] if xs else [None]
        else:                                           # spread terms: a list of selector lists, each of which may match (multiplicity)
            pairs = []
            for s_ in _parsed(sel):
                sels = _parsed(s_)
                pairs.append(required_pair(sels[0]) if sels else None)
            if not pairs:
                continue                                # nothing to match: count 0
        if any(pr is None for pr in pairs):
            always.append(ti)
        else:
            for pr in set(pairs):
                by_pair.setdefault(pr, []).append(ti)
    match_memo: Dict[str, list] = {}                # a class matches terms through its namespace, labels and host ports only
    match = []
    for c, p in enumerate(class_rep):
        labels = p["metadata"].get("labels") or {}
        mk = json.dumps([p["metadata"]["namespace"], labels, host_ports(p)], sort_keys=True)
        if mk not in match_memo:
            cand = set(always)
            for kv in labels.items():
                cand.update(by_pair.get(kv, ()))
            if host_ports(p):
                cand.update(port_terms)
            match_memo[mk] = [t for t in sorted(cand) for _ in range(int(class_matches(c, terms[t])))]
            if os.environ.get("SIMON_CHECK_MATCH_INDEX"):      # tests: the index must not lose a match
                assert match_memo[mk] == [t for t in range(len(terms)) for _ in range(int(class_matches(c, terms[t])))], (c, mk)
        match.append(match_memo[mk])

    T = len(terms)
    Kt = len(key_ids)
    if T:
        keys = sorted(key_ids, key=key_ids.get)
        topo_dom = np.full((Kt, N), -1, np.int32)
        topo_n = np.zeros(Kt, np.int32)
        for k, key in enumerate(keys):
            vals: Dict[str, int] = {}
            for j, n in enumerate(nodes):
                lab = n["metadata"].get("labels") or {}
                if key == NODE_KEY:
                    topo_dom[k, j] = j
                elif key in lab:
                    topo_dom[k, j] = vals.setdefault(lab[key], len(vals))
            topo_n[k] = N if key == NODE_KEY else max(len(vals), 1)

        def csr(lists):
            off = np.cumsum([0] + [len(x) for x in lists]).astype(np.int32)
            flat = [v for x in lists for v in x]
            return off, np.array(flat if flat else [0], np.int32)
        prob_kw.update(topo_dom=topo_dom, topo_n_dom=topo_n, term_topo_key=np.array([t[3] for t in terms], np.int32),
                       term_node_set=np.array([t[4] for t in terms], np.int32),
                       topo_is_hostname=np.array([k == k8s.LABEL_HOSTNAME for k in keys], np.uint8))
        if node_sets:
            prob_kw["node_sets"] = np.stack(node_sets)
        prob_kw["match_off"], prob_kw["match_idx"] = csr(match)
        prob_kw["anti_off"], prob_kw["anti_idx"] = csr(anti)
        if any(port_conf):
            prob_kw["port_off"], prob_kw["port_idx"] = csr(port_conf)
        if any(aff):
            prob_kw["aff_off"], prob_kw["aff_idx"] = csr(aff)
            prob_kw["class_flags"] = np.array([capi.CLASS_AFF_SELF if f else 0 for f in flags], np.uint8)
        if any(pref) or any(own):
            prob_kw["pref_off"], prob_kw["pref_idx"] = csr(pref)
            prob_kw["pref_w"] = csr(pref_w)[1]
            prob_kw["own_off"], prob_kw["own_idx"] = csr(own)
            prob_kw["own_w"] = csr(own_w)[1]
        if any(hard):
            prob_kw["spread_hard_off"], prob_kw["spread_hard_idx"] = csr(hard)
            prob_kw["spread_hard_skew"] = csr(hard_skew)[1]
            prob_kw["spread_hard_self"] = csr(hard_self)[1]
            prob_kw["spread_hard_set"] = csr(hard_set)[1]
        if any(soft):
            prob_kw["spread_soft_off"], prob_kw["spread_soft_idx"] = csr(soft)
            prob_kw["spread_soft_skew"] = csr(soft_skew)[1]
            prob_kw["spread_log"] = gomath.spread_log_table(N)
    # InterPodAffinity scores 0 everywhere when nobody owns a scoring term; ImageLocality / Open-Local are constants 0

    # ---- Open-Local: node storage and per-class volume specs -------------------------------------------------
    vols = [pod_local_volumes(p, storage_classes) for p in class_rep]
    vg_name_list: List[str] = []                    # interned volume-group names (local_vg_name / lvm_vg ids -> text, for the reasons)
    if any(v is not None for v in vols):
        storage = [node_local_storage(n) for n in nodes]
        vg_names: Dict[str, int] = {}
        lf = np.zeros(N, np.int32); vcnt = np.zeros(N, np.int32); vcap = np.zeros((N, capi.MAX_VG), np.int64)
        vreq = np.zeros((N, capi.MAX_VG), np.int64); vname = np.full((N, capi.MAX_VG), -1, np.int32)
        dcnt = np.zeros(N, np.int32); dcap = np.zeros((N, capi.MAX_LDEV), np.int64); dmedia = np.zeros(N, np.int32); dalloc = np.zeros(N, np.int32)
        for j, st in enumerate(storage):
            if st is None:
                continue
            vgs, devs = st
            if len(vgs) > capi.MAX_VG or len(devs) > capi.MAX_LDEV:
                raise Unsupported(f"node {node_names[j]}: more than {capi.MAX_VG} volume groups or {capi.MAX_LDEV} devices")
            lf[j] = 1
            vcnt[j], dcnt[j] = len(vgs), len(devs)
            for v, (name, cap, req) in enumerate(vgs):
                vcap[j, v], vreq[j, v], vname[j, v] = cap, req, vg_names.setdefault(name, len(vg_names))
            for d, (cap, media, alloc) in enumerate(devs):
                dcap[j, d] = cap
                dmedia[j] |= media << (2 * d)
                dalloc[j] |= int(alloc) << d
        spec_ids: Dict[str, int] = {}
        specs = []
        spec_of = np.full(Cp, -1, np.int32)
        for c, v in enumerate(vols):
            if v is None:
                continue
            lvm, ssd, hdd = v
            if max(len(lvm), len(ssd), len(hdd)) > capi.MAX_LVOL:
                raise Unsupported(f"more than {capi.MAX_LVOL} Open-Local volumes of one kind in a pod")
            key = json.dumps(v)
            if key not in spec_ids:
                spec_ids[key] = len(specs)
                row = np.zeros((), capi.LOCAL_SPEC_DTYPE)
                row["n_lvm"], row["n_ssd"], row["n_hdd"] = len(lvm), len(ssd), len(hdd)
                row["lvm_vg"][:] = -1
                for k, (size, name) in enumerate(lvm):
                    row["lvm_size"][k] = size
                    row["lvm_vg"][k] = -1 if name is None else vg_names.setdefault(name, len(vg_names))
                row["ssd_size"][:len(ssd)] = ssd
                row["hdd_size"][:len(hdd)] = hdd
                specs.append(row)
            spec_of[c] = spec_ids[key]
        vg_name_list = sorted(vg_names, key=vg_names.get)
        prob_kw.update(local_flags=lf, local_vg_cnt=vcnt, local_vg_cap=vcap, init_vg_req=vreq, local_vg_name=vname,
                       local_dev_cnt=dcnt, local_dev_cap=dcap, local_dev_media=dmedia, init_dev_alloc=dalloc,
                       local_spec_of=spec_of, local_specs=np.array(specs, capi.LOCAL_SPEC_DTYPE))

    prob = capi.Problem(
        alloc_cpu=alloc_cpu, alloc_mem=alloc_mem, alloc_pods=alloc_pods, alloc_eph=alloc_eph if alloc_eph.any() or req_eph.any() else None,
        node_class=node_class, scalar_alloc=scalar_alloc if scalar_names else None,
        gpu_cnt=node_gpu_cnt if (node_gpu_cnt.any() or gpu_mem.any()) else None,
        gpu_mem_total=node_gpu_mem if (node_gpu_cnt.any() or gpu_mem.any()) else None,
        req_cpu=req_cpu, req_mem=req_mem, req_eph=req_eph if req_eph.any() else None, nz_cpu=nz_cpu, nz_mem=nz_mem,
        scalar_req=scalar_req if scalar_names else None, scalar_entries=scalar_entries, priority=priority, pod_class=pod_class,
        preset_node=preset if (preset >= 0).any() else None,
        gate_node=np.array(gates, np.int32) if gates is not None and any(g >= 0 for g in gates) else None,
        pin_node=pin if (pin >= 0).any() else None,
        gpu_mem=gpu_mem if gpu_mem.any() else None, pod_gpu_cnt=gpu_cnt if gpu_mem.any() else None,
        gpu_index=gpu_index if gpu_index.any() else None,
        n_pod_classes=Cp, n_node_classes=Cn, static_mask=None if static_ok.all() else static_mask,
        static_reason=None if static_ok.all() else static_reason, simon_raw=simon_raw, const_score=const, **prob_kw).normalise()
    return Flat(problem=prob, node_names=node_names,
                pod_refs=[(p["metadata"]["namespace"], p["metadata"]["name"]) for p in pods], pods=pods,
                static_reasons={v: k for k, v in reason_ids.items()}, scalar_names=scalar_names,
                pod_class_of=pod_class, const_score=const,
                info={"n_pod_classes": Cp, "n_node_classes": Cn, "n_terms": T, "n_topology_keys": Kt,
                      "vg_names": vg_name_list})
