#!/usr/bin/env python3
"""compare_ref.py -- TEST INFRASTRUCTURE (part of the oracle): pins the C oracle + the Python host mirror against output of the
REFERENCE ITSELF, determinised (integration/go/parity/determinise.patch: first-maximum selectHost, one filter worker).

Two halves, driven by oracle/run_ref.sh:

  python oracle/compare_ref.py cases  --ref /root/reference --work oracle/_ref/work
      writes oracle/_ref/work/cases.json -- the manifest integration/go/parity/parity_test.go reads: the reference's own example/
      inputs (cluster / application / new-node directories under <ref>/example) and seeded random clusters from
      tests/randk8s.py written out as YAML directories.

  python oracle/compare_ref.py compare --work oracle/_ref/work --ref-out oracle/_ref/work/ref_out.json
      for every case: the same YAML through the mirror (k8s.py / workloads.py / flatten.py), pods put in the ORDER THE REFERENCE
      SUBMITTED THEM (the trace; the reference's sort.Sort result and its goroutine-ordered expansion are data, not re-derived),
      the C oracle run on the flattened problem, and a pod-by-pod diff of bound node / FitError text.  Also reports whether
      gosort.py (the go1.18 sort.Sort restatement) reproduces the reference's app-pod order.
      Exit code 0 = every case identical, 1 = differences, 2 = could not run.  --engine hip|both puts the HIP library (GPU box) in the
      oracle's place / next to it (env SIMON_PARITY_ENGINE in run_ref.sh).

  python oracle/compare_ref.py selftest --work <dir>
      no Go needed: manufactures a `ref_out.json` from tests/pyref_sched.py (the object-level restatement, which shares nothing
      with the SoA path) with shuffled pod order and renamed clones, runs `compare` on it (must pass), corrupts one placement
      (must fail).  tests/test_ref_parity.py runs this on CPU.

Nothing here is on the product path."""
import argparse
import copy
import json
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from open_simulator_amd import capi, fiterror, flatten as fl, k8s, simulate as sim, workloads as wl  # noqa: E402

EXAMPLE_CASES = [    # (name, cluster dir, [(app name, app dir)], new-node dir, clones) relative to <ref>/example
    ("example_simple", "cluster/demo_1", [("simple", "application/simple")], "", 0),
    ("example_simple_2new", "cluster/demo_1", [("simple", "application/simple")], "newnode/demo_1", 2),
    ("example_all_6new", "cluster/demo_1", [("simple", "application/simple"), ("complicate", "application/complicate"),
                                            ("more_pods", "application/more_pods")], "newnode/demo_1", 6),
    ("example_gpushare", "cluster/gpushare", [("gpushare", "application/gpushare")], "", 0),
    ("example_gpushare_1new", "cluster/gpushare", [("gpushare", "application/gpushare")], "newnode/gpushare", 1),
    # Open-Local: 4 replicas with LVM + one HDD device each on demo_1 -- 3 fail without new nodes (their reasons are
    # open-local's err.Error() texts with sizes, simon_explain_local_detail), all fit with 3 clones of the storage-carrying template
    ("example_open_local", "cluster/demo_1", [("open_local", "application/open_local")], "", 0),
    ("example_open_local_3new", "cluster/demo_1", [("open_local", "application/open_local")], "newnode/demo_1", 3),
]
RANDOM_SEEDS = [(3, {}), (14, {}), (15, {"gpu": True}), (112, {}), (201, {"gpu": True}), (7, {"n_nodes": 40, "n_workloads": 30, "max_replicas": 12}),
                (8, {"local": True}), (20, {"local": True, "gpu": True})]
# the shapes of rounds 3 / 4 (what `table_kernel` generations 6 / 7 take): Services + preferred self anti-affinity + hard zone constraints +
# required hostname anti-affinity + tolerations / node selectors (synth.typical_cluster_objects), zones round robin
TYPICAL_SEEDS = [(1, 30, 24, 10), (5, 48, 40, 14)]          # (seed, nodes, workloads, max replicas)


def write_yaml_dir(path, objs):
    os.makedirs(path, exist_ok=True)
    for i, o in enumerate(objs):
        with open(os.path.join(path, f"{i:03d}-{o['kind'].lower()}-{o['metadata']['name']}.yaml"), "w") as f:
            yaml.safe_dump(o, f, default_flow_style=False)


def make_cases(ref, work):
    import randk8s
    cases = []
    ex = os.path.join(ref, "example")
    if os.path.isdir(ex):
        for name, cl, apps, nn, k in EXAMPLE_CASES:
            if not os.path.isdir(os.path.join(ex, cl)) or (nn and not os.path.isdir(os.path.join(ex, nn))):
                continue
            cases.append({"name": name, "cluster": os.path.join(ex, cl), "apps": [{"name": a, "path": os.path.join(ex, p)} for a, p in apps],
                          "new_node": os.path.join(ex, nn) if nn else "", "new_nodes": k})
    for seed, kw in RANDOM_SEEDS:
        nodes, workloads, services = randk8s.rand_cluster(seed, **kw)
        base = os.path.join(work, "cases", f"random_{seed}")
        write_yaml_dir(os.path.join(base, "cluster"), nodes + services)
        write_yaml_dir(os.path.join(base, "app"), workloads + (randk8s.STORAGE_CLASSES if kw.get("local") else []))
        cases.append({"name": f"random_{seed}", "cluster": os.path.join(base, "cluster"), "apps": [{"name": "app", "path": os.path.join(base, "app")}],
                      "new_node": "", "new_nodes": 0})
    for name, objs in extra_cases():
        base = os.path.join(work, "cases", name)
        write_yaml_dir(os.path.join(base, "cluster"), objs["cluster"])
        write_yaml_dir(os.path.join(base, "app"), objs["app"])
        cases.append({"name": name, "cluster": os.path.join(base, "cluster"), "apps": [{"name": "app", "path": os.path.join(base, "app")}],
                      "new_node": "", "new_nodes": 0})
    os.makedirs(work, exist_ok=True)
    with open(os.path.join(work, "cases.json"), "w") as f:
        json.dump({"cases": cases}, f, indent=1)
    return cases


def extra_cases():
    """Generated cases beyond tests/randk8s.py: [(name, {"cluster": [objects], "app": [objects]})]."""
    import randk8s
    from open_simulator_amd import synth
    out = []
    for seed, n_nodes, n_workloads, max_rep in TYPICAL_SEEDS:
        nodes, workloads, services = synth.typical_cluster_objects(seed, n_nodes, n_workloads, max_rep)
        for j, n in enumerate(nodes):
            if j % 7 != 6:                                             # every seventh node carries no zone label
                n["metadata"]["labels"][k8s.LABEL_ZONE] = f"z{j % 3}"
        out.append((f"typical_{seed}", {"cluster": nodes + services, "app": workloads}))
    # pods that ARRIVE with a gpu-index annotation (open-gpu-share.go:51-81: the filter ignores idle memory, Reserve books the listed
    # devices): bare pods of a GPU cluster, some naming a device that is too full by then, one with an invalid list
    nodes, workloads, services = randk8s.rand_cluster(31, n_nodes=8, n_workloads=6, gpu=True)
    gpu_nodes = [n for n in nodes if "alibabacloud.com/gpu-count" in n["status"]["allocatable"]]
    bare = []
    for i, idx in enumerate(["0", "1", "0-1", "0", "00-1", "x", "1"]):
        bare.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": f"arrive-{i}", "namespace": "default",
                                                                    "annotations": {"alibabacloud.com/gpu-mem": f"{[4, 8, 6, 12, 2, 4, 8][i]}Gi", "alibabacloud.com/gpu-count": str(len(idx.split("-"))) if idx != "x" else "1",
                                                                                    "alibabacloud.com/gpu-index": idx}},
                     "spec": {"containers": [{"name": "c", "image": "busybox", "resources": {"requests": {"cpu": "500m", "memory": "1Gi"}}}]}})
    if gpu_nodes:
        out.append(("gpu_index_arriving", {"cluster": nodes + services, "app": bare + workloads}))
    return out


# ---- the mirror's side of one case ------------------------------------------------------------------------------------
def load_case(case):
    cluster = k8s.group_resources(k8s.load_objects(case["cluster"]))
    sim.attach_local_storage(cluster["Node"], case["cluster"])
    apps = [sim.AppResource(a["name"], k8s.group_resources(k8s.load_objects(a["path"]))) for a in case["apps"]]
    new = []
    if case.get("new_nodes", 0) > 0:
        tmpl = k8s.group_resources(k8s.load_objects(case["new_node"]))["Node"]
        sim.attach_local_storage(tmpl, case["new_node"])
        new = wl.new_fake_nodes(tmpl[0], case["new_nodes"])
    return cluster, apps, new


def pod_key(app, kind, wns, wname, pin, ns, name):
    """Identity of a pod up to what its workload's replicas share: the app, the owning workload and (DaemonSet pods) the node it is
    pinned to; bare pods by their own name."""
    if kind:
        return (app or "", kind, wns or "", wname, pin or "")
    return (app or "", "Pod", ns or "", name, "")


def mirror_pod_key(p):
    md = p["metadata"]
    a = md.get("annotations") or {}
    return pod_key((md.get("labels") or {}).get(wl.LABEL_APP_NAME, ""), a.get(wl.ANNO_WORKLOAD_KIND, ""), a.get(wl.ANNO_WORKLOAD_NAMESPACE, ""),
                   a.get(wl.ANNO_WORKLOAD_NAME, ""), p.get("_daemon_node", ""), md.get("namespace") or "default", md["name"])


def reorder_like_trace(pods, trace, rename):
    """The mirror's pods in the order of the reference's trace: entry by entry, the next unused mirror pod with the entry's key."""
    buckets = {}
    for i, p in enumerate(pods):
        buckets.setdefault(mirror_pod_key(p), []).append(i)
    order = []
    for e in trace:
        key = pod_key(e["app"], e["workload_kind"], e["workload_namespace"], e["workload_name"], rename.get(e["pin_node"], e["pin_node"]),
                      e["namespace"], e["name"])
        if not buckets.get(key):
            raise ValueError(f"the reference scheduled a pod the mirror does not generate: {key}")
        order.append(buckets[key].pop(0))
    left = [k for k, v in buckets.items() if v]
    if left:
        raise ValueError(f"the mirror generates pods the reference did not schedule: {left[:4]}")
    return order


def strip_pod_name(reason):
    """FitError text without the '(ns/name)' part: the reference appends random suffixes to pod names."""
    i = reason.find("): ")
    return reason[i + 3:] if i >= 0 else reason


def compare_case(case, ref_res, engine):
    if ref_res.get("error"):
        return {"name": case["name"], "status": "ref-error", "detail": ref_res["error"]}
    cluster, apps, new = load_case(case)
    nodes = list(cluster["Node"]) + new
    mirror_names = [n["metadata"]["name"] for n in nodes]
    if len(mirror_names) != len(ref_res["nodes"]):
        return {"name": case["name"], "status": "differ", "detail": f"node count {len(mirror_names)} vs {len(ref_res['nodes'])}"}
    rename = dict(zip(ref_res["nodes"], mirror_names))             # the clones' names are random in the reference: by position
    pods, _ = sim.build_stream(cluster, apps, nodes, len(nodes))
    trace = ref_res["trace"]
    order = reorder_like_trace(pods, trace, rename)
    # does the go1.18 sort.Sort restatement give the order the reference used?  (informational; the comparison uses the trace's)
    same_order = [mirror_pod_key(pods[i]) for i in order] == [mirror_pod_key(p) for p in pods]
    pods = [pods[i] for i in order]
    arrival = list(mirror_names)
    canon = k8s.canonical_node_order(nodes)
    nodes_c = [nodes[j] for j in canon]
    flat = fl.flatten(nodes_c, pods, cluster.get("Service", []), cluster.get("ReplicaSet", []), cluster.get("StatefulSet", []),
                      storage_classes=sim._storage_classes(cluster, apps), image_total=len(nodes_c), node_arrival_order=arrival)
    P = len(pods)
    scen = np.array([[len(nodes_c), 0]], np.int32)
    orders = np.arange(P, dtype=np.int32)[None, :]
    out = engine.run(flat.problem, scen, orders)
    reasons = {}
    if out.unscheduled[0] > 0:
        _, failed, codes, detail = engine.explain(flat.problem, len(nodes_c), orders[0], int(out.unscheduled[0]))
        for i, (pid, row) in enumerate(zip(failed.tolist(), codes)):
            ns, name = flat.pod_refs[pid]
            reasons[pid] = fiterror.unscheduled_reason(ns, name, row, node_names=flat.node_names, static_reasons=flat.static_reasons,
                                                       scalar_names=flat.scalar_names, local_detail=None if detail is None else detail[i],
                                                       vg_names=flat.info.get("vg_names", ()))
    diffs = []
    for i, e in enumerate(trace):
        j = int(out.placement[0][i])
        mine = "" if j < 0 else flat.node_names[j]
        theirs = rename.get(e["node"], e["node"])
        if mine != theirs:
            diffs.append({"slot": i, "pod": f"{e['namespace']}/{e['name']}", "reference": theirs, "oracle": mine})
        elif not theirs and e.get("reason"):
            want = strip_pod_name(e["reason"])
            for a, b in rename.items():                              # Open-Gpu-Share reasons name nodes ("Node:<name>")
                want = want.replace(a, b)
            got = strip_pod_name(reasons.get(i, ""))
            if want != got:
                diffs.append({"slot": i, "pod": f"{e['namespace']}/{e['name']}", "reference_reason": want, "oracle_reason": got})
    return {"name": case["name"], "status": "identical" if not diffs else "differ", "pods": P, "unscheduled": int(out.unscheduled[0]),
            "gosort_matches_reference_order": same_order, "differences": diffs[:10], "n_differences": len(diffs)}


class OracleEngine:
    """The C oracle behind the engine interface of simulate.py."""

    def run(self, prob, scen, orders, want_placement=True, node_ranks=None, want_gpu_slices=False):
        import oracle_lib
        return oracle_lib.run(prob, scen, orders, want_placement, node_ranks=node_ranks, want_gpu_slices=want_gpu_slices)

    def explain(self, prob, n_nodes, order, max_failed):
        import oracle_lib
        _, (nf, failed, codes) = oracle_lib.run(prob, [[n_nodes, 0]], np.asarray(order)[None], explain_scenario=0, max_failed=max_failed)
        return nf, failed, codes, oracle_lib.LAST_LOCAL_DETAIL[0]


class BothEngines:
    """--engine both: every case through the C oracle AND the HIP library (a GPU box), which must agree before either is compared with
    the reference -- so a difference is attributed to the right side."""

    def __init__(self):
        self.oracle, self.hip = OracleEngine(), sim.HipEngine()

    def run(self, prob, scen, orders, want_placement=True, node_ranks=None, want_gpu_slices=False):
        a = self.oracle.run(prob, scen, orders, want_placement, node_ranks, want_gpu_slices)
        b = self.hip.run(prob, scen, orders, want_placement, node_ranks=node_ranks, want_gpu_slices=want_gpu_slices)
        if not ((a.placement == b.placement).all() and a.unscheduled.tolist() == b.unscheduled.tolist()):
            raise AssertionError("the HIP engine and the C oracle disagree on this case (before any comparison with the reference)")
        return b

    def explain(self, prob, n_nodes, order, max_failed):
        a, b = self.oracle.explain(prob, n_nodes, order, max_failed), self.hip.explain(prob, n_nodes, order, max_failed)
        same = a[0] == b[0] and a[1].tolist() == b[1].tolist() and (a[2] == b[2]).all() and \
            ((a[3] is None and b[3] is None) or (a[3] == b[3]).all())
        if not same:
            raise AssertionError("the HIP engine and the C oracle disagree on the failure codes of this case")
        return b


def make_engine(kind):
    return {"oracle": OracleEngine, "hip": sim.HipEngine, "both": BothEngines}[kind]()


def compare(work, ref_out, engine=None, verbose=True):
    with open(os.path.join(work, "cases.json")) as f:
        cases = {c["name"]: c for c in json.load(f)["cases"]}
    with open(ref_out) as f:
        ref = json.load(f)
    engine = engine or OracleEngine()
    report = {"go_version": ref.get("go_version"), "cases": []}
    for r in ref["results"]:
        if r["name"] not in cases:
            continue
        try:
            report["cases"].append(compare_case(cases[r["name"]], r, engine))
        except Exception as e:                                       # noqa: BLE001 -- a case that cannot be compared is a difference
            report["cases"].append({"name": r["name"], "status": "differ", "detail": repr(e)})
    bad = [c for c in report["cases"] if c["status"] != "identical"]
    report["identical"] = len(report["cases"]) - len(bad)
    report["different"] = len(bad)
    with open(os.path.join(work, "parity_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    if verbose:
        for c in report["cases"]:
            print(f"  {c['name']:28s} {c['status']:10s} pods {c.get('pods', '-'):>5} unscheduled {c.get('unscheduled', '-'):>4} "
                  f"gosort-order {'=' if c.get('gosort_matches_reference_order') else '!='} reference  {c.get('detail', '')}")
            for d in c.get("differences", [])[:3]:
                print("      ", d)
        print(f"PARITY vs determinised reference ({report['go_version']}): {report['identical']} identical, {report['different']} different")
    return 0 if not bad and report["cases"] else 1


# ---- self-test: a manufactured reference output ---------------------------------------------------------------------------
def selftest(work):
    """The comparison machinery without Go: `ref_out.json` comes from the object-level restatement tests/pyref_sched.py, with the
    pods shuffled inside the freedom sort.Sort has and the clones renamed as utils.NewFakeNodes would."""
    import pyref_sched
    rng = np.random.default_rng(7)
    # the generated cases of the real manifest (no example/ inputs: they need the reference tree), a few of each kind
    picked = ("random_3", "random_15", "random_8", "typical_1", "gpu_index_arriving")
    cases = [c for c in make_cases("/nonexistent", work) if c["name"] in picked]
    assert len(cases) == len(picked), [c["name"] for c in cases]
    with open(os.path.join(work, "cases.json"), "w") as f:
        json.dump({"cases": cases}, f)
    results = []
    for case in cases:
        cluster, apps, new = load_case(case)
        nodes = list(cluster["Node"]) + new
        pods, _ = sim.build_stream(cluster, apps, nodes, len(nodes))
        perm = rng.permutation(len(pods)).tolist()                   # ANY order is an order the comparison must follow
        pods = [pods[i] for i in perm]
        canon = [nodes[j] for j in k8s.canonical_node_order(nodes)]
        placed = pyref_sched.Scheduler(canon, cluster.get("Service", []), cluster.get("ReplicaSet", []), cluster.get("StatefulSet", []),
                                       sim._storage_classes(cluster, apps)).run(copy.deepcopy(pods))
        trace = []
        for p, node in zip(pods, placed):
            md, a = p["metadata"], p["metadata"].get("annotations") or {}
            trace.append({"namespace": md.get("namespace") or "default", "name": md["name"] + "-x7k2q", "workload_kind": a.get(wl.ANNO_WORKLOAD_KIND, ""),
                          "workload_name": a.get(wl.ANNO_WORKLOAD_NAME, ""), "workload_namespace": a.get(wl.ANNO_WORKLOAD_NAMESPACE, ""),
                          "app": (md.get("labels") or {}).get(wl.LABEL_APP_NAME, ""), "has_node_selector": p["spec"].get("nodeSelector") is not None,
                          "has_tolerations": p["spec"].get("tolerations") is not None, "preset_node": p["spec"].get("nodeName") or "",
                          "pin_node": p.get("_daemon_node", ""), "node": node or "", "reason": ""})   # (no reason: only the bindings are compared here)
            if not trace[-1]["workload_kind"]:
                trace[-1]["name"] = md["name"]                        # bare pods keep their names
        results.append({"name": case["name"], "nodes": [n["metadata"]["name"] for n in nodes], "new_nodes": [], "trace": trace,
                        "unscheduled": sum(1 for x in placed if not x)})
    good = os.path.join(work, "ref_out.json")
    with open(good, "w") as f:
        json.dump({"go_version": "selftest (tests/pyref_sched.py, not Go)", "results": results}, f)
    rc_good = compare(work, good, verbose=False)
    bad_results = copy.deepcopy(results)
    victim = next(e for e in bad_results[0]["trace"] if e["node"])
    victim["node"] = next(n for n in bad_results[0]["nodes"] if n != victim["node"])
    bad = os.path.join(work, "ref_out_corrupt.json")
    with open(bad, "w") as f:
        json.dump({"go_version": "selftest", "results": bad_results}, f)
    rc_bad = compare(work, bad, verbose=False)
    ok = rc_good == 0 and rc_bad == 1
    print(f"selftest: faithful trace -> rc {rc_good} (want 0), corrupted trace -> rc {rc_bad} (want 1): {'OK' if ok else 'FAILED'}")
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("mode", choices=["cases", "compare", "selftest"])
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--work", default=os.path.join(ROOT, "oracle", "_ref", "work"))
    ap.add_argument("--ref-out", default=None)
    ap.add_argument("--engine", choices=["oracle", "hip", "both"], default="oracle",
                    help="compare: what the reference's output is compared with -- the C oracle (default, no GPU needed), the HIP library "
                         "through simulate.HipEngine, or both (they must agree first)")
    a = ap.parse_args()
    if a.mode == "cases":
        cases = make_cases(a.ref, a.work)
        print(f"{len(cases)} cases -> {os.path.join(a.work, 'cases.json')}")
        return 0
    if a.mode == "selftest":
        return selftest(a.work)
    return compare(a.work, a.ref_out or os.path.join(a.work, "ref_out.json"), engine=make_engine(a.engine))


if __name__ == "__main__":
    sys.exit(main())
