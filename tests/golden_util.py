"""Load tests/golden/*.json back into Problem objects (tests only)."""
import hashlib
import json
import os

import numpy as np

from open_simulator_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = ["kav_two_nodes", "example_gpushare", "core_test_simple"]


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        d = json.load(f)
    pd = dict(d["problem"])
    kw = {k: (np.asarray(v, np.uint64) if k == "static_mask" else np.asarray(v)) for k, v in pd.items()
          if k not in ("n_pod_classes", "n_node_classes")}
    prob = capi.Problem(n_pod_classes=pd["n_pod_classes"], n_node_classes=pd["n_node_classes"], **kw).normalise()
    scen = np.asarray(d["scenarios"], np.int32)
    orders = np.asarray(d["orders"], np.int32)
    return d, prob, scen, orders


def digests():
    with open(os.path.join(GOLDEN, "synthetic_digests.json")) as f:
        return json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_against(d, res):
    o = d["oracle"]
    assert res.unscheduled.tolist() == o["unscheduled"]
    assert res.used_cpu.tolist() == o["used_cpu"] and res.used_mem.tolist() == o["used_mem"]
    assert res.placement.tolist() == o["placement"]
