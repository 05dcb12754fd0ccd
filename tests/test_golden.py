"""CPU: the oracle against the committed golden fixtures and the reference expectations they were pinned to."""
import collections

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O
from open_simulator_amd import synth


@pytest.mark.parametrize("name", G.SMALL)
def test_oracle_reproduces_golden(name):
    d, prob, scen, orders = G.load(name)
    G.check_against(d, O.run(prob, scen, orders))


def test_reference_expectations_hold_in_golden():
    d, *_ = G.load("kav_two_nodes")
    assert d["oracle"]["placement"] == [[0, 0]]                       # pod -> A, second pod -> A again
    d, *_ = G.load("example_gpushare")
    assert d["oracle"]["unscheduled"] == [0]                          # example/simon-gpushare-config.yaml fits 2 nodes
    d, prob, *_ = G.load("core_test_simple")
    assert d["oracle"]["unscheduled"] == [0]                          # core_test.go:346 failedPodsNum == 0
    pl = d["oracle"]["placement"][0]
    counts = collections.Counter(n.rsplit("-", 1)[0] if n.rsplit("-", 1)[-1].isdigit() else n
                                 for n, p in zip(d["pod_names"], pl) if p >= 0)
    for workload, want in d["expected_counts"].items():               # checkResult, core_test.go:364-591
        assert counts[workload] == want, workload
    names = d["pod_names"]
    assert pl[names.index("metrics-server")] in (1, 2)                # node affinity master, master-1 taint not tolerated
    assert pl[names.index("busybox-ds")] == 3 and pl[names.index("pi")] != 0
    assert pl[:4] == [0, 0, 0, 0]                                     # static pods keep their Spec.NodeName


def test_full_size_digests():
    dg = G.digests()
    for hom in (False, True):
        prob, scen, orders = synth.config2(hom)
        r = O.run(prob, scen, orders)
        want = dg[f"config2_{'homogeneous' if hom else 'heterogeneous'}"]
        assert r.unscheduled.tolist() == want["unscheduled"] and G.sha(r.placement) == want["placement_sha256"]
    prob, scen, orders = synth.config3()
    want = dg["config3_subset"]
    assert scen[want["pick"]].tolist() == want["scenarios"]
    r = O.run(prob, scen[want["pick"][:3]], orders)                   # 3 of the 12 on CPU (the GPU test checks all)
    assert r.unscheduled.tolist() == want["unscheduled"][:3]
    assert [G.sha(row) for row in r.placement] == want["placement_sha256"][:3]
