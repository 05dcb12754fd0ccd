"""GPU parity tests, round 5: BASELINE config 4 as a whole through the device group (8 members), team shapes of generations 4 - 6
(several waves per scenario for the batches a real `simon apply` offers), and the regressions of the round-4 review."""
import os

import numpy as np
import pytest

import oracle_lib as O
from open_simulator_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_config4_whole_batch_through_a_group_of_eight_members():
    """BASELINE config 4: 10 000 pods x 488..1 511 nodes, 1 024 node counts x 32 pod orders = 32 768 scenarios dealt over 8 members
    (scenario s -> member s % 8, as 8 GPUs of one node would take them; here the 8 members share device 0, so the minimum-node plan is
    reduced on the host -- the same rule the RCCL all-gather's reader applies).  ALL 32 768 scenarios run; checked against the oracle:
      * >= 512 scenarios placement by placement: every order at the node counts around the plan's, plus a spread over the grid;
      * the global plan: the oracle's plan rule over the per-scenario results equals simon_group_min_plan's answer, and the
        winning scenario itself and all 32 scenarios one node count below it are among the oracle-checked ones."""
    prob, scen, orders = synth.config3(n_orders=32, seed=synth.SEED + 4)
    assert len(scen) == 32768
    with capi.Group([0] * 8) as grp:
        assert grp.size == 8
        grp.load_problem(prob)
        grp.load_scenarios(scen, orders)
        grp.run_loaded(want_placement=True)
        res = grp.fetch(want_placement=False)
        plan, _ = grp.min_plan()
        assert grp.collective() == "host"
        for i in range(8):
            st = grp.member_stats(i)
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE, (i, st.kernel_variant)
        assert plan.found
        # the plan rule (apply.go:203-259 + satisfyResourceSetting) over ALL per-scenario results, by the oracle's implementation
        ref_plan = O.min_plan(prob, scen, res)
        assert plan.as_dict() == ref_plan.as_dict()
        s_win = plan.scenario
        n_win = int(scen[s_win, 0])
        assert n_win == plan.n_nodes
        # sample: all 32 orders at the winning count and the three counts around it, plus 480 spread over the grid
        counts = sorted(set(scen[:, 0].tolist()))
        ci = counts.index(n_win)
        near = [c for c in (ci - 2, ci - 1, ci, ci + 1, ci + 2, ci + 3) if 0 <= c < len(counts)][:4]
        pick, by_count = set(), {}
        for s in range(len(scen)):
            by_count.setdefault(int(scen[s, 0]), []).append(s)
        for c in near:
            pick.update(by_count[counts[c]])
        pick.update(np.linspace(0, len(scen) - 1, 480).astype(int).tolist())
        pick.add(s_win)
        pick = np.array(sorted(pick))
        assert len(pick) >= 512 and set(scen[pick, 1].tolist()) == set(range(32))
        ref = O.run_threaded(prob, scen[pick], orders)
        assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist()
        assert res.used_cpu[pick].tolist() == ref.used_cpu.tolist()
        assert res.used_mem[pick].tolist() == ref.used_mem.tolist()
        for j, s in enumerate(pick.tolist()):
            assert (grp.fetch_placement(s) == ref.placement[j]).all(), s
        # minimality at the edge: no order fits one node count below the plan's (all 32 of them are in the oracle-checked sample)
        if ci > 0:
            assert all(int(res.unscheduled[s]) > 0 for s in by_count[counts[ci - 1]])

