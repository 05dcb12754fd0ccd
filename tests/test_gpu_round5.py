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



def test_group_all_gather_failure_aborts_the_communicators_and_reduces_on_the_host(monkeypatch):
    """First-run safety of the device group: a collective that fails AFTER it was enqueued must not leave a stream
    waiting for a peer -- the communicators are aborted, this call and every later one reduce the members' plans on the host, the answer
    is the same.  SIMON_GROUP_FAULT_INJECT makes the next all-gather count as failed once it is in flight (single member, RCCL forced on)."""
    monkeypatch.setenv("SIMON_GROUP_RCCL", "1")
    monkeypatch.setenv("SIMON_GROUP_FAULT_INJECT", "1")
    prob, scen, orders = synth.config3(n_counts=12, n_orders=2, n_pods=500, n_het=40)
    ref = O.run(prob, scen, orders)
    rp = O.min_plan(prob, scen, ref)
    with capi.Group([0]) as grp:
        grp.load_problem(prob)
        grp.run_batch(scen, orders)
        plan, _ = grp.min_plan()
        assert grp.collective() == "host" and plan.as_dict() == rp.as_dict()            # the injected failure: aborted, reduced on the host
        plan2, _ = grp.min_plan()
        assert grp.collective() == "host" and plan2.as_dict() == rp.as_dict()           # ... and it stays there
    monkeypatch.delenv("SIMON_GROUP_FAULT_INJECT")
    with capi.Group([0]) as grp:                                                          # without the fault: the collective itself
        grp.load_problem(prob)
        grp.run_batch(scen, orders)
        plan3, _ = grp.min_plan()
        assert grp.collective() == "rccl_all_gather" and plan3.as_dict() == rp.as_dict()


# ---- generation 4 with the scenario's workspace in LDS (small batches of small problems) -------------------------------------------------
import randprob
from test_gpu_parity import NARROW_FEATURES, assert_same

WS_HOMES = ("0", "1")                            # SIMON_LDS_WS: the workspace in HBM / in LDS whenever it fits


def run_ws(prob, scen, orders, home, monkeypatch):
    monkeypatch.setenv("SIMON_LDS_WS", home)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(True)
        return ctx.fetch(True), ctx.stats()


@pytest.mark.parametrize("idx", range(len(NARROW_FEATURES)))
def test_workspace_in_lds_matches_the_oracle(idx, monkeypatch):
    """Every cpu+memory feature set of the parity suite with the scenario's workspace (byte table, node state, NonZero rows) in HBM and in
    LDS, against the oracle: static masks, presets, gates, pinned pods, initial state, NonZero != request, zero requests, tight pod
    counts, gcd-1 units.  Sizes: 1 / 2 / 4 summary entries per lane, <= 64 and > 64 signatures; the last one does not fit 159 KB and must
    run from HBM whatever the variable says."""
    feat = NARROW_FEATURES[idx]
    used_lds = 0
    for seed in range(4):
        N = [29, 700, 1800, 3900][seed]
        prob = randprob.rand_problem(53000 + 100 * idx + seed, N=N, P=400 + 250 * seed, n_pod_classes=[5, 30, 70, 60][seed], n_node_classes=2 + seed, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=5, min_n=1 if seed == 0 else None)
        ref = O.run_threaded(prob, scen, orders)
        lds = {}
        for home in WS_HOMES:
            res, st = run_ws(prob, scen, orders, home, monkeypatch)
            assert_same(res, ref)
            lds[home] = (st.kernel_variant, st.kernel_generation, st.lds_bytes)
        if lds["0"][0] == capi.KERNEL_NARROW_CACHE and lds["0"][1] == 4 and lds["1"][2] > lds["0"][2]:
            used_lds += 1
            assert lds["1"][1] == 4 and lds["1"][2] <= 159 * 1024
    if not set(feat) & {"odd_units", "nz_differs"}:                       # (those may leave the score table or its one-level layout)
        assert used_lds >= 2, (feat, used_lds)


def test_the_library_puts_the_workspace_in_lds_for_resident_batches_only(monkeypatch):
    """A batch whose workgroups are all resident at once with table + state + summaries in LDS (64 scenarios of a 1 000-node pool: one per
    CU, 63 KB each): LDS.  800 scenarios of the same pool (four per CU would need 250 KB): HBM, as the 4 096-scenario benchmark.
    SIMON_LDS_WS=0 / 1 force either.  Same placements everywhere."""
    monkeypatch.delenv("SIMON_LDS_WS", raising=False)
    prob, scen, orders = synth.config3(n_counts=200, n_orders=4, n_pods=400, n_het=900)
    assert len(scen) == 800
    ref = O.run_threaded(prob, scen, orders)
    seen = {}
    for n_scen in (64, 800):
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen[:n_scen], orders)
            seen[n_scen] = ctx.stats().lds_bytes
        assert (res.placement == ref.placement[:n_scen]).all() and res.unscheduled.tolist() == ref.unscheduled[:n_scen].tolist()
    assert seen[64] > 40 * 1024 > seen[800] > 0, seen


def test_workspace_in_lds_with_per_scenario_node_ranks(monkeypatch):
    """A sweep over a cluster with several zones gives every cluster size its own nodeTree order (simon_set_node_ranks): the LDS home of the
    workspace has its own instantiations for that; a homogeneous pool makes every cycle a tie that the ranks decide."""
    prob, scen, orders = synth.config3(n_counts=6, n_orders=2, n_pods=900, n_het=40, homogeneous=True)
    rng = np.random.default_rng(5)
    N = prob.n_nodes
    ranks = np.zeros((len(scen), N), np.int32)
    for i, (n, _) in enumerate(scen.tolist()):
        ranks[i, :n] = rng.permutation(n)
        ranks[i, n:] = np.arange(n, N)
    ref = O.run(prob, scen, orders, node_ranks=ranks)
    lds = {}
    for home in WS_HOMES:
        monkeypatch.setenv("SIMON_LDS_WS", home)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            res, st = ctx.fetch(True), ctx.stats()
        assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 4
        assert_same(res, ref)
        lds[home] = st.lds_bytes
    assert lds["1"] > lds["0"]


def test_config3_sixty_four_scenarios_with_the_workspace_in_lds(monkeypatch):
    """BASELINE config 3's pool at the batch a real `simon apply` offers (16 node counts x 4 orders = 64 scenarios, 10 000 pods each) and
    BASELINE config 2 (one scenario): both homes of the workspace against the oracle, every placement."""
    for prob, scen, orders in (synth.config3(n_counts=16), synth.config2()):
        ref = O.run_threaded(prob, scen, orders)
        for home in WS_HOMES:
            res, st = run_ws(prob, scen, orders, home, monkeypatch)
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 4
            assert_same(res, ref)


# ---- generation 6 with its position-mask rows in LDS (small gpushare sweeps) ---------------------------------------------------------------
from test_gpu_parity import REST_FEATURES


@pytest.mark.parametrize("idx", range(len(REST_FEATURES)))
def test_mask_rows_in_lds_match_the_oracle(idx, monkeypatch):
    """Generation 6 (Open-Gpu-Share devices, required (anti-)affinity incl. zone-like keys and the first-pod escape, host ports, ephemeral
    storage / extended resources as position masks) with the mask rows, their totals and the canonical indices in HBM and in LDS, against
    the oracle -- placements and the devices Reserve books."""
    feat = REST_FEATURES[idx]
    monkeypatch.setenv("SIMON_NO_FOLD", "1")
    monkeypatch.setenv("SIMON_NO_GPU_FOLD", "1")
    for seed in range(4):
        N = [37, 150, 700, 1500][seed]
        prob = randprob.rand_problem(54000 + 100 * idx + seed, N=N, P=500 + 300 * seed, n_pod_classes=6 + 5 * seed + (60 if seed == 3 else 0), n_node_classes=3 + 2 * seed, **feat)
        scen, orders = randprob.rand_scenarios(seed, prob, S=5, min_n=1 if seed == 0 else None)
        gpu = bool(feat.get("gpu"))
        ref = O.run_threaded(prob, scen, orders, want_gpu_slices=gpu)
        lds = {}
        for home in WS_HOMES:
            monkeypatch.setenv("SIMON_LDS_WS", home)
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                res = ctx.run_batch(scen, orders, want_gpu_slices=gpu)
                st = ctx.stats()
            assert_same(res, ref)
            if gpu:
                assert (res.gpu_slices == ref.gpu_slices).all()
            lds[home] = (st.kernel_variant, st.kernel_generation, st.lds_bytes)
        if lds["0"][0] == capi.KERNEL_NARROW_CACHE and lds["0"][1] == 6:
            # the rows did move to LDS -- with at most 128 signatures (seed 3 draws 81 pod classes: beyond 128 signatures generation 6 runs the
            # signature groups of MANY since the end of round 6, rows in HBM, where it used to leave the table)
            assert lds["1"][1] == 6 and (lds["1"][2] > lds["0"][2] or (seed == 3 and lds["1"][2] == lds["0"][2])), lds


def test_config5_sixty_four_full_size_scenarios_with_the_mask_rows_in_lds(monkeypatch):
    """BASELINE config 5 at full size (50 000 pods x 2 500 .. 5 000 nodes; GPU share + self anti-affinity groups + taints): 64 of the 256
    benchmarked scenarios with the mask rows in LDS (what a 256-scenario batch gets by default: one scenario per CU), every placement."""
    monkeypatch.setenv("SIMON_LDS_WS", "1")
    prob, scen, orders = synth.config5()
    pick = np.unique(np.linspace(0, len(scen) - 1, 64).astype(int))
    ref = O.run_threaded(prob, scen[pick], orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen[pick], orders)
        st = ctx.stats()
    assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 6 and st.lds_bytes > 64 * 1024, (st.kernel_generation, st.lds_bytes)
    assert_same(res, ref)
