"""GPU parity tests, round 6 (ABI v6): pods of unequal priority (the DefaultPreemption-risk flag), ScalarResources ENTRIES of quantity 0,
and the kernels touched this round (generation 6 with its feasible-node counters in LDS; global address space named for every table)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, synth

pytestmark = pytest.mark.gpu

GiB = 1 << 30


def run_gpu(prob, scen, orders, env=None, explain=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            res = ctx.run_batch(scen, orders, True)
            res.preempt_risk = ctx.fetch_preempt_risk()
            st = ctx.stats()
            ex = ctx.explain(*explain) if explain else None
            return res, st, ex
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def assert_same(a, b):
    assert a.unscheduled.tolist() == b.unscheduled.tolist()
    assert a.used_cpu.tolist() == b.used_cpu.tolist() and a.used_mem.tolist() == b.used_mem.tolist()
    assert (a.placement == b.placement).all()
    assert a.preempt_risk.tolist() == b.preempt_risk.tolist()


PRIO_FEATURES = [dict(), dict(presets=True, gates=True, tight_pods=True), dict(gpu=True), dict(anti=True, static_mask=True),
                 dict(spread_soft=True), dict(eph=True, scalars=2, init_state=True)]


@pytest.mark.parametrize("idx", range(len(PRIO_FEATURES)))
def test_preempt_risk_flag_matches_the_oracle_on_every_kernel(idx):
    """simon_set_pod_priorities / simon_fetch_preempt_risk: per scenario, "some pod failed while a pod of lower priority was placed" --
    what DefaultPreemption needs to act (V/scheduler.go:479, default_preemption.go:578-592).  The flag is computed on the device from the
    placement matrix of whatever kernel ran; the oracle keeps a running minimum inside its scheduling loop.  Tight clusters (many failures),
    preset pods (they count as victims), gated pods (they do not exist), every kernel family, with and without the placement matrix asked for."""
    feat = PRIO_FEATURES[idx]
    seen = set()
    for seed in range(4):
        prob = randprob.rand_problem(6000 + 100 * idx + seed, N=30 + 17 * seed, P=500, **feat)
        rng = np.random.default_rng(seed)
        prob.priority = rng.choice([0, 0, 0, 10, 1000, -5], prob.n_pods).astype(np.int32)
        prob.init_min_priority = [0x7fffffff, 0, 100, 2000][seed]
        scen, orders = randprob.rand_scenarios(seed, prob, S=10, min_n=4)
        ref = O.run(prob, scen, orders)
        for env in ({}, {"SIMON_FORCE_WIDE": "1"}, {"SIMON_NO_CACHE": "1"}):
            res, st, _ = run_gpu(prob, scen, orders, env=env)
            assert_same(res, ref)
            seen.add((st.kernel_variant, st.kernel_generation))
        assert ref.preempt_risk.any() or ref.unscheduled.max() == 0 or seed > 0
        with capi.Context(0) as ctx:                                   # want_placement = 0: the library stores the matrix all the same
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.run_loaded(False)
            assert ctx.fetch_preempt_risk().tolist() == ref.preempt_risk.tolist()
        with capi.Group([0, 0, 0]) as grp:                             # scenario s on member s % 3: flags come back in global order
            grp.load_problem(prob)
            grp.load_scenarios(scen, orders)
            grp.run_loaded(True)
            assert grp.fetch_preempt_risk().tolist() == ref.preempt_risk.tolist()
    assert len(seen) >= 2 or idx == 5          # (extra resources over an occupied pool: the all-feature kernel whatever the switch)
    # no priorities loaded: no flag anywhere, and loading pods again clears what was set
    prob.priority = None
    res, _, _ = run_gpu(prob, scen, orders)
    assert not res.preempt_risk.any()


def test_preempt_risk_at_full_size_on_the_benchmarked_batch():
    """Size-independent property on BASELINE config 3's batch (4 096 scenarios x 10 000 pods): with priorities that DEcrease along every pod
    order no pod can ever fail above a placed one -- no flag -- and with one low-priority pod first every scenario with a failure is flagged."""
    prob, scen, orders = synth.config3(n_counts=256, n_orders=2)
    scen = scen[scen[:, 1] == 1]                                       # the descending-dominant-share order: the small sizes leave pods out
    P = prob.n_pods
    prio = np.zeros(P, np.int32)
    prio[orders[1]] = np.arange(P, 0, -1)
    prob.priority = prio
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders, False)
        assert res.unscheduled.max() > 0 and not ctx.fetch_preempt_risk().any()
        prio2 = np.full(P, 5, np.int32)
        prio2[orders[1][0]] = 1
        prob.priority = prio2
        ctx.load_problem(prob)
        res = ctx.run_batch(scen, orders, False)
        risk = ctx.fetch_preempt_risk()
        assert risk.tolist() == (res.unscheduled > 0).astype(np.uint8).tolist() and risk.any() and not risk.all()


def test_zero_quantity_scalar_entries_on_every_kernel():
    """simon_set_scalar_entries: fit.go:244-249 -- len(ScalarResources) == 0, not "all quantities zero" -- and :275-299 -- every ENTRY is
    compared.  A node that bound pods over-committed (init state) rejects an all-zero pod that holds an entry ("Insufficient cpu" /
    "Insufficient memory") and admits the same pod without one; an entry for a resource the node is over-committed ON fails with that
    resource's bit.  Score-table kernel (cpu + memory: the entry only disables the shortcut), generations 1 / 2, the all-feature kernel."""
    for seed in range(4):
        prob = randprob.rand_problem(7000 + seed, N=40, P=300, zero_pods=True, nz_differs=True, init_state=True)
        rng = np.random.default_rng(seed)
        over = rng.random(prob.n_nodes) < 0.4                          # over-committed at the start: Requested > Allocatable
        prob.init_req_cpu = np.where(over, prob.alloc_cpu + 1000, prob.init_req_cpu).astype(np.int64)
        prob.init_nz_cpu = np.maximum(prob.init_nz_cpu, prob.init_req_cpu)
        zero = rng.random(prob.n_pods) < 0.2                           # BestEffort pods: no request at all (the non-zero defaults stay)
        prob.req_cpu = np.where(zero, 0, prob.req_cpu).astype(np.int64)
        prob.req_mem = np.where(zero, 0, prob.req_mem).astype(np.int64)
        ent = np.where(zero & (rng.random(prob.n_pods) < 0.5), 0x80, 0).astype(np.uint8)
        scen, orders = randprob.rand_scenarios(seed, prob, S=6, min_n=6)
        base = O.run(prob, scen, orders)
        prob.scalar_entries = ent
        ref = O.run(prob, scen, orders)
        assert (ref.placement != base.placement).any()                 # the entries matter on this input
        for env in ({}, {"SIMON_FORCE_WIDE": "1"}, {"SIMON_NO_CACHE": "1"}, {"SIMON_NARROW_V1": "1"}, {"SIMON_LDS_WS": "0"}):
            res, st, _ = run_gpu(prob, scen, orders, env=env)
            assert_same(res, ref)
        # tracked resources: quantities for some pods, zero-quantity ENTRIES for others, a pool over-committed on resource 0
        prob2 = randprob.rand_problem(7100 + seed, N=30, P=250, scalars=2, zero_pods=True, init_state=True)
        prob2.init_scalar_req = np.zeros((2, prob2.n_nodes), np.int64)
        prob2.init_scalar_req[0] = np.where(rng.random(prob2.n_nodes) < 0.5, prob2.scalar_alloc[0] + 1, 0)
        ent2 = np.zeros(prob2.n_pods, np.uint8)
        pick = (prob2.scalar_req[0] == 0) & (rng.random(prob2.n_pods) < 0.5)
        ent2[pick] |= 1
        ent2[(prob2.scalar_req[1] == 0) & (rng.random(prob2.n_pods) < 0.3)] |= 2
        scen2, orders2 = randprob.rand_scenarios(seed, prob2, S=5, min_n=6)
        base2 = O.run(prob2, scen2, orders2)
        prob2.scalar_entries = ent2
        nmax = int(scen2[0, 0])
        ref2, (nf, failed, codes) = O.run(prob2, scen2, orders2, explain_scenario=0, max_failed=8)
        assert (ref2.placement != base2.placement).any()
        res2, st2, ex = run_gpu(prob2, scen2, orders2, explain=(nmax, orders2[scen2[0, 1]], 8))
        assert st2.kernel_variant == capi.KERNEL_WIDE                  # an over-committed pool: not the position-mask rows
        assert_same(res2, ref2)
        assert ex[0] == nf and ex[1].tolist() == failed.tolist() and (ex[2] == codes).all()


def test_config5_and_rest_features_with_counters_in_lds():
    """Generation 6 keeps its feasible-node counters per (signature, class) in LDS since round 6 (tcarve: rest): a 64-scenario slice of
    BASELINE config 5 at full size, placement by placement, on both homes of the mask rows."""
    prob, scen, orders = synth.config5(n_scen=64, n_orders=4)
    pick = np.arange(0, 64, 8)
    ref = O.run_threaded(prob, scen[pick], orders)
    for mode in ("0", "1"):
        res, st, _ = run_gpu(prob, scen[pick], orders, env={"SIMON_LDS_WS": mode})
        assert st.kernel_generation == 6
        assert res.unscheduled.tolist() == ref.unscheduled.tolist()
        assert (res.placement == ref.placement).all()


def test_bench_strong_scaling_splits_the_fixed_batch_over_the_ranks(tmp_path):
    """bench.py --gpus 2 --strong: BASELINE config 3's fixed batch (--orders-per-gpu orders in TOTAL) split over the ranks, "scaling": "strong"
    (the day a SCALE run happens both curves exist).  Two ranks on device 0 over gloo, as the weak-scaling twin in
    tests/test_gpu_parity.py."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, SIMON_BENCH_BACKEND="gloo", SIMON_BENCH_SHARE_DEVICE="1", SIMON_BENCH_DETAIL=str(tmp_path / "bench_detail.json"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--strong", "--steps", "2", "--warmup", "1",
           "--counts", "32", "--pods", "2000", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["scenarios_per_gpu"] == 32 * 4 // 2 and d["value"] > 0
    assert "4 pod orders = 128 scenarios" in d["config"]["workload"] and "fixed batch" in d["config"]["workload"]


def test_service_workload_with_30_node_shapes_in_3_zones_stays_on_generation_7():
    """Generation 7 splits the internal node classes by zone, so 30 node shapes in 3 zones are 90 classes: since round 6 the walks of
    soft constraints keep two classes per lane (simon_table.hip: CN2, simon_table_spread2.hip) instead of handing the problem to the
    all-feature kernel.  BASELINE config 3's pool behind Services with the existing nodes in 30 shapes; random problems with 65 .. 128
    classes, NonZeroRequested differing, presets / gates / pins, per-scenario node ranks; SIMON_NO_CN2 = the old route, same answers."""
    prob, scen, orders = synth.config_service(n_counts=24, n_pods=4000, n_shapes=30)
    pick = np.array([0, 5, 37, 70, 95])
    ref = O.run_threaded(prob, scen[pick], orders)
    for team, wg in (("0", 64), ("1", 256)):                           # one wave per scenario; a team of four (what a batch this small gets)
        res, st, _ = run_gpu(prob, scen[pick], orders, env={"SIMON_TEAM": team})
        assert st.kernel_generation == 7 and st.workgroup_size == wg
        assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
        assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()
    res0, st0, _ = run_gpu(prob, scen[pick], orders, env={"SIMON_NO_CN2": "1"})
    assert st0.kernel_variant == capi.KERNEL_WIDE and (res0.placement == ref.placement).all()
    on7 = 0
    for seed, (N, P, ncls, feat) in enumerate([(400, 600, 20, {}), (900, 1200, 25, dict(nz_differs=True, init_state=True)), (2500, 1500, 22, dict(gates=True, presets=True)),
                                               (700, 900, 18, dict(static_small=True, tight_pods=True, zero_pods=True)), (4000, 1200, 24, dict(pins=True, odd_units=True))]):
        p2 = randprob.rand_problem(4242 + seed, N=N, P=P, spread_soft=True, n_node_classes=ncls, n_pod_classes=[3, 30, 60, 8, 100][seed], **feat)
        s2, o2 = randprob.rand_scenarios(seed, p2, S=4, min_n=N // 3)
        ranks = None
        if seed % 2:                                                   # per-scenario node order: ties follow the ranks
            rng = np.random.default_rng(seed)
            ranks = np.zeros((len(s2), p2.n_nodes), np.int32)
            for i, (n, _) in enumerate(np.asarray(s2).tolist()):
                ranks[i, :n] = rng.permutation(n)
        ref2 = O.run(p2, s2, o2, node_ranks=ranks) if ranks is not None else O.run_threaded(p2, s2, o2)
        with capi.Context(0) as ctx:
            ctx.load_problem(p2)
            ctx.load_scenarios(s2, o2)
            if ranks is not None:
                ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            r2 = ctx.fetch(True)
            on7 += ctx.stats().kernel_generation == 7
        assert r2.unscheduled.tolist() == ref2.unscheduled.tolist() and (r2.placement == ref2.placement).all()
        assert r2.used_cpu.tolist() == ref2.used_cpu.tolist() and r2.used_mem.tolist() == ref2.used_mem.tolist()
    assert on7 >= 4


def test_gpushare_cluster_with_80_node_shapes_stays_on_generation_6():
    """Generation 6's REST select keeps two node classes per lane beyond 64 internal classes (simon_table.hip: CN2, simon_table_rest2.hip):
    BASELINE config 5's shape on 80 node shapes -- GPU share, required self anti-affinity, taints -- placement by placement and with the
    booked devices compared; random problems with 70 .. 110 node classes and every REST feature (zone-level terms, required affinity, ports,
    ephemeral storage / extended resources, per-scenario ranks); SIMON_NO_CN2 = the old route (all-feature kernel), same answers."""
    prob, scen, orders = synth.config5(n_pods=6000, n_nodes=900, n_scen=8, n_orders=2, n_groups=20, group_size=30, n_shapes=80)
    ref = O.run_threaded(prob, scen, orders)
    res, st, _ = run_gpu(prob, scen, orders)
    assert st.kernel_generation == 6
    assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
    assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()
    res0, st0, _ = run_gpu(prob, scen, orders, env={"SIMON_NO_CN2": "1"})
    assert st0.kernel_variant == capi.KERNEL_WIDE and (res0.placement == ref.placement).all()
    on6 = 0
    feats = [dict(gpu=True), dict(gpu=True, anti=True, aff=True), dict(anti_host=True, ports=True, nz_differs=True), dict(gpu=True, eph=True, scalars=2, anti_host=True),
             dict(anti=True, presets=True, gates=True, tight_pods=True), dict(gpu=True, anti_host=True, static_small=True, zero_pods=True)]
    for seed, feat in enumerate(feats):
        N = [300, 700, 1500, 500, 2500, 900][seed]
        p2 = randprob.rand_problem(5150 + seed, N=N, P=700, n_node_classes=[70, 90, 110, 80, 100, 120][seed], n_pod_classes=[3, 30, 60, 8, 100, 20][seed], **feat)
        s2, o2 = randprob.rand_scenarios(seed, p2, S=4, min_n=N // 2)
        ranks = None
        if seed % 3 == 2:
            rng = np.random.default_rng(seed)
            ranks = np.zeros((len(s2), p2.n_nodes), np.int32)
            for i, (n, _) in enumerate(np.asarray(s2).tolist()):
                ranks[i, :n] = rng.permutation(n)
        ref2 = O.run(p2, s2, o2, node_ranks=ranks) if ranks is not None else O.run_threaded(p2, s2, o2)
        with capi.Context(0) as ctx:
            ctx.load_problem(p2)
            ctx.load_scenarios(s2, o2)
            if ranks is not None:
                ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            r2 = ctx.fetch(True)
            on6 += ctx.stats().kernel_generation == 6
        assert r2.unscheduled.tolist() == ref2.unscheduled.tolist() and (r2.placement == ref2.placement).all()
        assert r2.used_cpu.tolist() == ref2.used_cpu.tolist() and r2.used_mem.tolist() == ref2.used_mem.tolist()
    assert on6 >= 4


def test_soft_spread_constraints_over_the_position_mask_rows():
    """REST && SPREAD in one instantiation (simon_table_rs.hip): pods behind Services next to Open-Gpu-Share requests,
    required anti-affinity (hostname and zone-like keys), host ports and extra resources that do NOT fold into the table -- the spread walk
    honours the mask rows and counts the admitted nodes of every class itself (podtopologyspread/scoring.go:60-256 runs over the nodes that
    pass ALL filters).  BASELINE config 5 as it is drawn with its anti-affinity groups behind Services (folds switched off: at this size they
    would still take it); random problems per feature; SIMON_NO_RS = the old route, same answers."""
    env = {"SIMON_NO_FOLD": "1", "SIMON_NO_GPU_FOLD": "1"}
    prob, scen, orders = synth.config5(n_pods=6000, n_nodes=900, n_scen=8, n_orders=2, n_groups=20, group_size=30, services=True)
    ref = O.run_threaded(prob, scen, orders)
    res, st, _ = run_gpu(prob, scen, orders, env=env)
    assert st.kernel_generation == 7 and st.kernel_variant != capi.KERNEL_WIDE
    assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
    assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()
    res0, st0, _ = run_gpu(prob, scen, orders, env=dict(env, SIMON_NO_RS="1"))
    assert st0.kernel_variant == capi.KERNEL_WIDE and (res0.placement == ref.placement).all()
    on7 = 0
    feats = [dict(gpu=True), dict(anti_host=True, ports=True), dict(gpu=True, anti_host=True, nz_differs=True, init_state=True), dict(gpu=True, presets=True, gates=True, pins=True),
             dict(anti=True, tight_pods=True, zero_pods=True), dict(gpu=True, anti=True, static_small=True, odd_units=True)]
    for seed, feat in enumerate(feats):
        N = [200, 500, 1400, 300, 2600, 800][seed]
        p2 = randprob.rand_problem(7300 + seed, N=N, P=700, spread_soft=True, n_node_classes=[4, 2, 9, 1, 4, 6][seed], n_pod_classes=[3, 30, 60, 8, 20, 40][seed], **feat)
        s2, o2 = randprob.rand_scenarios(seed, p2, S=4, min_n=max(1, N // 2))
        ranks = None
        if seed % 3 == 1:
            rng = np.random.default_rng(seed)
            ranks = np.zeros((len(s2), p2.n_nodes), np.int32)
            for i, (n, _) in enumerate(np.asarray(s2).tolist()):
                ranks[i, :n] = rng.permutation(n)
        ref2 = O.run(p2, s2, o2, node_ranks=ranks) if ranks is not None else O.run_threaded(p2, s2, o2)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            with capi.Context(0) as ctx:
                ctx.load_problem(p2)
                ctx.load_scenarios(s2, o2)
                if ranks is not None:
                    ctx.set_node_ranks(ranks)
                ctx.run_loaded(True)
                r2 = ctx.fetch(True)
                on7 += ctx.stats().kernel_generation == 7
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert r2.unscheduled.tolist() == ref2.unscheduled.tolist() and (r2.placement == ref2.placement).all()
        assert r2.used_cpu.tolist() == ref2.used_cpu.tolist() and r2.used_mem.tolist() == ref2.used_mem.tolist()
    assert on7 >= 5


def test_ranked_generation_7_without_counter_rows_stays_inside_its_workspace_slice(monkeypatch):
    """Found by fuzz_spread.py case 702439 (round 6): per-scenario node ranks + preferred terms whose weights cancel -- no term keeps a counter row,
    so the scenario's workspace slice has no SPREAD block, and the prologue wrote the per-position ranks behind the slice: into the NEXT scenario's
    byte table (wrong placements that depended on what ran before).  The workspace starts from a byte pattern here so that a stray read or write
    shows whatever ran before."""
    import fuzz_spread
    monkeypatch.setenv("SIMON_WS_FILL", "170")
    for case in (702439, 702439):
        ok, info = fuzz_spread.one_case(case)
        assert ok, info


def test_team_mode_beyond_128_signatures():
    """Round 6: the team of waves takes the signature groups of MANY along (129 .. 1 023 signatures next to soft constraints) -- a slice of
    fuzz_spread.py's cases from 300 000 on, each run on one wave and on the team; at least some must land on generation 7's team shape."""
    import fuzz_spread
    teams = 0
    for case in range(300000, 300024):
        ok, info = fuzz_spread.one_case(case)
        assert ok, info
        teams += info["team"]
    assert teams >= 3, teams


MANY_CLASS_FEATURES = [dict(), dict(static_small=True, tight_pods=True), dict(pins=True, presets=True, gates=True), dict(nz_differs=True, init_state=True, static_mask=True)]


@pytest.mark.parametrize("idx", range(len(MANY_CLASS_FEATURES)))
def test_129_to_256_distinct_node_classes_on_the_score_table(idx):
    """VERDICT r5 next-6 (second half): clusters of more than 128 distinct node shapes stay on the score table -- generation 4's one-level layout (a class segment
    padded to 16 positions), the class terms' re-base walking four groups of 64 classes (simon_table_cls4.hip).  With the static score tables (their maxima run over
    all groups), pins / presets / gates, NonZeroRequested != Requested, and per-scenario node ranks; 257 classes leave the table."""
    feat = MANY_CLASS_FEATURES[idx]
    for seed, (n_classes, N, P) in enumerate([(129, 400, 1200), (160, 700, 2000), (200, 1500, 2500), (256, 1000, 1500), (257, 600, 800)]):
        rng = np.random.default_rng(9600 + 10 * idx + seed)
        prob = randprob.rand_problem(9600 + 10 * idx + seed, N=N, P=P, n_node_classes=n_classes, n_pod_classes=6, **feat)
        cls = np.concatenate([np.arange(n_classes), rng.integers(0, n_classes, N - n_classes)]).astype(np.int32)   # every class occurs
        rng.shuffle(cls)
        prob.node_class = cls
        cpu = 2000 + 500 * (np.arange(n_classes) % 23) + 7000 * (np.arange(n_classes) // 23)                      # a distinct allocatable per class
        prob.alloc_cpu = cpu[cls].astype(np.int64)
        prob.alloc_mem = ((4 + np.arange(n_classes) % 9)[cls].astype(np.int64)) << 30
        if prob.init_req_cpu is not None:
            prob.init_req_cpu = np.minimum(prob.init_req_cpu, prob.alloc_cpu // 2); prob.init_req_mem = np.minimum(prob.init_req_mem, prob.alloc_mem // 2)
            prob.init_nz_cpu = np.maximum(prob.init_nz_cpu, prob.init_req_cpu); prob.init_nz_mem = np.maximum(prob.init_nz_mem, prob.init_req_mem)
        prob.normalise()
        scen, orders = randprob.rand_scenarios(96 + seed, prob, S=5)
        scen[0, 0] = N
        ranks = None
        if seed % 2 == 1:
            ranks = np.zeros((len(scen), prob.n_nodes), np.int32)
            for s_, (n, _) in enumerate(np.asarray(scen).tolist()):
                ranks[s_, :n] = rng.permutation(n)
        ref = O.run(prob, scen, orders, node_ranks=ranks) if ranks is not None else O.run_threaded(prob, scen, orders)
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            if ranks is not None:
                ctx.set_node_ranks(ranks)
            ctx.run_loaded(True)
            res = ctx.fetch(True)
            st = ctx.stats()
        on_table = st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 4
        assert on_table == (n_classes <= 256), (n_classes, st.kernel_variant, st.kernel_generation)
        assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
        assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()


def test_config3_on_160_node_shapes_stays_on_generation_4():
    """bench.py's `config3_classes160` row at a size the oracle finishes: BASELINE config 3's pods on 160 distinct node shapes."""
    prob, scen, orders = synth.config3_classes(160, n_counts=12, n_orders=2, n_pods=4000)
    ref = O.run_threaded(prob, scen, orders)
    res, st, _ = run_gpu(prob, scen, orders)
    assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 4, (st.kernel_variant, st.kernel_generation)
    assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
    assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()


def test_config3_classes160_benchmarked_batch_against_the_oracle():
    """The batch of bench.py's `config3_classes160` row (10 000 pods x 488 .. 1 511 nodes in 160 shapes, 4 096 scenarios, four per CU on the kernels of
    simon_table_cls4.hip): 1 024 evenly spaced scenarios of the FULL batch row by row against the oracle (~10 s on 16 host threads; 256 on a small box)."""
    prob, scen, orders = synth.config3_classes(160)
    pick = np.unique(np.linspace(0, len(scen) - 1, 1024 if O.host_threads() >= 8 else 256).astype(int))
    ref = O.run_threaded(prob, scen[pick], orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(True)
        st = ctx.stats()
        res = ctx.fetch(True)
    assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 4 and len(scen) == 4096
    assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist()
    assert res.used_cpu[pick].tolist() == ref.used_cpu.tolist() and res.used_mem[pick].tolist() == ref.used_mem.tolist()
    bad = np.argwhere(res.placement[pick] != ref.placement)
    assert len(bad) == 0, f"{len(bad)} placements differ, first (scenario, pod) = {bad[0].tolist()}"


@pytest.mark.parametrize("n_pods", [2047, 2048, 5003, 16384, 16385])
def test_placement_rows_through_the_lds_unpermute(n_pods):
    """`unpermute_lds_kernel` (rows of 2 048 .. 16 384 pods: the step-ordered row staged in LDS, gathered there) and the gather through L2 on both
    sides of its limits -- 16 384 pods ask for exactly the 64 KB a workgroup gets without an attribute; every placement row against the oracle."""
    prob, scen, orders = synth.config3(n_counts=3, n_orders=2, n_pods=n_pods, n_het=900)
    ref = O.run_threaded(prob, scen, orders)
    res, st, _ = run_gpu(prob, scen, orders)
    assert st.kernel_variant == capi.KERNEL_NARROW_CACHE, st.kernel_variant
    assert res.unscheduled.tolist() == ref.unscheduled.tolist() and (res.placement == ref.placement).all()
    assert res.used_cpu.tolist() == ref.used_cpu.tolist() and res.used_mem.tolist() == ref.used_mem.tolist()
