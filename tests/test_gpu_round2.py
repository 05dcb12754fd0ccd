"""GPU parity tests, round 2: the BENCHMARKED inputs themselves (every scenario of BASELINE config 3, full-size config-5
scenarios, config 4's shuffled orders), the multi-device / multi-rank entry points, and the ABI v3 additions."""
import json
import os
import subprocess
import tempfile
import sys
import threading

import numpy as np
import pytest

import oracle_lib as O
import randprob
from open_simulator_amd import capi, sweep, synth
from cabi_util import build_cabi_smoke
from test_gpu_parity import assert_same, run_gpu, _random_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_every_benchmarked_scenario():
    """The batch bench.py times (10k pods x 488..1511 nodes, 1024 counts x 4 orders = 4096 scenarios): EVERY scenario's
    unscheduled count, used cpu / memory and full placement row against the oracle (threaded: ~21 s on 16 host threads;
    on a box with fewer than 8 usable threads an evenly spaced 512-scenario sample instead)."""
    prob, scen, orders = synth.config3()
    pick = np.arange(len(scen)) if O.host_threads() >= 8 else np.unique(np.linspace(0, len(scen) - 1, 512).astype(int))
    ref = O.run_threaded(prob, scen[pick], orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        assert ctx.stats().kernel_variant == capi.KERNEL_NARROW_CACHE
        res = ctx.fetch(want_placement=True)
    assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist()
    assert res.used_cpu[pick].tolist() == ref.used_cpu.tolist() and res.used_mem[pick].tolist() == ref.used_mem.tolist()
    bad = np.argwhere(res.placement[pick] != ref.placement)
    assert len(bad) == 0, f"{len(bad)} placements differ, first (scenario, pod) = {bad[0].tolist()}"
    assert len(pick) >= 512


def test_config4_shuffled_orders_one_rank_of_eight():
    """BASELINE config 4 = config 3's pool with 32 pod orders (orders 3..31 are seeded Fisher-Yates shuffles), dealt over 8
    ranks: rank 5's shard, 96 scenarios spread over every order, against the oracle."""
    prob, scen_all, orders = synth.config3(n_orders=32, seed=synth.SEED + 4)
    scen = sweep.shard(scen_all, 5, 8)
    assert len(scen) == 4096 and set(scen[:, 1].tolist()) == {5, 13, 21, 29}      # a rank sees 4 orders of the 32
    # all 32 orders: take, for every order, three node counts from the unsharded grid
    pick = np.array([c * 32 + o for o in range(32) for c in (3 + o, 500 + 7 * o, 1023 - o)])
    sub = scen_all[pick]
    assert set(sub[:, 1].tolist()) == set(range(32))
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)
    # and the rank's own shard end to end: results of the sharded run equal the unsharded scenarios they came from
    k = np.unique(np.linspace(0, len(scen) - 1, 64).astype(int))
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        shard_res = ctx.fetch(want_placement=False)
        ref2 = O.run_threaded(prob, scen[k], orders)
        assert shard_res.unscheduled[k].tolist() == ref2.unscheduled.tolist()
        assert shard_res.used_cpu[k].tolist() == ref2.used_cpu.tolist()
        for j, s in enumerate(k.tolist()):
            assert (ctx.fetch_placement(s) == ref2.placement[j]).all(), s


@pytest.mark.parametrize("engine", ["score_table", "all_feature"])
def test_config5_sixteen_full_size_scenarios(engine, monkeypatch):
    """BASELINE config 5 at full size (50k pods x 2500..5000 nodes; GPU share + anti-affinity + taints): 16 of the 256
    benchmarked scenarios, every placement compared -- on the score-table kernel's REST path (generation 6, what the bench
    times) and on the all-feature kernel (SIMON_NO_REST=1)."""
    if engine == "all_feature":
        monkeypatch.setenv("SIMON_NO_REST", "1")
    prob, scen, orders = synth.config5()
    pick = np.unique(np.linspace(0, len(scen) - 1, 16).astype(int))
    ref = O.run_threaded(prob, scen[pick], orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        st = ctx.stats()
        if engine == "all_feature":
            assert st.kernel_variant == capi.KERNEL_WIDE
        else:
            assert st.kernel_variant == capi.KERNEL_NARROW_CACHE and st.kernel_generation == 6
        res = ctx.fetch(want_placement=False)
        assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist()
        assert res.used_cpu[pick].tolist() == ref.used_cpu.tolist() and res.used_mem[pick].tolist() == ref.used_mem.tolist()
        for j, s in enumerate(pick.tolist()):
            row = ctx.fetch_placement(s)
            bad = np.flatnonzero(row != ref.placement[j])
            assert len(bad) == 0, f"scenario {s}: {len(bad)} placements differ, first pod {bad[0]}"


@pytest.mark.parametrize("n_sigs", [64, 100, 126])
def test_config3_many_signatures(n_sigs):
    """config 3 with more distinct request signatures than one wave has lanes (the regime behind the 42-signature benchmark)."""
    prob, scen, orders = synth.config3(n_counts=48, n_orders=3, n_pods=6000, n_sigs=n_sigs)
    sub = scen[::5]
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert_same(res, ref)
    assert variant == capi.KERNEL_NARROW_CACHE, "more than 64 signatures must stay on the score-table kernel"


def test_large_pool_stays_on_cache_kernel():
    """Up to 4095 nodes the cpu+memory path keeps the score-table kernel (it used to stop at 2047)."""
    prob, scen, orders = synth.config3(n_counts=16, n_orders=2, n_pods=12000, n_het=3000)
    sub = scen[[0, 7, 15, 16, 31]]
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert_same(res, ref)
    assert variant == capi.KERNEL_NARROW_CACHE


def test_narrow_problem_beyond_8192_nodes_falls_through_to_the_all_feature_kernel():
    """a cpu+memory problem with more than 8 x 1024 nodes used to fail with SIMON_ERANGE."""
    prob, scen, orders = synth.config3(n_counts=4, n_orders=1, n_pods=3000, n_het=8400)
    sub = scen[[0, 3]]
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert variant == capi.KERNEL_WIDE
    assert_same(res, ref)


# ---- device groups (ABI v3) ------------------------------------------------------------------------------------
def test_group_two_members_on_one_device():
    """simon_group over the device list [0, 0]: two contexts, two host threads, scenario s on member s % 2; results in the
    caller's order and the cross-member minimum plan equal the single-context run and the oracle."""
    prob, scen, orders = synth.config3(n_counts=40, n_orders=3, n_pods=3000)
    ref = O.run_threaded(prob, scen, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        one = ctx.run_batch(scen, orders)
        plan1 = ctx.min_plan(100, 100)
        plan1_cap = ctx.min_plan(60, 100)
    with capi.Group([0, 0]) as g:
        assert g.size == 2
        g.load_problem(prob)
        res = g.run_batch(scen, orders)
        assert_same(res, ref)
        assert_same(res, one)
        plan, _ = g.min_plan(100, 100)
        assert plan.as_dict() == plan1.as_dict() == O.min_plan(prob, scen, ref).as_dict()
        plan_cap, _ = g.min_plan(60, 100)
        assert plan_cap.as_dict() == plan1_cap.as_dict() == O.min_plan(prob, scen, ref, 60, 100).as_dict()
        for s in (0, 1, 57, len(scen) - 1):
            assert (g.fetch_placement(s) == ref.placement[s]).all()
        assert g.member_stats(0).kernel_variant == g.member_stats(1).kernel_variant == capi.KERNEL_NARROW_CACHE
        # error path: fewer scenarios than members
        with pytest.raises(capi.SimonError):
            g.run_batch(scen[:1], orders)
    with pytest.raises(capi.SimonError):
        capi.Group([0, 99])


def test_group_of_three_on_the_all_feature_kernel():
    prob = randprob.rand_problem(4242, N=90, P=500, gpu=True, anti=True, static_mask=True, eph=True)
    scen, orders = randprob.rand_scenarios(3, prob, S=11)
    ref = O.run(prob, scen, orders)
    with capi.Group([0, 0, 0]) as g:
        g.load_problem(prob)
        assert_same(g.run_batch(scen, orders), ref)
        plan, _ = g.min_plan()
        assert plan.as_dict() == O.min_plan(prob, scen, ref).as_dict()


def test_two_contexts_on_two_os_threads():
    """The header's promise: distinct contexts may live on distinct OS threads.  Two different problems (score-table
    kernel / all-feature kernel) run concurrently, several rounds, each checked against the oracle."""
    pa, sa, oa = synth.config3(n_counts=24, n_orders=2, n_pods=2500)
    pb = randprob.rand_problem(99, N=120, P=600, gpu=True, anti=True, ipa=True, static_mask=True)
    sb, ob = randprob.rand_scenarios(9, pb, S=8)
    ra, rb = O.run_threaded(pa, sa, oa), O.run(pb, sb, ob)
    errors = []

    def worker(prob, scen, orders, ref, want_variant):
        try:
            with capi.Context(0) as ctx:
                ctx.load_problem(prob)
                for _ in range(4):
                    res = ctx.run_batch(scen, orders)
                    assert ctx.stats().kernel_variant == want_variant
                    assert_same(res, ref)
                    assert ctx.min_plan().as_dict() == O.min_plan(prob, scen, ref).as_dict()
        except BaseException as e:              # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(pa, sa, oa, ra, capi.KERNEL_NARROW_CACHE)),
          threading.Thread(target=worker, args=(pb, sb, ob, rb, capi.KERNEL_WIDE))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_explain_loaded_uses_the_scenarios_own_ranks():
    """Two loaded scenarios of ONE size with different node ranks: the replay of scenario 1 must break ties with scenario
    1's ranks (round 1 picked the first loaded scenario of that size)."""
    prob = randprob.rand_problem(6100, N=40, P=400, tight_pods=True, static_mask=True)
    prob.alloc_cpu[:] = prob.alloc_cpu[0]; prob.alloc_mem[:] = prob.alloc_mem[0]; prob.node_class[:] = 0
    n = 30
    scen = np.array([[n, 0], [n, 0], [n, 0]], np.int32)
    orders = np.arange(prob.n_pods, dtype=np.int32)[None]
    ranks = _random_ranks(np.random.default_rng(11), scen, prob.n_nodes)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.set_node_ranks(ranks)
        ctx.run_loaded(True)
        res = ctx.fetch(True)
        assert_same(res, O.run(prob, scen, orders, node_ranks=ranks))
        assert res.unscheduled.min() > 0, "the test needs unscheduled pods"
        differ = 0
        for s in range(3):
            ref, (nf, failed, codes) = O.run(prob, scen, orders, explain_scenario=s, max_failed=32, node_ranks=ranks)
            k, f2, c2 = ctx.explain_loaded(s, max_failed=32)
            assert k == nf and f2.tolist() == failed.tolist() and (c2 == codes).all(), s
            if s:
                ref0 = O.run(prob, scen, orders, explain_scenario=0, max_failed=32, node_ranks=ranks)[1]
                differ += int(len(ref0[1]) != len(failed) or (ref0[2] != codes).any() or (ref0[1] != failed).any())
        assert differ, "scenario 0's ranks would have given the same answer: the test does not discriminate"
        with pytest.raises(capi.SimonError, match="explain_loaded"):
            ctx.explain(n, orders[0])                       # ad-hoc scenario while ranks are loaded: refused
        with pytest.raises(capi.SimonError):
            ctx.explain_loaded(3)
        ctx.set_node_ranks(None)
        k, f2, c2 = ctx.explain_loaded(1, max_failed=32)    # pool order again
        ref, (nf, failed, codes) = O.run(prob, scen, orders, explain_scenario=1, max_failed=32)
        assert k == nf and f2.tolist() == failed.tolist() and (c2 == codes).all()


# ---- bench.py ---------------------------------------------------------------------------------------------------
def _bench(args, env=None, timeout=900, detail=None):
    """Runs bench.py; returns (process, FULL record).  The full record lives in the sidecar the compact last line names; the last
    line itself -- what the driver parses -- must be one JSON object of at most 4 KB that carries the contract's keys."""
    detail = detail or os.path.join(tempfile.mkdtemp(prefix="simon_bench_"), "bench_detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=dict(os.environ, SIMON_BENCH_DETAIL=detail, **(env or {})),
                         capture_output=True, text=True, timeout=timeout)
    lines = out.stdout.strip().splitlines()
    if out.returncode != 0 or not lines or not lines[-1].startswith("{"):
        return out, None
    assert len(lines[-1]) <= 4096, f"final bench line: {len(lines[-1])} bytes"
    line = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "parity_sample", "detail"):
        assert k in line, k
    assert "workload" in line["config"] and line["detail"] == detail
    with open(detail) as f:
        full = json.load(f)
    assert full["value"] == line["value"] and full["config"]["workload"] == line["config"]["workload"]
    full["_line"] = line
    return out, full


def test_bench_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher and WORLD_SIZE unset must run TWO ranks (round 1 parsed the flag and ran
    one).  On a single-GPU box the ranks share device 0 over gloo (SIMON_BENCH_SHARE_DEVICE=1, test hook)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SIMON_BENCH_SHARE_DEVICE"] = "1"
    env["SIMON_BENCH_DETAIL"] = os.path.join(tempfile.mkdtemp(prefix="simon_bench_"), "bench_detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--counts", "32",
                          "--pods", "2000"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["ranks"]["backend"] == "gloo" and line["ranks"]["one_device_test_hook"] is True
    # the parsed line says how many ranks went through RCCL (none here: gloo) and which device every rank sat on
    assert line["ranks"]["rccl_ranks"] == 0 and len(line["ranks"]["devices"]) == 2 and all(d.startswith("0:") for d in line["ranks"]["devices"])
    d = json.load(open(env["SIMON_BENCH_DETAIL"]))
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo"
    assert "self-spawned" in d["ranks"]["launched_by"]
    # the multi-rank self-check: what the ranks saw (the hook is stated, so a one-device run cannot pass for a multi-GPU one)
    sc = d["ranks"]["selfcheck"]
    assert sc["one_device_test_hook"] is True and sc["distinct_devices_over_ranks"] == 1 and len(sc["devices"]) == 2
    assert d["ranks"]["all_gather_ms_per_step"] > 0
    assert d["config"]["scenarios_per_gpu"] == 32 * 4 and "8 pod orders = 256 scenarios" in d["config"]["workload"]
    assert d["parity_sample"]["mismatches"] == 0 and d["parity_sample"]["scenarios"] >= 16
    assert d["parity_sample"]["placement_rows"] == d["parity_sample"]["scenarios"]
    # without the test hook a box with one GPU must refuse instead of silently running one rank
    env.pop("SIMON_BENCH_SHARE_DEVICE")
    if capi.load_library().simon_hip_device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                             text=True, timeout=300)
        assert out.returncode != 0 and "only 1 GPU" in (out.stderr + out.stdout)


def test_bench_line_is_self_verifying():
    """One reduced single-GPU bench line: roofline fractions are fractions, the parity sample covers placements, the end-to-end leg
    ran, and every sub-record (config 2, config 5 at the small and at the saturating batch size) is a measurement of its own:
    roofline record, CPU baseline, parity sample."""
    out, d = _bench(["--steps", "2", "--warmup", "1", "--counts", "64", "--pmc", "replay"],
                    env={"SIMON_BENCH_CPU_BUDGET_S": "2", "SIMON_BENCH_C5_SCEN": "16", "SIMON_BENCH_C5_CHECK": "3"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["kernel"] == "score_table"
    ps = d["parity_sample"]
    assert ps["mismatches"] == 0 and ps["scenarios"] >= 16 and ps["placement_rows"] == ps["scenarios"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["peak"] == pytest.approx(1228.8)
    assert r["frac"] is None or 0 < r["frac"] < 1                # replayed counters only exist for the full-size workload
    assert r["algorithmic_ratio"] > 0 and r["kernel"] == "simon::table_kernel"
    e2e = d["end_to_end"]
    assert e2e["batch_ms"] > e2e["kernel_ms"] > 0 and e2e["load_problem_ms"] > 0 and e2e["scenarios"] == d["config"]["scenarios_per_gpu"]
    names = [w["workload"] for w in d["other_workloads"]]
    assert names == ["config2", "config3_sigs200", "config3_service", "config3_service_anti20", "config3_service_pref60", "config5_S16", "config5_S2048",
                     "config3_service_S64", "config3_service_gpu20_S256", "config3_service_shapes30_S256", "config5_service_S16", "config5_asdrawn_service_S16", "typical_cluster_x64", "wide_mix_x64", "config3_sigs300", "config3_classes80", "config3_classes160", "config3_S64"]
    for w in d["other_workloads"]:
        assert "error" not in w, w
        assert w["parity_sample"]["mismatches"] == 0 and w["value"] > 0
        assert w["parity_sample"]["placement_rows"] == w["parity_sample"]["scenarios"] >= 1
        assert w["cpu_baseline"]["kind"] == "port" and w["cpu_baseline"]["value"] > 0
        if w["workload"] in ("config3_sigs300",):   # the cliff row: whichever kernel the host prefers there (beyond 256 signatures generation 2 where it is eligible)
            assert w["roofline"]["kernel_ms"] > 0 and w["roofline"]["kernel"] in ("simon::fast_kernel", "simon::narrow_kernel", "simon::wide_kernel")
            continue
        assert w["roofline"]["kernel_ms"] > 0 and w["roofline"]["kernel"] == ("simon::wide_kernel" if w["workload"] == "wide_mix_x64" else "simon::table_kernel")
    by = {w["workload"]: w for w in d["other_workloads"]}
    # the former cliffs as numbers: 80 node shapes stay on the score table (two classes per lane), and since the end of round 6 so do 160
    assert by["config3_classes80"]["kernel_generation"] in (4, 5)
    assert by["config3_classes160"]["kernel_generation"] == 4               # (129 .. 256 node classes: simon_table_cls4.hip, end of round 6)
    assert by["config3_service_gpu20_S256"]["kernel_generation"] == 7         # a gpushare cluster behind Services: GPU share folded into the table
    assert by["config3_service_shapes30_S256"]["kernel_generation"] == 7      # 30 node shapes x 3 zones: two node classes per lane in the walks (round 6)
    assert by["config5_service_S16"]["kernel_generation"] == 7 and by["config5_service_S16"]["pods"] == 50000   # config 5's shape behind Services
    assert by["config5_asdrawn_service_S16"]["kernel_generation"] == 7 and by["config5_asdrawn_service_S16"]["pods"] == 50000   # ... and as drawn: the walks over the mask rows (round 6)
    # the `simon apply` shapes: 64 candidate scenarios run generation 7 in team mode, with the one-wave time of the same batch beside it
    for w in (by["config3_service_S64"], by["typical_cluster_x64"]):
        assert w["kernel_generation"] == 7 and w["workgroup"] == 256 and w["team"]["waves_per_scenario"] == 4 and w["team"]["one_wave_kernel_ms"] > 0
    assert by["wide_mix_x64"]["kernel"].startswith("wide") and by["wide_mix_x64"]["scenarios"] == 64
    assert by["config3_sigs200"]["kernel_generation"] == 5 and by["config3_sigs200"]["scenarios"] == 4096
    assert by["config3_service"]["kernel_generation"] == 7 and by["config3_service"]["scenarios"] == 4096
    assert by["config3_service_anti20"]["kernel_generation"] == 7 and by["config3_service_anti20"]["scenarios"] == 4096
    assert by["config3_service_pref60"]["kernel_generation"] == 7 and by["config3_service_pref60"]["scenarios"] == 4096
    assert by["config5_S16"]["kernel_generation"] == 6 and by["config5_S2048"]["scenarios"] == 2048
    assert by["config3_S64"]["kernel_generation"] in (4, 5) and by["config3_S64"]["scenarios"] == 64      # BASELINE config 3's pool at the batch a `simon apply` offers
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    for w in d["other_workloads"]:
        assert w["steps"] >= 5, w["workload"]                     # every sub-record times at least five steps
    # what the DRIVER reads: the compact last line (size asserted in _bench) with the headline roofline, the CPU baseline and a digest
    line = d["_line"]
    assert line["roofline"]["kernel"] == "simon::table_kernel" and line["roofline"]["kernel_ms"] > 0 and "frac" in line["roofline"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1
    assert line["parity_sample"]["mismatches"] == 0
    assert [r[0] for r in line["digest"]["rows"]] == names and all(r[-1] == 0 for r in line["digest"]["rows"])


def test_bench_group_mode_one_process_two_members():
    """`bench.py --group 2`: ONE process, simon_group over two members (both on device 0 under the test hook): the Go-host shape."""
    out, d = _bench(["--group", "2", "--steps", "2", "--warmup", "1", "--counts", "32", "--pods", "2000"],
                    env={"SIMON_BENCH_SHARE_DEVICE": "1", "SIMON_BENCH_CPU_BUDGET_S": "2"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ranks"]["devices"] == [0, 0] and "simon_group" in d["ranks"]["mode"]
    assert d["config"]["scenarios_per_gpu"] == 32 * 4 and len(d["ranks"]["member_kernel_ms"]) == 2
    assert d["parity_sample"]["mismatches"] == 0 and d["parity_sample"]["scenarios"] >= 8
    if capi.load_library().simon_hip_device_count() < 2:          # without the hook: refuse, do not fold the members onto one device
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group", "2", "--steps", "1"], capture_output=True, text=True,
                             timeout=300, env={k: v for k, v in os.environ.items() if k != "SIMON_BENCH_SHARE_DEVICE"})
        assert out.returncode != 0 and "only 1 GPU" in (out.stderr + out.stdout)


# ---- a non-Python consumer of the C-ABI -------------------------------------------------------------------------
def test_c_consumer_of_the_abi(tmp_path):
    """tests/cabi/cabi_smoke.c, built with plain gcc against include/simon_hip.h: load -> run_batch -> min_plan -> explain /
    explain_loaded, the device group, and two contexts on two pthreads, against the committed binary fixtures."""
    exe = build_cabi_smoke(tmp_path)
    fx = [os.path.join(ROOT, "tests", "golden", n) for n in ("cabi_kav.bin", "cabi_config2_sweep.bin", "cabi_features.bin")]
    out = subprocess.run([exe, *fx], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    # cabi_features.bin: every optional array the Go shim fills (ephemeral storage, extended resources, GPU share with arriving
    # gpu-index lists, presets / gates / pins, static masks) + the golden gpu_slices, from plain C
    assert "cabi_smoke ok" in out.stdout and out.stdout.count("group of 2 members") == 3


def test_score_table_kernel_randomised_slice():
    """A slice of tests/fuzz_table.py (random sizes up to 4 095 nodes, up to 128 signatures, caller classes that do not share
    their allocatable, every cpu+memory feature) in the regular suite."""
    import fuzz_table
    bad, on_table, two_level = [], 0, 0
    for case in range(0, 24):
        ok, info = fuzz_table.one_case(case)
        on_table += info["generation"] in (4, 5)
        two_level += info["generation"] == 5
        if not ok:
            bad.append(info)
    assert not bad, bad
    assert on_table >= 12, "the slice should mostly run on the score-table kernel"
    assert two_level >= 3, "every third case forces the two-level summary"


def _run_two_level(prob, scen, orders, monkeypatch):
    monkeypatch.setenv("SIMON_TABLE_COARSE", "1")           # read once per context (simon_ctx_create)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(True)
        return ctx.fetch(True), ctx.stats()


@pytest.mark.parametrize("n_sigs", [42, 100, 126])
def test_two_level_summary_config3(n_sigs, monkeypatch):
    """Generation 5 (LDS entries of 64 positions, per-16 entries and counters in the HBM workspace) on config-3 shaped batches."""
    prob, scen, orders = synth.config3(n_counts=48, n_orders=3, n_pods=6000, n_sigs=n_sigs)
    sub = scen[1::6]
    ref = O.run_threaded(prob, sub, orders)
    res, st = _run_two_level(prob, sub, orders, monkeypatch)
    assert st.kernel_generation == 5
    assert_same(res, ref)


def test_two_level_summary_large_pool(monkeypatch):
    prob, scen, orders = synth.config3(n_counts=16, n_orders=2, n_pods=12000, n_het=3000)
    sub = scen[[1, 8, 15, 17, 30]]
    ref = O.run_threaded(prob, sub, orders)
    res, st = _run_two_level(prob, sub, orders, monkeypatch)
    assert st.kernel_generation == 5
    assert_same(res, ref)


def test_pool_of_6000_nodes_runs_on_the_two_level_score_table():
    """Between 4 096 and 8 191 nodes the cpu+memory path keeps the score-table kernel (13-bit position / canonical fields)."""
    prob, scen, orders = synth.config3(n_counts=8, n_orders=2, n_pods=9000, n_het=6000)
    sub = scen[[0, 5, 15]]
    ref = O.run_threaded(prob, sub, orders)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(sub, orders)
        ctx.run_loaded(True)
        res, st = ctx.fetch(True), ctx.stats()
    assert st.kernel_generation == 5
    assert_same(res, ref)


def test_pod_classes_are_interned_by_content_for_the_score_table():
    """Thousands of pod classes with identical static-mask and Simon rows (one per workload template) are ONE table class."""
    prob, scen, orders = synth.config3(n_counts=12, n_orders=2, n_pods=3000)
    P = len(prob.req_cpu)
    Cp0 = prob.n_pod_classes
    rep = 40                                                     # 40 copies of every class row
    prob.pod_class = (prob.pod_class + Cp0 * (np.arange(P) % rep)).astype(np.int32)
    prob.simon_raw = np.tile(prob.simon_raw, (rep, 1))
    if prob.static_mask is not None:
        prob.static_mask = np.tile(prob.static_mask, (rep, 1))
    if getattr(prob, "static_reason", None) is not None:
        prob.static_reason = np.tile(prob.static_reason, (rep, 1))
    if getattr(prob, "const_score", None) is not None:
        prob.const_score = np.tile(prob.const_score, rep)
    prob.n_pod_classes = Cp0 * rep
    sub = scen[::5]
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)


def test_many_signatures_at_batch_scale_pick_the_two_level_summary():
    """With 100 signatures x 1 464 nodes the one-level summary lets a CU hold 7 scenario waves; a batch that offers 16 per CU
    switches itself (simon_hip.hip: simon_load_scenarios, cost model)."""
    prob, scen, orders = synth.config3(n_counts=64, n_orders=64, n_pods=1000, n_sigs=100, n_het=1400)
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(False)
        st = ctx.stats()
        res = ctx.fetch(False)
    assert st.kernel_generation == 5 and st.lds_bytes <= 10240
    pick = np.arange(0, len(scen), 97)
    ref = O.run_threaded(prob, scen[pick], orders)
    assert res.unscheduled[pick].tolist() == ref.unscheduled.tolist()
    assert res.used_cpu[pick].tolist() == ref.used_cpu.tolist()


def test_rest_path_randomised_slice():
    """A slice of tests/fuzz_rest.py (GPU share and / or node-level anti-affinity, up to 8 191 nodes, many pod classes) in the
    regular suite."""
    import fuzz_rest
    bad, on_rest = [], 0
    for case in range(0, 24):
        ok, info = fuzz_rest.one_case(case)
        on_rest += info["generation"] == 6
        if not ok:
            bad.append(info)
    assert not bad, bad
    assert on_rest >= 6, "a good part of the slice should run on generation 6"


def test_ephemeral_allocatable_without_requests_keeps_the_score_table():
    """Nodes that advertise ephemeral storage nobody asks for (every real node does) do not push a cpu+memory problem onto the
    all-feature kernel: `Allocatable < request + Requested` is 0 < 0 on every node (fit.go:264-270)."""
    prob, scen, orders = synth.config3(n_counts=12, n_orders=2, n_pods=2500)
    prob.alloc_eph = np.full(len(prob.alloc_cpu), 100 << 30, np.int64)
    sub = scen[::4]
    ref = O.run_threaded(prob, sub, orders)
    res, variant = run_gpu(prob, sub, orders)
    assert variant == capi.KERNEL_NARROW_CACHE
    assert_same(res, ref)
