"""CPU tests of the host-side mirror: FitError strings, the C-ABI surface, Quantity conversions."""
import ctypes
import os
import re
import sys

import pytest

import numpy as np

import oracle_lib as O
from open_simulator_amd import capi, fiterror
from open_simulator_amd.quantity import parse_quantity as pq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GiB = 1 << 30


def test_fit_error_string_matches_reference_format():
    # 3 nodes: cpu+mem short, cpu short, too many pods + mem short (same case as test_oracle.test_fit_reasons...)
    prob = capi.Problem(alloc_cpu=[1000, 1000, 4000], alloc_mem=[GiB, 4 * GiB, GiB], alloc_pods=[110, 110, 0],
                        req_cpu=[2000], req_mem=[2 * GiB], n_pod_classes=1, n_node_classes=1)
    _, (nf, failed, codes) = O.run(prob, [[3, 0]], np.arange(1)[None], explain_scenario=0, max_failed=2)
    assert nf == 1
    msg = fiterror.fit_error(codes[0])
    # V/core/generic_scheduler.go:72-90: "<count> <reason>" sorted as strings
    assert msg == "0/3 nodes are available: 1 Too many pods, 2 Insufficient cpu, 2 Insufficient memory."
    full = fiterror.unscheduled_reason("simple", "big-pod", codes[0])
    assert full.startswith("failed to schedule pod (simple/big-pod): Unschedulable: 0/3 nodes are available: ")


def test_fit_error_mixed_plugins():
    F = capi
    codes = [F.FAIL_STATIC | 7, F.FAIL_STATIC | 3, F.FAIL_ANTI_INCOMING, F.FAIL_ANTI_EXISTING, F.FAIL_GPUSHARE,
             F.FAIL_FIT | F.FIT_CPU | (F.FIT_SCALAR0 << 1)]
    msg = fiterror.fit_error(codes, node_names=[f"n{j}" for j in range(6)],
                             static_reasons={7: fiterror.taint_reason("node-role.kubernetes.io/master", "")},
                             scalar_names=["example.com/foo", "alibabacloud.com/gpu-count"])
    assert msg == ("0/6 nodes are available: 1 Insufficient alibabacloud.com/gpu-count, 1 Insufficient cpu, 1 Node:n4, "
                   "1 node(s) didn't match Pod's node affinity, 1 node(s) didn't match pod anti-affinity rules, "
                   "1 node(s) didn't satisfy existing pods anti-affinity rules, "
                   "1 node(s) had taint {node-role.kubernetes.io/master: }, that the pod didn't tolerate, "
                   "2 node(s) didn't match pod affinity/anti-affinity.")


def test_library_exports_every_declared_symbol():
    """include/simon_hip.h is the contract: every function it declares must be exported (no compute calls here)."""
    hdr = open(os.path.join(ROOT, "include", "simon_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(simon_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(capi.EXPORTS)
    lib = ctypes.CDLL(capi.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.simon_hip_version() == capi.ABI_VERSION
    # without a device the context constructor fails cleanly instead of crashing or falling back
    lib.simon_ctx_create.restype = ctypes.c_void_p
    if lib.simon_hip_device_count() == 0:
        assert lib.simon_ctx_create(0) is None


def test_build_covers_every_hip_unit():
    """__graft_entry__.build() compiles an explicit list of translation units: a .hip file in csrc/ that is not on it (or an #include of
    the kernel source without the dependency that makes it rebuild) would silently stay out of libsimon_hip.so."""
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, "open-simulator_amd", "csrc")
    on_disk = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    assert on_disk == sorted(g.HIP_SOURCES)
    for f in on_disk:
        for inc in re.findall(r'#include "([a-z_0-9]+\.hip)"', open(os.path.join(csrc, f)).read()):
            assert inc in g.EXTRA_DEPS.get(f, []), (f, inc)
    for h in re.findall(r'#include "(simon_[a-z_]+\.h)"', "".join(open(os.path.join(csrc, f)).read() for f in on_disk)):
        assert h in g.HIP_HEADERS or os.path.join(ROOT, "include", h) in g.HIP_HEADERS, h


def test_struct_layouts_match_the_header():
    """ctypes mirrors must have the sizes the C compiler gives the header structs."""
    import subprocess, tempfile, textwrap
    src = textwrap.dedent("""
        #include <stdio.h>
        #include "simon_hip.h"
        int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(simon_nodes_soa), sizeof(simon_pods_soa),
            sizeof(simon_class_tables), sizeof(simon_scenario), sizeof(simon_batch_out), sizeof(simon_plan), sizeof(simon_stats)); return 0; }
    """)
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, c])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (capi.NodesSoA, capi.PodsSoA, capi.ClassTables, capi.Scenario, capi.BatchOut, capi.Plan,
                                       capi.Stats)]
    assert sizes == mine


def test_quantity_value_is_ceil():
    # quantity.go:729-750: Value()/MilliValue() round up
    assert pq("100m").milli_value() == 100 and pq("1").milli_value() == 1000
    assert pq("1500m").int_value() == 2 and pq("0.5").milli_value() == 500
    assert pq("16Gi").int_value() == 16 * GiB and pq("256000Mi").int_value() == 256000 << 20
    assert pq("1n").milli_value() == 1


def test_c_consumer_builds_and_links_against_the_header(tmp_path):
    """tests/cabi/cabi_smoke.c compiles with plain gcc against include/simon_hip.h and links libsimon_hip.so; without a
    GPU it stops after the version handshake (exit code 77).  The GPU suite runs it against the fixtures."""
    import subprocess
    from cabi_util import build_cabi_smoke
    exe = build_cabi_smoke(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "cabi_kav.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode in (0, 77), out.stdout + out.stderr
    assert f"ABI {capi.ABI_VERSION}" in out.stdout or "cabi_smoke ok" in out.stdout
    # cabi_terms.c: the consumer that attaches EVERY optional array by name (what integration/go/hipengine/flatten_terms.go fills)
    exe = build_cabi_smoke(tmp_path, "cabi_terms")
    out = subprocess.run([exe, os.path.join(root, "tests", "golden", "cabi_terms.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode in (0, 77), out.stdout + out.stderr


def test_cabi_fixtures_match_the_oracle():
    """The committed binary fixtures of the C consumer are what the oracle computes today."""
    import struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from test_oracle import kav_problem
    prob, _ = kav_problem(n_pods=10)
    raw = open(os.path.join(root, "tests", "golden", "cabi_kav.bin"), "rb").read()
    assert raw[:8] == b"SIMONFX1"
    N, P, Cp, Cn, S, n_orders = struct.unpack("<6i", raw[8:32])
    assert (N, P, S, n_orders) == (2, 10, 2, 1)
    ref = O.run(prob, [[2, 0], [1, 0]], np.arange(10, dtype=np.int32)[None])
    off = 32 + N * 8 * 2 + N * 4 * 2 + P * 8 * 2 + P * 4 + Cp * Cn * 8 + S * 8 + n_orders * P * 4
    un = np.frombuffer(raw, "<i4", S, off)
    pl = np.frombuffer(raw, "<i4", S * P, off + S * 4 + S * 16).reshape(S, P)
    assert un.tolist() == ref.unscheduled.tolist() == [0, 2]
    assert (pl == ref.placement).all() and pl[0, :2].tolist() == [0, 0]      # the first two pods land on node A


def test_cabi_fixtures_are_regenerated_byte_for_byte(tmp_path, monkeypatch):
    """tests/golden/make_cabi_fixture.py rewrites all three fixtures of the C consumer from today's oracle: identical files (the
    SIMONFX2 one holds every optional array the Go shim fills and the golden gpu_slices)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_cabi_fixture", os.path.join(root, "tests", "golden", "make_cabi_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "HERE", str(tmp_path))
    mod.main()
    for name in ("cabi_kav.bin", "cabi_config2_sweep.bin", "cabi_features.bin"):
        assert open(tmp_path / name, "rb").read() == open(os.path.join(root, "tests", "golden", name), "rb").read(), name
    assert open(tmp_path / "cabi_features.bin", "rb").read()[:8] == b"SIMONFX2"


def test_bench_final_line_fits_the_driver(capsys, tmp_path, monkeypatch):
    """The driver keeps the last ~8 KB of bench.py's stdout and parses the LAST line.  Round 3's record (24.8 KB on one line, committed
    as profiles/r03/r03f_bench_default.json) lost its head that way: `parsed` was null.  bench.emit() must put the full record into the
    sidecar + `#detail` lines and end with ONE compact JSON line that still carries the contract's keys, `roofline`, `cpu_baseline`,
    `parity_sample` and one digest row per other workload -- also when many more sub-records are added."""
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03", "r03f_bench_default.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    full["other_workloads"] = full["other_workloads"] * 3                      # 21 sub-records: the line must still fit
    side = tmp_path / "bench_detail.json"
    monkeypatch.setenv("SIMON_BENCH_DETAIL", str(side))
    bench.emit(full)
    lines = capsys.readouterr().out.strip().splitlines()
    last = lines[-1]
    assert len(last) <= bench.LINE_BUDGET <= 3072 and all(l.startswith("#detail ") for l in lines[:-1])
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == full[k], k
    assert d["config"]["workload"] == full["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["frac"] == full["roofline"]["frac"] and r["kernel_ms"] == full["roofline"]["kernel_ms"]
    assert r["traffic"] == full["roofline"]["traffic"] and r["achieved"] and r["peak"] and r["unit"] and r["counters"] == "live"
    assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 16
    assert d["parity_sample"] == {"scenarios": 2376, "placement_rows": 2376, "mismatches": 0}
    assert len(d["digest"]["rows"]) == 21 and d["digest"]["rows"][0][0] == "config2" and len(d["digest"]["cols"]) == len(d["digest"]["rows"][0])
    assert json.load(open(side)) == full                                         # nothing is lost: the sidecar holds the whole record


def test_bench_line_of_an_eight_rank_rccl_run_names_its_ranks_and_devices(capsys, tmp_path, monkeypatch):
    """the day `bench.py --gpus 8` runs on a node, the driver's SCALE record must show `rccl_ranks: 8` and the device
    of every rank in the PARSED line (not only in the sidecar).  emit() on a record shaped like the one rank 0 assembles at world 8."""
    import json
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03", "r03f_bench_default.json")).read().strip().splitlines()[-1])
    full["n_gpus"] = 8
    full["ranks"] = {"world_size": 8, "backend": "nccl", "collective": "all_gather of one 32-byte plan record per rank and step", "all_gather_ms_per_step": 0.05,
                     "selfcheck": {"visible_devices": 8, "distinct_devices_over_ranks": 8, "one_device_test_hook": False, "rccl_version": "2.26.6",
                                   "devices": [f"rank {r}: local_rank {r}" for r in range(8)],
                                   "device_list": [f"{r}:GPU-{r:02x}deadbeef" for r in range(8)]},
                     "launched_by": "external launcher"}
    monkeypatch.setenv("SIMON_BENCH_DETAIL", str(tmp_path / "d.json"))
    bench.emit(full)
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert len(last) <= bench.LINE_BUDGET
    rk = json.loads(last)["ranks"]
    assert rk["rccl_ranks"] == 8 and rk["world_size"] == 8 and rk["backend"] == "nccl" and rk["distinct_devices"] == 8
    assert len(rk["devices"]) == 8 and rk["devices"][3].startswith("3:")


def test_bench_record_stays_the_last_stdout_line_when_a_library_prints(tmp_path):
    """RCCL prints its version banner to stdout through C stdio when a communicator comes up; through a pipe that text arrives at exit,
    AFTER bench.py's record (seen on the GPU box with a one-rank `nccl` group: profiles/r04/r06b_*).  bench.claim_stdout() keeps the real
    stdout for the record and points fd 1 at stderr: a C-stdio banner printed before or after the record must not follow it."""
    import json
    import subprocess
    code = (
        "import ctypes, json, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "libc = ctypes.CDLL(None)\n"
        "bench.claim_stdout()\n"
        "libc.printf(b'RCCL version : banner before\\n')\n"
        "bench.emit({'metric': 'm', 'value': 1.0, 'unit': 'u', 'n_gpus': 1, 'steps': 1, 'warmup': 0, 'ms_per_step': 1.0, 'higher_is_better': True,\n"
        "            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic', 'config': {'workload': 'w'}})\n"
        "libc.printf(b'Librccl path : banner after\\n')\n"
        "print('a stray python print')\n")
    env = dict(os.environ, SIMON_BENCH_DETAIL=str(tmp_path / "d.json"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert json.loads(lines[-1])["metric"] == "m" and all(l.startswith("#detail ") for l in lines[:-1])
    assert "banner before" in out.stderr and "banner after" in out.stderr and "stray python print" in out.stderr


def test_quantity_binary_si_and_open_local_reason_texts():
    """UnscheduledPod.Reason for Open-Local failures is err.Error() of open-local's predicate (pkg/simulator/plugin/open-local.go:78-88).
    The strings below are written out by hand from vendor/github.com/alibaba/open-local/pkg/scheduler/errors/errors.go (formats) and
    vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go:407-444 (resource.NewQuantity(x, BinarySI).String())."""
    from open_simulator_amd import capi, fiterror
    q = fiterror.quantity_binary_si
    # below 1024: DecimalSI with an exponent that is a multiple of three; from 1024 on: the factors of 1024 become the suffix
    assert [q(v) for v in (0, 1, 20, 100, 999, 1000, 1023)] == ["0", "1", "20", "100", "999", "1k", "1023"]
    assert [q(v) for v in (1024, 1025, 1536, 2048, 1 << 20, 3 << 20, 1 << 30, 100 << 30, 1500 << 20, 1 << 40, 1 << 60)] == \
        ["1Ki", "1025", "1536", "2Ki", "1Mi", "3Mi", "1Gi", "100Gi", "1500Mi", "1Ti", "1Ei"]
    G = 1 << 30
    r = fiterror.local_reason
    assert r([capi.LOCAL_ERR_LVM, 40 * G, 170 * G, 200 * G], "n1") == "Insufficient LVM storage, requested 40Gi, used 170Gi, capacity 200Gi"
    assert r([capi.LOCAL_ERR_NO_VG, 0, 0, 0], "worker-7") == "not LVM on node worker-7"
    assert r([capi.LOCAL_ERR_NO_SUCH_VG, 1, 0, 0], "n1", ["yoda-pool0", "share"]) == "not LVM named share"
    assert r([capi.LOCAL_ERR_DEVICE, 2, 1, 4], "n1") == "Insufficient Device storage, requested 2, available 1, capacity 4"
    # the histogram counts each distinct text (generic_scheduler.go:72-90): two nodes with the same sizes share an entry
    codes = [capi.FAIL_LOCAL_LVM, capi.FAIL_LOCAL_LVM, capi.FAIL_LOCAL, capi.FAIL_LOCAL_DEV, capi.FAIL_LOCAL_LVM]
    detail = [[capi.LOCAL_ERR_LVM, 10 * G, 95 * G, 100 * G], [capi.LOCAL_ERR_LVM, 10 * G, 95 * G, 100 * G], [0, 0, 0, 0],
              [capi.LOCAL_ERR_DEVICE, 1, 0, 2], [capi.LOCAL_ERR_NO_VG, 0, 0, 0]]
    assert fiterror.fit_error(codes, node_names=list("abcde"), local_detail=detail) == (
        "0/5 nodes are available: 1 Insufficient Device storage, requested 1, available 0, capacity 2, "
        "1 not LVM on node e, 2 Insufficient LVM storage, requested 10Gi, used 95Gi, capacity 100Gi.")
    with pytest.raises(ValueError):
        fiterror.fit_error(codes)                                      # the sizes are not optional


def test_bench_strong_mode_keeps_the_batch_fixed():
    """--strong: the workload is config 3's 4 096 scenarios whatever the world size (weak: 4 orders per rank)."""
    import argparse
    import bench
    from open_simulator_amd import synth
    base = dict(workload="config3", orders_per_gpu=4, counts=16, pods=300, c5_scenarios=0)
    (prob, scen_w, _), n_w = bench.build_workload(argparse.Namespace(strong=False, **base), synth, 4)
    (_, scen_s, _), n_s = bench.build_workload(argparse.Namespace(strong=True, **base), synth, 4)
    (_, scen_1, _), n_1 = bench.build_workload(argparse.Namespace(strong=False, **base), synth, 1)
    assert (n_w, n_s, n_1) == (16, 4, 4) and len(scen_w) == 4 * len(scen_s) and (scen_s == scen_1).all()


def test_spill_scan_flags_a_spill_ahead_of_the_exec_restore(tmp_path):
    """profiles/spill_scan.py (run by build() over every unit's ISA): the one pattern that cost a wrong `used_vg` in an experimental build -- a VGPR
    spill in a join block's prologue, BEFORE the block's `s_or_b64 exec, exec, ...` -- is reported; the same store after the restore, or behind
    real work of the block, is not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("spill_scan", os.path.join(ROOT, "profiles", "spill_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = tmp_path / "bad.s"
    bad.write_text("kern:\n\ts_cbranch_execz .LBB0_2\n.LBB0_1:\n\tglobal_store_dword v[2:3], v1, off\n.LBB0_2:\n\ts_mov_b32 s61, s39\n"
                   "\tscratch_store_dwordx2 off, v[12:13], off offset:796 ; 8-byte Folded Spill\n\ts_or_b64 exec, exec, s[0:1]\n\ts_endpgm\n")
    n, found = mod.scan(str(bad))
    assert n == 1 and len(found) == 1 and found[0][0] == "kern" and found[0][1] == ".LBB0_2"
    good = tmp_path / "good.s"
    good.write_text("kern:\n.LBB0_2:\n\ts_or_b64 exec, exec, s[0:1]\n\tscratch_store_dwordx2 off, v[12:13], off offset:796 ; 8-byte Folded Spill\n"
                    ".LBB0_3:\n\tv_add_u32_e32 v0, 1, v0\n\tscratch_store_dword off, v0, off offset:4 ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[2:3]\n\ts_endpgm\n")
    assert mod.scan(str(good)) == (2, [])
