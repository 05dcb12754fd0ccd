#!/usr/bin/env python3
"""Same-box A/B of library builds over the digest workloads: kernel time (HIP events, best and mean of `reps`) and a checksum of every
placement row, so that two builds can be compared for speed AND for identical results in one gpurun call.
usage: python profiles/ab_probe.py <workload,workload,...> [reps=3]      (library: SIMON_HIP_LIB, see profiles/build_variant.sh)
workloads: c5s256 c5s2048 c5s64 c3 c3s64 c2 widemix typical service64 service shapes30 c5shapes80 c5asdrawn c5asdrawn64 c5service c3sig200 c3cls80 c3cls160"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                      # noqa: E402
from open_simulator_amd import capi, synth              # noqa: E402


def build(name):
    if name.startswith("c5s") and name[3:].isdigit():
        return synth.config5(n_scen=int(name[3:]), n_orders=4)
    if name == "c3":
        return synth.config3(n_counts=1024, n_orders=4, n_pods=10000, seed=synth.SEED + 3)
    if name == "c3s64":
        return synth.config3(n_counts=16)
    if name == "c2":
        return synth.config2()
    if name == "service":
        return synth.config_service()
    if name == "service64":
        return synth.config_service(n_counts=16)
    if name == "shapes30":                           # 30 node shapes x 3 zones behind Services: 90 internal node classes (CN2), 256 scenarios
        return synth.config_service(n_counts=64, n_shapes=30)
    if name == "c5shapes80":                         # config 5 on 80 node shapes: generation 6 with two node classes per lane (CN2), 64 scenarios
        return synth.config5(n_scen=64, n_orders=4, n_shapes=80)
    if name == "c5asdrawn":                          # BASELINE config 5 as it is drawn with its anti-affinity groups behind Services: REST && SPREAD (simon_table_rs.hip)
        return synth.config5(n_scen=256, n_orders=4, services=True)
    if name == "c5asdrawn64":
        return synth.config5(n_scen=64, n_orders=4, services=True)
    if name == "c5service":
        return synth.config5_service(n_scen=256, n_orders=4)
    if name == "typical":
        return synth.typical_cluster_sweep()
    if name == "c3sig200":
        return synth.config3(n_counts=1024, n_orders=4, n_pods=10000, seed=synth.SEED + 3, n_sigs=200)
    if name == "c3cls80":
        return synth.config3_classes(80)
    if name == "c3cls160":                           # 160 node shapes: generation 4 for 129 .. 256 classes (simon_table_cls4.hip)
        return synth.config3_classes(160)
    if name == "widemix":
        import bench
        return bench.wide_mix_sweep()
    raise SystemExit(f"unknown workload {name}")


def main():
    names = sys.argv[1].split(",")
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = os.environ.get("SIMON_HIP_LIB", "libsimon_hip.so")
    for name in names:
        t0 = time.perf_counter()
        prob, scen, orders = build(name)
        t_build = time.perf_counter() - t0
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ts = []
            for _ in range(reps + 1):
                ctx.run_loaded(True)
                ts.append(ctx.stats().kernel_ms)
            st = ctx.stats()
            res = ctx.fetch(True)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(res.placement).tobytes())
        h.update(np.ascontiguousarray(res.unscheduled).tobytes())
        h.update(np.ascontiguousarray(res.used_cpu).tobytes())
        print(f"AB {os.path.basename(lib)} {name}: S={len(scen)} gen={st.kernel_generation} var={st.kernel_variant} wg={st.workgroup_size} "
              f"lds={st.lds_bytes} best_ms={min(ts[1:]):.3f} mean_ms={np.mean(ts[1:]):.3f} unsched_sum={int(res.unscheduled.sum())} "
              f"sha={h.hexdigest()[:16]} (build {t_build:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
