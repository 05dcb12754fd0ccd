#!/usr/bin/env python3
"""Wide-kernel step-time experiments (GPU box): fixed per-step overhead vs per-node work.
usage: python profiles/exp_wide.py  -> one line per variant: n, P, S, T, ms, us/step"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_simulator_amd import capi, synth

def run(tag, prob, scen, orders, env):
    for k, v in env.items():
        os.environ[k] = v
    with capi.Context(0) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        ctx.run_loaded(want_placement=True)
        st = ctx.stats()
    for k in env:
        os.environ.pop(k, None)
    P = prob.n_pods
    print(f"{tag:42s} n~{int(scen[:,0].mean()):5d} P={P:6d} S={len(scen):4d} T={st.workgroup_size:4d} variant={st.kernel_variant} "
          f"kernel_ms={st.kernel_ms:9.2f} us/step={st.kernel_ms*1e3/P:7.2f}", flush=True)

which = sys.argv[1:] or ["small", "plain5k", "cfg5"]
if "small" in which:
    prob, scen, orders = synth.config3(n_counts=64, n_orders=4, n_pods=10000, n_het=488)
    for T in ("64", "256", "512", "1024"):
        run("cpu+mem only, forced wide", prob, scen, orders, {"SIMON_FORCE_WIDE": "1", "SIMON_WG": T})
if "plain5k" in which:
    prob, scen, orders = synth.config3(n_counts=64, n_orders=4, n_pods=20000, n_het=4000)
    for T in ("512", "1024"):
        run("cpu+mem only, 4k nodes, forced wide", prob, scen, orders, {"SIMON_FORCE_WIDE": "1", "SIMON_WG": T})
if "cfg5" in which:
    prob, scen, orders = synth.config5(n_pods=20000)
    for T in ("512", "1024"):
        run("config5 (20k pods)", prob, scen, orders, {"SIMON_WG": T})
    p2 = synth.config5(n_pods=20000)[0]
    p2.gpu_mem = np.zeros_like(p2.gpu_mem); p2.pod_gpu_cnt = np.zeros_like(p2.pod_gpu_cnt)
    run("config5 without GPU pods", p2, scen, orders, {"SIMON_WG": "512"})
    p3 = synth.config5(n_pods=20000, n_groups=0)[0]
    run("config5 without anti-affinity groups", p3, scen, orders, {"SIMON_WG": "512"})
