"""Bit-exact restatements of the Go standard-library float routines the hot path depends on.

The scheduler's PodTopologySpread score multiplies pod counts by math.Log(float64(size + 2))
(vendor/k8s.io/kubernetes/pkg/scheduler/framework/plugins/podtopologyspread/scoring.go:279-281).  math.Log is Go
stdlib (go.mod pins go 1.18, Dockerfile golang:1.18.3) and NOT under the reference tree; its portable
implementation (src/math/log.go, a port of FreeBSD e_log.c) is a fixed sequence of IEEE binary64 operations, which
CPython floats reproduce exactly (no FMA contraction in either).  In the Go host this module does not exist: the
shim fills simon_class_tables.spread_log with its own math.Log.
"""
from __future__ import annotations

import math

_LN2_HI = float.fromhex("0x1.62e42fee00000p-1")    # 6.93147180369123816490e-01
_LN2_LO = float.fromhex("0x1.a39ef35793c76p-33")   # 1.90821492927058770002e-10
_L1 = float.fromhex("0x1.5555555555593p-1")        # 6.666666666666735130e-01
_L2 = float.fromhex("0x1.999999997fa04p-2")        # 3.999999999940941908e-01
_L3 = float.fromhex("0x1.2492494229359p-2")        # 2.857142874366239149e-01
_L4 = float.fromhex("0x1.c71c51d8e78afp-3")        # 2.222219843214978396e-01
_L5 = float.fromhex("0x1.7466496cb03dep-3")        # 1.818357216161805012e-01
_L6 = float.fromhex("0x1.39a09d078c69fp-3")        # 1.531383769920937332e-01
_L7 = float.fromhex("0x1.2f112df3e5244p-3")        # 1.479819860511658591e-01
_SQRT2_2 = math.sqrt(2.0) / 2                      # Sqrt2/2 (constant-folded by the Go compiler to the same double)


def go_log(x: float) -> float:
    """math.Log for finite x > 0 (Go 1.18 src/math/log.go:80-128)."""
    if not (x > 0) or math.isinf(x):
        raise ValueError("go_log: finite positive argument expected")
    f1, ki = math.frexp(x)
    if f1 < _SQRT2_2:
        f1 *= 2
        ki -= 1
    f = f1 - 1
    k = float(ki)
    s = f / (2 + f)
    s2 = s * s
    s4 = s2 * s2
    t1 = s2 * (_L1 + s4 * (_L3 + s4 * (_L5 + s4 * _L7)))
    t2 = s4 * (_L2 + s4 * (_L4 + s4 * _L6))
    r = t1 + t2
    hfsq = 0.5 * f * f
    return k * _LN2_HI - ((hfsq - (s * (hfsq + r) + k * _LN2_LO)) - f)


def spread_log_table(n_nodes: int):
    """spread_log[i] = math.Log(float64(i + 2)), i = 0..n_nodes (topologyNormalizingWeight, scoring.go:279-281)."""
    import numpy as np
    return np.array([go_log(float(i + 2)) for i in range(n_nodes + 1)], np.float64)
