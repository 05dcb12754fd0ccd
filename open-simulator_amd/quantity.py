"""Host-side restatement of the slice of apimachinery's resource.Quantity the hot path depends on.

Reference: vendor/k8s.io/apimachinery/pkg/api/resource/{quantity.go,amount.go,suffix.go}.  In the
Go host this module does not exist -- the shim calls the real Quantity methods; the Python mirror
needs it to build the same int64 inputs (Value/MilliValue are CEIL, quantity.go:729-750) and the
same SimonPlugin raw-score table (AsApproximateFloat64, quantity.go:449-474) from YAML-style
strings.

Only the int64Amount fast path is modelled (|value| < 2^63, scale >= -9); quantities that would
fall into the inf.Dec slow path raise, and the caller must route that workload to the Go path.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass

DECIMAL_SI = "DecimalSI"
BINARY_SI = "BinarySI"
DECIMAL_EXPONENT = "DecimalExponent"

_DEC_SUFFIX = {"n": -9, "u": -6, "m": -3, "": 0, "k": 3, "M": 6, "G": 9, "T": 12, "P": 15, "E": 18}
_BIN_SUFFIX = {"Ki": 10, "Mi": 20, "Gi": 30, "Ti": 40, "Pi": 50, "Ei": 60}
_RE = re.compile(r"^([+-]?)(\d*)(?:\.(\d*))?((?:[eE][+-]?\d+)|[a-zA-Z]*)$")

_POW10TAB = [float(f"1e{i}") for i in range(23)]


def go_pow10(n: int) -> float:
    """math.Pow10 of the Go 1.18 stdlib for |n| <= 22: table lookup, resp. 1/table."""
    if n >= 0:
        return _POW10TAB[n] if n <= 22 else math.pow(10.0, n)
    return 1.0 / _POW10TAB[-n] if -n <= 22 else math.pow(10.0, n)


@dataclass(frozen=True)
class Quantity:
    value: int = 0          # int64Amount.value (unscaled)
    scale: int = 0          # int64Amount.scale (decimal exponent)
    format: str = DECIMAL_SI

    # -- quantity.go:729-750 ---------------------------------------------------------------
    def scaled_value(self, scale: int) -> int:
        """ScaledValue: value * 10^(self.scale - scale), rounded UP (away from zero for +)."""
        d = self.scale - scale
        if d >= 0:
            return self.value * 10 ** d
        q, r = divmod(self.value, 10 ** (-d))
        return q + (1 if r else 0)  # divmod floors; +1 on remainder = ceil (negativeScaleInt64, scale_int.go)

    def int_value(self) -> int:
        return self.scaled_value(0)

    def milli_value(self) -> int:
        return self.scaled_value(-3)

    def is_zero(self) -> bool:
        return self.value == 0

    # -- amount.go:160-202 -----------------------------------------------------------------
    def add(self, b: "Quantity") -> "Quantity":
        fmt = b.format if self.is_zero() else self.format  # quantity.go:563-575
        if b.value == 0:
            return Quantity(self.value, self.scale, fmt)
        if self.value == 0:
            return Quantity(b.value, b.scale, fmt)
        if self.scale == b.scale:
            v, s = self.value + b.value, self.scale
        elif self.scale > b.scale:
            v, s = self.value * 10 ** (self.scale - b.scale) + b.value, b.scale
        else:
            v, s = self.value + b.value * 10 ** (b.scale - self.scale), self.scale
        if not -(1 << 63) <= v < (1 << 63):
            raise OverflowError("Quantity leaves the int64Amount fast path")
        return Quantity(v, s, fmt)

    def sub(self, b: "Quantity") -> "Quantity":
        return self.add(Quantity(-b.value, b.scale, b.format))

    # -- quantity.go:449-474 ---------------------------------------------------------------
    def as_approximate_float64(self) -> float:
        base = float(self.value)
        if self.scale == 0:
            return base
        if self.format in (DECIMAL_SI, DECIMAL_EXPONENT):
            return base * go_pow10(self.scale)
        if 0 < self.scale < 7:
            return base * float(1 << (self.scale * 10))
        return base * math.pow(2.0, float(self.scale * 10))


ZERO = Quantity()


_PARSED: dict = {}          # text -> Quantity (frozen: shared freely); a cluster repeats a few dozen spellings tens of thousands of times


def parse_quantity(s) -> Quantity:
    """ParseQuantity fast path (quantity.go:262-330)."""
    if isinstance(s, Quantity):
        return s
    if isinstance(s, (int,)):
        return Quantity(int(s), 0, DECIMAL_SI)
    if isinstance(s, str):
        q = _PARSED.get(s)
        if q is None:
            q = _parse_text(s)
            if len(_PARSED) < 65536:
                _PARSED[s] = q
        return q
    return _parse_text(str(s))


def _parse_text(s: str) -> Quantity:
    s = s.strip()
    m = _RE.match(s)
    if not m or (m.group(2) == "" and not m.group(3)):
        raise ValueError(f"unparseable quantity {s!r}")
    sign, num, denom, suf = m.group(1), m.group(2) or "0", m.group(3) or "", m.group(4)
    num = num.lstrip("0") or "0"
    if suf in _BIN_SUFFIX:
        if denom.strip("0"):
            raise ValueError(f"{s!r}: fractional binary-SI quantity needs the inf.Dec path")
        v = int(num) << _BIN_SUFFIX[suf]
        fmt, scale = BINARY_SI, 0
    elif suf in _DEC_SUFFIX:
        scale = _DEC_SUFFIX[suf] - len(denom)
        v = int(num + denom)
        fmt = DECIMAL_SI
    elif suf[:1] in ("e", "E"):
        scale = int(suf[1:]) - len(denom)
        v = int(num + denom)
        fmt = DECIMAL_EXPONENT
    else:
        raise ValueError(f"bad suffix in {s!r}")
    if scale < -9:
        raise ValueError(f"{s!r}: sub-nano precision needs the inf.Dec path")
    if sign == "-":
        v = -v
    if not -(1 << 63) <= v < (1 << 63):
        raise OverflowError(f"{s!r} leaves the int64Amount fast path")
    return Quantity(v, scale, fmt)


def share(alloc: float, total: float) -> float:
    """algo.Share, pkg/algo/greed.go:70-83."""
    if total == 0:
        return 0.0 if alloc == 0 else 1.0
    return alloc / total


def simon_raw_score(pod_requests: dict, node_allocatable: dict) -> int:
    """SimonPlugin.Score == GpuSharePlugin.Score raw value (pkg/simulator/plugin/simon.go:45-68).

    pod_requests: resource name -> Quantity (PodRequestsAndLimits sum, kubectl resource.go:34-58);
    node_allocatable: resource name -> Quantity (node.Status.Allocatable).
    """
    if len(pod_requests) == 0:
        return 100
    res = 0.0
    for name, alloc in node_allocatable.items():
        req = pod_requests.get(name, ZERO)
        avail = alloc.sub(req)
        sh = share(req.as_approximate_float64(), avail.as_approximate_float64())
        if sh > res:
            res = sh
    return int(float(100 - 0) * res)  # int64() truncates toward zero; res >= 0
