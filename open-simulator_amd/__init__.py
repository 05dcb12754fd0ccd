"""simon-hip: MI355X-native batched capacity-planning core for open-simulator's hot path.

The directory is named `open-simulator_amd` (not importable as-is); import it as
`open_simulator_amd` (a shim package at the repo root points its __path__ here).
"""
from . import capi, fiterror, quantity, synth  # noqa: F401
from .capi import Context, Problem, SimonError, load_library  # noqa: F401

__all__ = ["capi", "fiterror", "quantity", "synth", "Context", "Problem", "SimonError", "load_library"]
