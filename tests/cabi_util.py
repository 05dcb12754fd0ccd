"""Builds tests/cabi/cabi_smoke.c (the plain-C consumer of the C-ABI) with gcc."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cabi_smoke(out_dir, name="cabi_smoke"):
    """name: cabi_smoke (SIMONFX1 / FX2 fixtures) or cabi_terms (SIMONFX3: every optional array by name, Open-Local error sizes)."""
    exe = os.path.join(str(out_dir), name)
    lib_dir = os.path.join(ROOT, "open-simulator_amd", "csrc")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", name + ".c"), "-L", lib_dir, "-lsimon_hip",
                           f"-Wl,-rpath,{lib_dir}", "-lpthread", "-o", exe])
    return exe
