"""Builds tests/cabi/cabi_smoke.c (the plain-C consumer of the C-ABI) with gcc."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cabi_smoke(out_dir):
    exe = os.path.join(str(out_dir), "cabi_smoke")
    lib_dir = os.path.join(ROOT, "open-simulator_amd", "csrc")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-L", lib_dir, "-lsimon_hip",
                           f"-Wl,-rpath,{lib_dir}", "-lpthread", "-o", exe])
    return exe
