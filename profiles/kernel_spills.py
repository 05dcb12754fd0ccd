#!/usr/bin/env python3
"""Registers, scratch and spill counts of every kernel of one .hip file, from the metadata hipcc --save-temps leaves in the
gfx950 assembly (cross-compiles: no GPU needed).  usage: python profiles/kernel_spills.py <file.hip> [filter-substring]"""
import glob, os, re, subprocess, sys, tempfile

src = os.path.realpath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", src, "-o", "out.o",
                "--save-temps"] + sys.argv[3:], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
s = open(glob.glob(tmp + "/*gfx950.s")[0]).read()
pat = (r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
       r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)")
for m in re.finditer(pat, s, re.S):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    t = re.search(r"(\w+_kernel)<(.*?)>", name)
    short = f"{t.group(1)}<{t.group(2)}>" if t else name[:60]
    if flt in short:
        print(f"{short:90s} scratch {m.group(2):>5s} sgpr {m.group(3):>4s} (spilled {m.group(4):>4s}) vgpr {m.group(5):>4s} (spilled {m.group(6):>3s})")
