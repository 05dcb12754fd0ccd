#!/usr/bin/env python3
"""Writes tests/golden/gosort_vectors.json: flag vectors (Less(i, j) = flag[i], the queues of pkg/algo/affinity.go:21-23 and
toleration.go:19-21) of lengths 13 ... 60 with the permutation go1.18's sort.Sort leaves -- computed by oracle/gosort_check (the C
restatement of the published go1.18 src/sort/sort.go, independent of open-simulator_amd/gosort.py).  Lengths chosen for the branches:
13 (first length that reaches doPivot), 25 / 29 (two levels), 41 / 50 (Tukey ninther, hi - lo > 40), 60 with few / many flags (the
duplicate-protection loop).  Run from the repo root after `make -C oracle`."""
import json
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnd = random.Random(20260924)
vectors = []
for n, p in [(13, 0.3), (13, 0.6), (25, 0.4), (29, 0.2), (41, 0.5), (50, 0.3), (50, 0.7), (60, 0.05), (60, 0.9)]:
    flags = "".join("1" if rnd.random() < p else "0" for _ in range(n))
    out = subprocess.run([os.path.join(ROOT, "oracle", "gosort_check"), flags], capture_output=True, text=True, check=True).stdout.split()
    vectors.append({"flags": flags, "order": [int(x) for x in out]})
with open(os.path.join(ROOT, "tests", "golden", "gosort_vectors.json"), "w") as f:
    json.dump({"source": "oracle/gosort_check.c (C restatement of go1.18 src/sort/sort.go), NOT a Go binary", "vectors": vectors}, f, indent=1)
print(len(vectors), "vectors")
