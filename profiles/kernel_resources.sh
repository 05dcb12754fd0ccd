#!/bin/bash
# Print LDS / scratch / SGPR / VGPR of every kernel of one .hip file (cross-compile, no GPU needed).
# usage: bash profiles/kernel_resources.sh open-simulator_amd/csrc/simon_wide.hip
set -e
SRC=$(realpath "$1")
TMP=$(mktemp -d)
( cd "$TMP" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c "$SRC" -o out.o --save-temps >/dev/null 2>&1 )
python3 - "$TMP" <<'PY'
import re, sys, glob
s = open(glob.glob(sys.argv[1] + '/*gfx950.s')[0]).read()
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', s, re.S):
    print(f"{m.group(2)[-48:]:48s} lds {m.group(1):>6s} scratch {m.group(3):>5s} sgpr {m.group(4):>4s} vgpr {m.group(5):>4s}")
PY
rm -rf "$TMP"
