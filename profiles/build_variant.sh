#!/bin/bash
# Build a second copy of the library with extra defines for simon_wide.hip (kernel A/B runs, profile builds):
#   bash profiles/build_variant.sh prof -DSIMON_WIDE_PROFILE        -> open-simulator_amd/csrc/libsimon_hip_prof.so
# use it with SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_prof.so (capi.library_path).
set -e
NAME=$1; shift
C=open-simulator_amd/csrc
python __graft_entry__.py > /dev/null          # the regular objects
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c -o $C/simon_wide_$NAME.o $C/simon_wide.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libsimon_hip_$NAME.so $C/simon_hip.o $C/simon_narrow.o $C/simon_fast.o $C/simon_cache.o $C/simon_wide_$NAME.o
echo built $C/libsimon_hip_$NAME.so
