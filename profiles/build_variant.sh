#!/bin/bash
# Build a second copy of the library with extra defines (kernel A/B runs, profile builds):
#   bash profiles/build_variant.sh prof -DSIMON_WIDE_PROFILE        -> open-simulator_amd/csrc/libsimon_hip_prof.so
#   bash profiles/build_variant.sh tprof -DSIMON_TABLE_PROFILE      -> ... with the phase profiler of simon_table.hip
# use it with SIMON_HIP_LIB=$PWD/open-simulator_amd/csrc/libsimon_hip_<name>.so (capi.library_path).
set -e
NAME=$1; shift
C=open-simulator_amd/csrc
python __graft_entry__.py > /dev/null          # the regular objects
OBJS=""
for F in simon_hip simon_group simon_narrow simon_fast simon_table simon_table_spread simon_table_spread2 simon_table_team4 simon_table_team4z simon_table_rest simon_table_rest2 simon_table_rs simon_table_rsz simon_table_cls4 simon_table_lds simon_table_restlds simon_wide simon_wide_local simon_wide_explain; do
  if [ "${F#simon_wide}" != "$F" ] || [ "${F#simon_table}" != "$F" ] || [ "$F" = simon_hip ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c -o $C/${F}_$NAME.o $C/$F.hip &
    OBJS="$OBJS $C/${F}_$NAME.o"
  else
    OBJS="$OBJS $C/$F.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libsimon_hip_$NAME.so $OBJS
echo built $C/libsimon_hip_$NAME.so
