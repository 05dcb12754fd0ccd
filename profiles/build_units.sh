#!/bin/bash
# Build a second copy of the library in which only the NAMED translation units are recompiled with extra defines (fast kernel A/B runs):
#   bash profiles/build_units.sh x1 "simon_table_rest simon_table_restlds" -DSIMON_EXP=1   -> open-simulator_amd/csrc/libsimon_hip_x1.so
# every other unit is the product object (python __graft_entry__.py builds them).  Use with SIMON_HIP_LIB (profiles/gpu_r6a.sh takes the suffix).
set -e
NAME=$1; UNITS=$2; shift 2
C=open-simulator_amd/csrc
OBJS=""
for F in simon_hip simon_group simon_narrow simon_fast simon_table simon_table_spread simon_table_spread2 simon_table_team4 simon_table_team4z simon_table_rest simon_table_rest2 simon_table_rs simon_table_rsz simon_table_cls4 simon_table_lds simon_table_restlds simon_wide simon_wide_local simon_wide_explain; do
  case " $UNITS " in
    *" $F "*) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c -o $C/${F}_$NAME.o $C/$F.hip & OBJS="$OBJS $C/${F}_$NAME.o" ;;
    *) OBJS="$OBJS $C/$F.o" ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libsimon_hip_$NAME.so $OBJS
echo built $C/libsimon_hip_$NAME.so
