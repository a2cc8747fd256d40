#!/bin/bash
# Development: A/B variants of the column-persistent kernel (pf_column.hpp) on ONE box.  Builds libpfamd_<tag>.so from the
# production objects (build/obj, from __graft_entry__.build()) + the float column unit recompiled with the given defines.
#   -DPFC_INVERTED=0  systematic resampling by binary search in the LDS cdf instead of the inverted grid
#   -DPFC_EXP=<mask>  ablations that price stages (results WRONG by construction): 1 no search (ancestor = floor(p N)),
#                     2 no Philox / Box-Muller, 4 no first-stage term in the APF weight, 8 no weighted moments
# Usage: tools/column_ablation.sh build <tag> <defines...>   |   tools/column_ablation.sh run <kbench configs...>   (GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/build/obj
if [ "$1" = build ]; then
  tag=$2; shift; shift
  cd $ROOT/pyfilter_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c pf_kernels.hip -DPF_TU_COLUMN_F32 "$@" -o $OBJ/pf_col_f32_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/pf_main.o $OBJ/pf_col_f64.o $OBJ/pf_f32d1_v4_m0.o $OBJ/pf_f32d1_v4_m1.o \
    $OBJ/pf_f32d1_v1_m0.o $OBJ/pf_f32d1_v1_m1.o $OBJ/pf_f32dn_m0.o $OBJ/pf_f32dn_m1.o $OBJ/pf_f64_m0.o $OBJ/pf_f64_m1.o $OBJ/pf_col_f32_$tag.o \
    -o $ROOT/pyfilter_amd/libpfamd_$tag.so
  ls -la $ROOT/pyfilter_amd/libpfamd_$tag.so
else
  shift
  for rep in 1 2; do
    echo "== production (pass $rep)"; KB_T=500 python $ROOT/tools/kbench.py "$@" 2>&1 | grep -v amdgpu.ids
    for lib in $ROOT/pyfilter_amd/libpfamd_?*.so; do
      echo "== $(basename $lib) (pass $rep)"; PF_AMD_LIB=$lib KB_T=500 python $ROOT/tools/kbench.py "$@" 2>&1 | grep -v amdgpu.ids
    done
  done
fi
