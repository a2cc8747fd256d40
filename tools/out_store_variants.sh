#!/bin/bash
# DEVELOPMENT: builds libpfamd_os{1,2,3}.so - the float scalar-state VEC = 4 step kernels (both tile geometries) with the
# output stores non-temporal / sc1 / sc0 sc1 (pf_device.hpp: PF_OUT_STORE), everything else from build/obj - for
#   PF_AMD_LIB=pyfilter_amd/libpfamd_os2.so python tools/kbench.py apf_lgo_1m ...
set -e
cd "$(dirname "$0")/../pyfilter_amd/csrc"
OBJ=../../build/obj
for k in 1 2 3; do
  for m in 0 1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c pf_kernels.hip -DPF_OUT_STORE=$k -DPF_TU_F32D1_ONLY -DPF_TU_VEC=4 -DPF_TU_MULTI=$m -o $OBJ/pf_f32d1_v4_m${m}_os$k.o &
  done
done
wait
for k in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/pf_main.o $OBJ/pf_f32d1_v4_m0_os$k.o $OBJ/pf_f32d1_v4_m1_os$k.o $OBJ/pf_f32d1_v1_m0.o $OBJ/pf_f32d1_v1_m1.o \
     $OBJ/pf_f32dn_m0.o $OBJ/pf_f32dn_m1.o $OBJ/pf_f64_m0.o $OBJ/pf_f64_m1.o -o ../libpfamd_os$k.so
done
ls -la ../libpfamd_os*.so
