#!/bin/bash
# Development tool: an A/B library next to the shipped one.  Rebuilds the named translation units of pf_kernels.hip with
# extra flags and links them with the production objects of the others (build/obj, from __graft_entry__.build()) into
# pyfilter_amd/libpfamd_<name>.so - load it with `PF_AMD_LIB=.../libpfamd_<name>.so` (tools/kbench.py, bench.py).
# Usage: tools/build_variant.sh <name> "<extra hipcc flags>" <unit> [<unit> ...]      units: f32d1_v4_m0 f32d1_v4_m1 f32d1_v1_m0
#        f32d1_v1_m1 f32dn_m0 f32dn_m1 f64_m0 f64_m1 main col_f32 col_f64 clu_f32 clu_f64
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2
OBJ=$ROOT/build/obj; OUT=$ROOT/build/$NAME; mkdir -p $OUT
NS=${PF_VARIANT_SLP:--fno-slp-vectorize}  # the float32 units build without the SLP vectorizer (__graft_entry__.NO_SLP)
declare -A FLAGS=(
  [main]="-DPF_TU_NO_F64 -DPF_TU_NO_F32DN -DPF_TU_NO_F32D1 $NS" [col_f32]="-DPF_TU_COLUMN_F32 $NS" [col_f64]="-DPF_TU_COLUMN_F64"
  [clu_f32]="-DPF_TU_CLUSTER_F32 $NS" [clu_f64]="-DPF_TU_CLUSTER_F64"
  [f32d1_v4_m0]="-DPF_TU_F32D1_ONLY -DPF_TU_VEC=4 -DPF_TU_MULTI=0 $NS" [f32d1_v4_m1]="-DPF_TU_F32D1_ONLY -DPF_TU_VEC=4 -DPF_TU_MULTI=1 $NS"
  [f32d1_v1_m0]="-DPF_TU_F32D1_ONLY -DPF_TU_VEC=1 -DPF_TU_MULTI=0 $NS" [f32d1_v1_m1]="-DPF_TU_F32D1_ONLY -DPF_TU_VEC=1 -DPF_TU_MULTI=1 $NS"
  [f32dn_m0]="-DPF_TU_F32DN_ONLY -DPF_TU_MULTI=0 $NS" [f32dn_m1]="-DPF_TU_F32DN_ONLY -DPF_TU_MULTI=1 $NS"
  [f64_m0]="-DPF_TU_F64_ONLY -DPF_TU_MULTI=0" [f64_m1]="-DPF_TU_F64_ONLY -DPF_TU_MULTI=1")
pids=()
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -c $ROOT/pyfilter_amd/csrc/pf_kernels.hip ${FLAGS[$u]} $EXTRA -o $OUT/pf_$u.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=()
for u in "${!FLAGS[@]}"; do
  if [ -f $OUT/pf_$u.o ] && [[ " $* " == *" $u "* ]]; then objs+=($OUT/pf_$u.o); else objs+=($OBJ/pf_$u.o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC "${objs[@]}" -o $ROOT/pyfilter_amd/libpfamd_$NAME.so
echo $ROOT/pyfilter_amd/libpfamd_$NAME.so
