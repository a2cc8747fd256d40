#!/bin/bash
# Development tool: same-box comparison of SEVERAL builds of the library (tools/build_variant.sh): cycles through them ROUNDS
# times over the given tools/kbench.py configurations.  Usage: tools/ab_many.sh <rounds> "<lib1.so lib2.so ...>" <config> [...]
R=$1; LIBS=$2; shift 2
for r in $(seq 1 $R); do
  for L in $LIBS; do
    echo "== $(basename $L) (round $r)"
    PF_AMD_LIB=$PWD/$L KB_T=${KB_T:-250} python tools/kbench.py "$@" 2>&1 | grep "us/step"
  done
done
