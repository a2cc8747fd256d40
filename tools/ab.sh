#!/bin/bash
# Development tool: same-box A/B of two builds of the library (tools/build_variant.sh): alternates them ROUNDS times over the
# given tools/kbench.py configurations.   Usage: tools/ab.sh <libA.so> <libB.so> <rounds> <config> [<config> ...]
A=$1; B=$2; R=$3; shift 3
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo "== $(basename $L) (round $r)"
    PF_AMD_LIB=$PWD/$L KB_T=${KB_T:-250} python tools/kbench.py "$@" 2>&1 | grep "us/step"
  done
done
