#!/bin/bash
# Development tool: same-box comparison of several builds of the library: ROUNDS alternations over the given kbench configs.
# Usage: LIBS="a.so b.so ..." tools/abn.sh <rounds> <config> [<config> ...]
R=$1; shift
for r in $(seq 1 $R); do
  for L in $LIBS; do
    echo "== $(basename $L) (round $r)"
    PF_AMD_LIB=$PWD/$L KB_T=${KB_T:-250} python tools/kbench.py "$@" 2>&1 | grep "us/step"
  done
done
