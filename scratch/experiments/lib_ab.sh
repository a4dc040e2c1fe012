#!/bin/bash
# steady / first forward with alternative builds of the library (compile-time variants): lib_ab.sh lib1.so lib2.so ... -- w1 w2 ...
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
for w in "$@"; do
  for l in "" "${libs[@]}"; do
    echo "== $w [${l:-in-tree}]"
    GNNVC_LIBRARY=$l timeout -k 10 300 python scratch/experiments/first_trace.py $w 2>&1 | grep "^  forward [0-3]: " | sed -n 1,4p | cut -c1-24 | tr '\n' ' '; echo
  done
done
