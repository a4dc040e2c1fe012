#!/bin/bash
# What sits between a small forward's kernels: the events of a forward (option forward_timing: 0 = none, the default; 2 = one per
# stage, what every forward recorded before)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for opts in "" "forward_timing=2"; do
    echo "== [$opts]"
    python scratch/experiments/small_sizes.py $opts 2>&1 | grep -E "^n (2000|20000|41000|60000|100000|140000|300000) " | cut -c1-260
  done
done
