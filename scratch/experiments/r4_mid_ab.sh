#!/bin/bash
# ER-100K / ER-300K forward: wide F = 1 tiles feeding the table tiles (default) vs the plain F = 1 kernel vs wide tiles everywhere
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for opts in "" "wide_tiles_max_n=49152" "table_tiles=0 wide_tiles_max_n_f16=400000" "wide_tiles=0"; do
    echo "== [$opts]"
    python scratch/experiments/small_sizes.py $opts 2>&1 | grep -E "^n (41000|100000|300000) " | cut -c1-215
  done
done
