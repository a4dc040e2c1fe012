#!/bin/bash
# power-law 1 M: where the giant rows begin (k_long_*'s longest row is a chain of d / 256 rounds: 16 K entries = 64 rounds)
cd "$GRAFT_REPO_ROOT"
for opts in "" "giant_row_threshold=8192" "giant_row_threshold=4096" "giant_row_threshold=2048" "giant_row_threshold_f16=8192" "giant_row_threshold_f16=4096"; do
  echo "== [$opts]"
  python scratch/experiments/first_trace.py powerlaw1m $opts 2>&1 | grep -E "^  forward [0-9]: " | head -4 | cut -c1-60
  python scratch/experiments/first_trace.py powerlaw1m $opts 2>&1 | grep -E "^  forward [0-9]: " | head -4 | cut -c1-60
done
