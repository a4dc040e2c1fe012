#!/bin/bash
# long-row thresholds under the two-queue arrangement: first / steady forward per setting
for w in rmat20 rmat22 rmat24; do
  for o in "" "sorted_long_row_threshold=2048 long_row_threshold=512" "sorted_long_row_threshold=4096 long_row_threshold=512" "sorted_long_row_threshold=2048 long_row_threshold=1024" "prune_heavy_entries=1000000000"; do
    echo "== $w [$o]"
    timeout -k 10 300 python scratch/experiments/first_trace.py $w $o 2>&1 | grep "^  forward [0-3]: " | sed -n 1,4p | cut -c1-24 | tr '\n' ' '; echo
  done
done
