#!/bin/bash
# giant-row thresholds under the two-queue arrangement with quarter rounds: first / steady forward per setting
for w in rmat22 powerlaw1m rmat20 rmat24; do
  for o in "" "giant_row_threshold_f16=16384" "giant_row_threshold_f16=32768" "giant_row_threshold_f16=131072" "giant_row_threshold_f16=1000000" "giant_row_threshold=65536 giant_row_threshold_f16=65536" "giant_row_threshold=1000000 giant_row_threshold_f16=1000000"; do
    echo "== $w [$o]"
    timeout -k 10 300 python scratch/experiments/first_trace.py $w $o 2>&1 | grep "^  forward [0-3]: " | sed -n 1,4p | cut -c1-24 | tr '\n' ' '; echo
  done
done
