#!/bin/bash
# F = 1 stage: rows from which degree on go the giant (parallel exact scan) way — first and steady forwards
for w in rmat22 powerlaw1m rmat24; do
  for t in 16384 4096 1024; do
    echo "== $w giant_row_threshold=$t"
    timeout -k 10 300 python scratch/experiments/first_trace.py $w giant_row_threshold=$t giant_row_threshold_f16=65536 2>&1 | grep -v "^      k_\|^      (k_\|amdgpu.ids\|logits differ" | sed -n 2,5p | cut -c1-220
  done
done
