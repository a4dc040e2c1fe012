#!/bin/bash
# first forward with the filtered gather (default) and without, on the skewed workloads
for w in rmat20 powerlaw1m rmat24; do
  for o in "" "filter_zero_rows=0"; do
    timeout -k 10 300 python scratch/experiments/first_trace.py $w $o 2>&1 | grep -v "^      k_\|^      (k_\|amdgpu.ids" | sed -n 1,8p | cut -c1-150
  done
done
