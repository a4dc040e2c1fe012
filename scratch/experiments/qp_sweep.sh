#!/bin/bash
# forward time of the workloads with 0..8 other engines alive in the process (where the runtime places the engine's side queue)
for w in powerlaw1m rmat22 rmat20 er3m; do
  echo "== $w"
  timeout -k 10 300 python scratch/experiments/queue_pressure.py $w 2>&1 | grep -v amdgpu.ids
done
echo "== default bench"
bash scratch/experiments/bench_wl.sh d X=1
