#!/bin/bash
# phase stamps inside the tile kernels on ER-100K / ER-300K (the probe library is built on the CPU box: make probe-lib below)
#   hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -DGNNVC_PHASE_PROBE=1 -shared -o scratch/experiments/libgnnvc_probe.so csrc/*.hip csrc/*.cpp
set -u
cd "$GRAFT_REPO_ROOT"
export GNNVC_LIBRARY=$PWD/scratch/experiments/libgnnvc_probe.so
for size in "100000 1000000" "300000 3000000"; do
  for tt in 1 0; do
    echo "=== n m = $size, table_tiles=$tt"
    timeout -k 10 200 python scratch/experiments/phase_probe.py $size table_tiles=$tt || exit 1
  done
done
