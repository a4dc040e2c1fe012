#!/bin/bash
# rocprofv3 kernel stats of a small graph's forwards: scratch/experiments/small_prof.sh N M [deep rows ...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; M=$2; shift 2
for d in "$@"; do
  OUT=gpurun_out/small_prof_${N}_$d
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 scratch/experiments/small_one.py $N $M $d > "$OUT/log.txt" 2>&1
  echo "deep=$d rc=$?"
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 tools/kstats.py "$f" | grep -E "k_stage|k_zero"
done
