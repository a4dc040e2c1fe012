#!/bin/bash
# rocprofv3 kernel stats of bench.py on one workload: scratch/experiments/prof_workload.sh NAME [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=$1; shift
OUT=gpurun_out/prof_$W
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --workload $W --no-cpu-baseline --no-variants --no-host-path --no-workloads --kernel-trace 0 "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
echo "rc=$?"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && python3 tools/kstats.py "$f"
