#!/bin/bash
# kernel durations of the plan builders (rocprofv3 kernel trace of a short bench run); $1 = workload, rest = env assignments
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
wl=${1:-er10m}; shift
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pb_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb_trace -- python3 bench.py --workload $wl --no-cpu-baseline --no-variants --steps 3 --warmup 1 > gpurun_out/pb_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pb_trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_lt_count", "k_lt_scatter", "k_lt_steps", "k_ltw_steps", "k_lt_bytes", "fill", "Fill", "memset")):
        i = n.find("k_lt"); short = n[i:i + 12] if i >= 0 else n[:40]
        print(short, r["Calls"], "total_us", int(r["TotalDurationNs"]) // 1000, "min_us", int(r["MinNs"]) // 1000, "max_us", int(r["MaxNs"]) // 1000)
PY
grep -o '"plan_build_ms": [0-9.]*' gpurun_out/pb_trace.log
