#!/bin/bash
# HIP API durations of a short bench run (where do the one-time milliseconds of the first plan use go?)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/api_trace
timeout -k 10 400 rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/api_trace -- python3 bench.py "$@" > gpurun_out/api_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = [x for x in glob.glob("gpurun_out/api_trace/**/*hip_api_stats.csv", recursive=True)][0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:16]:
    print(r["Name"], r["Calls"], "total_ms", round(int(r["TotalDurationNs"]) / 1e6, 2), "max_ms", round(int(r["MaxNs"]) / 1e6, 3))
PY
grep -o '"first_forward_ms": [0-9.]*\|"second_forward_ms": [0-9.]*' gpurun_out/api_trace.log
