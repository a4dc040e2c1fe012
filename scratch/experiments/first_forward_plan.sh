#!/bin/bash
# first / second / third forward with the compact-table plan built inside the first forward (option
# compact_first_forward_entries) and without
set -u
OUT=gpurun_out/first_forward_plan.txt
: > $OUT
for wl in "$@"; do
  for v in 0 1; do
    timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --no-variants --steps 10 --warmup 2 --opt compact_first_forward_entries=$v > gpurun_out/_ff.json 2> gpurun_out/_ff.err || { echo "FAILED $wl $v" >> $OUT; tail -5 gpurun_out/_ff.err >> $OUT; exit 1; }
    python3 - "$wl" "$v" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ff.json").read().strip().splitlines()[-1])
print(sys.argv[1], "first_too", sys.argv[2], "ms", round(d["ms_per_step"], 3), "first", round(d["first_forward_ms"], 2), "second", round(d["second_forward_ms"], 2), "third", round(d["third_forward_ms"], 2), "build", d["plan_build_ms"])
PY
  done
done
cat $OUT
