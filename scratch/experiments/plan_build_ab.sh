#!/bin/bash
# plan_build_ms / first forwards with the staged scatter on and off (GNNVC_LT_STAGE=0: direct stores)
set -u
OUT=gpurun_out/plan_build_ab.txt
: > $OUT
for wl in er10m rmat22 er3m; do
  for st in default 0; do
    if [ "$st" = default ]; then unset GNNVC_LT_STAGE; else export GNNVC_LT_STAGE=$st; fi
    timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --no-variants --steps 10 --warmup 2 > gpurun_out/_pb.json 2> gpurun_out/_pb.err || { echo "FAILED $wl $st" >> $OUT; tail -5 gpurun_out/_pb.err >> $OUT; exit 1; }
    python3 - "$wl" "$st" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_pb.json").read().strip().splitlines()[-1])
print(sys.argv[1], "stage", sys.argv[2], "ms", round(d["ms_per_step"], 3), "first", round(d["first_forward_ms"], 2), "second", round(d["second_forward_ms"], 2), "third", round(d["third_forward_ms"], 2), "build", d["plan_build_ms"])
PY
  done
done
cat $OUT
