#!/bin/bash
# power-law 1 M: the long rows on a third queue beside the tile kernel and the giant chain (GNNVC_EXP_THIRD_QUEUE) vs ahead of the tile kernel
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for third in 0 1; do
    if [ $third = 1 ]; then export GNNVC_EXP_THIRD_QUEUE=1; else unset GNNVC_EXP_THIRD_QUEUE; fi
    timeout -k 10 200 python bench.py --workload powerlaw1m --steps 50 --warmup 10 --no-cpu-baseline --no-workloads --no-host-path > gpurun_out/r4_q3_${third}_$rep.json 2> gpurun_out/r4_q3_${third}_$rep.err || exit 1
    python - gpurun_out/r4_q3_${third}_$rep.json $third <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("third queue", sys.argv[2], "ms_per_step", d["ms_per_step"], "first", d.get("first_forward_ms"))
PY
    grep exp3 gpurun_out/r4_q3_${third}_$rep.err | head -3
  done
done
