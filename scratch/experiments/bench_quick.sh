#!/bin/bash
# ms_per_step / first_forward_ms / plain_forward_ms of the named workloads (no CPU legs): bench_quick.sh TAG w1 w2 ...
TAG=$1; shift
for w in "$@"; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-host-path --no-workloads --kernel-trace 0 > gpurun_out/bq_${TAG}_$w.json 2> gpurun_out/bq_${TAG}_$w.log
  python - "$w" gpurun_out/bq_${TAG}_$w.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    so = d.get("score_once", {})
    print(sys.argv[1], "ms_per_step", d["ms_per_step"], "first", so.get("first_forward_ms"), "attach+first", so.get("attach_plus_first_forward_ms"), "plain", d.get("plain_forward_ms"), "frac", d["roofline"]["frac"])
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
