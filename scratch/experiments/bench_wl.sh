#!/bin/bash
# the default bench's `workloads` block (and headline) under an environment: bench_wl.sh TAG [VAR=VALUE ...]
TAG=$1; shift
env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bwl_$TAG.json 2> gpurun_out/bwl_$TAG.log
python - gpurun_out/bwl_$TAG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("er10m", d["ms_per_step"], "first", d["score_once"]["first_forward_ms"])
for w, v in d["workloads"].items():
    print(w, {k: round(v[k], 3) for k in ("ms_per_step", "first_forward_ms", "attach_ms") if k in v})
PY
