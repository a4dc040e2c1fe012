# rows the skewed-graph LDS-table plan takes (GPU box): bash scratch/experiments/lt_sweep.sh workload...
set -u
cd "$GRAFT_REPO_ROOT"
for w in "$@"; do
  for p in 512 1024 2048 4096 16384; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-variants --workload $w --opt lds_table_skewed_rows=$p > gpurun_out/lt.json 2> gpurun_out/lt.err || echo FAILED
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/lt.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("$w rows<$p:", round(d["ms_per_step"],3), [round(x,3) for x in d["stage_ms"]], "k_lt_agg", round(k.get("k_lt_agg",{}).get("ms_per_forward",0),3), "build", d["plan_build_ms"])
PY
  done
done
