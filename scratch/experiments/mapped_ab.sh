# A/B of the skewed-graph compact plan on the GPU box: bash scratch/experiments/mapped_ab.sh
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in rmat22 powerlaw1m; do
  for o in "compact_skewed=1" "compact_skewed=0"; do
    echo "== $w $o"
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-variants --workload $w --opt $o > gpurun_out/ab_${w}_${o}.json 2> gpurun_out/ab_${w}_${o}.err || { echo FAILED; tail -5 gpurun_out/ab_${w}_${o}.err; }
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_${w}_${o}.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d.get("plans",{}).get("compact_gather_last"), d.get("parity"))
    for k in d.get("kernels",[])[:12]: print("   ",k)
except Exception as ex: print("no json", ex)
PY
  done
done
