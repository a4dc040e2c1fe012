# A/B of the skewed-graph plans on the GPU box: bash scratch/experiments/mapped_ab.sh "opt=1" "opt=0" [workloads...]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
A=$1; B=$2; shift 2
for w in "${@:-rmat22 powerlaw1m}"; do
  for o in "$A" "$B"; do
    echo "== $w $o"
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-variants --workload $w --opt $o > gpurun_out/ab_${w}_${o}.json 2> gpurun_out/ab_${w}_${o}.err || { echo FAILED; tail -5 gpurun_out/ab_${w}_${o}.err; }
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_${w}_${o}.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["stage_ms"], "plan build", d["plan_build_ms"])
    for k,v in d["roofline"]["kernels"].items(): print("   ", k, round(v["ms_per_forward"],4), v["launches_per_forward"])
except Exception as ex: print("no json", ex)
PY
  done
done
