# the workload table of DESIGN.md §5 (GPU box): bash scratch/experiments/bench_workloads.sh TAG workload...
set -u
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
for w in "$@"; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --workload $w > gpurun_out/${tag}_$w.json 2> gpurun_out/${tag}_$w.err || echo "FAILED $w"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$w.json").read().strip().splitlines()[-1])
    print("$w", "ms", round(d["ms_per_step"],3), "Gedges/s", round(d["value"]/1e9,2), "frac", round(d["roofline"]["frac"],3), "first", round(d["first_forward_ms"],2),
          "plain", round(d.get("plain_forward_ms") or 0,2), "build", d["plan_build_ms"], "stages", [round(x,3) for x in d["stage_ms"]])
    print("   ", d["plan"].get("pruned_adjacency"), "long", d["plan"]["long_rows"], "giant", d["plan"]["giant_rows"], d["plan"]["giant_entries"])
except Exception as ex: print("no json $w", ex)
PY
done
