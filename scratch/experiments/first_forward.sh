# first forward on a fresh graph, pruned adjacency built in the first forward (prune_early_entries=1) or the second (=0) (GPU box)
set -u
cd "$GRAFT_REPO_ROOT"
for w in "$@"; do
  for o in 1 0; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --kernel-trace 0 --workload $w --opt prune_early_entries=$o > gpurun_out/ff.json 2> gpurun_out/ff.err || echo FAILED
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ff.json").read().strip().splitlines()[-1])
print("$w prune_early_entries=$o: first", round(d["first_forward_ms"],2), "second", round(d["second_forward_ms"],2), "third", round(d["third_forward_ms"],2), "fresh engine first", round(d.get("fresh_engine_first_forward_ms",0),2), "steady", round(d["ms_per_step"],3), "build", d["plan_build_ms"])
PY
  done
done
