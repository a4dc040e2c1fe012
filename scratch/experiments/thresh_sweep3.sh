set -u
cd "$GRAFT_REPO_ROOT"
w=$1
for slt in 1024 2048 4096; do
  for gt in 16384 65536; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-variants --kernel-trace 0 --workload $w --long-threshold 512 --sorted-long-threshold $slt --giant-threshold $gt > gpurun_out/ts.json 2> gpurun_out/ts.err || echo FAILED
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ts.json").read().strip().splitlines()[-1])
print("$w sorted_long $slt giant $gt:", round(d["ms_per_step"],3), [round(x,3) for x in d["stage_ms"]], "long", d["plan"]["long_rows"], "giant", d["plan"]["giant_rows"])
PY
  done
done
