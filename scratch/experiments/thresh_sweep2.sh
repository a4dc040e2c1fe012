# long-row thresholds (list / 16-wide stages) with rows classed by the entries they have left (GPU box)
set -u
cd "$GRAFT_REPO_ROOT"
w=$1
for lt in 128 256 512; do
  for slt in 256 512 1024; do
    [ $slt -lt $lt ] && continue
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-variants --kernel-trace 0 --workload $w --long-threshold $lt --sorted-long-threshold $slt --giant-threshold ${GT:-16384} > gpurun_out/ts.json 2> gpurun_out/ts.err || echo FAILED
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ts.json").read().strip().splitlines()[-1])
print("$w long $lt sorted_long $slt:", round(d["ms_per_step"],3), [round(x,3) for x in d["stage_ms"]], "long", d["plan"]["long_rows"], "giant", d["plan"]["giant_rows"])
PY
  done
done
