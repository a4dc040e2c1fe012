# PMC pass over bench.py on a workload (GPU box): bash scratch/experiments/pmc_workload.sh rmat22 "<counters>" [bench args]
set -u
w=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${TAG:-pmc_$w}
rm -rf gpurun_out/$tag
rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-variants --kernel-trace 0 --workload $w --steps 2 --warmup 1 "$@" > gpurun_out/$tag.log 2>&1
cp $(find gpurun_out/$tag -name "*counter_collection.csv" | head -1) gpurun_out/$tag.csv
python3 - <<PY
import csv, collections
last = {}
for r in csv.DictReader(open("gpurun_out/$tag.csv")):
    n = r["Kernel_Name"]
    if "gnnvc" not in n: continue
    k = n[n.index("k_"):].split("(")[0].replace(" ", "")
    did = int(r["Dispatch_Id"])
    key = (k, r["Counter_Name"])
    if key not in last or did >= last[key][0]: last[key] = (did, float(r["Counter_Value"]))
agg = collections.defaultdict(dict)
for (k, c), (_, v) in last.items(): agg[k][c] = v
for k, c in sorted(agg.items()):
    if max(c.values()) > 1e6: print(k[:58].ljust(58), {n: f"{v:.3g}" for n, v in sorted(c.items())})
PY
