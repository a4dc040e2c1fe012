#!/bin/bash
# Kernel timeline of a fresh engine's ATTACH and first forward on a workload (everything after graph generation):
#   scratch/experiments/r4_timeline.sh WORKLOAD [key=value ...]   -> gpurun_out/r4_timeline_WORKLOAD.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=$1
TAG=${TAG:-$W}
OUT=gpurun_out/ft_$TAG
rm -rf "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 scratch/experiments/first_once.py "$@" > gpurun_out/ft_$TAG.log 2>&1
echo "rc=$?"
python3 - "$TAG" <<'PY' > gpurun_out/r4_timeline_$TAG.txt
import csv, glob, sys
w = sys.argv[1]
f = glob.glob(f"gpurun_out/ft_{w}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gnnvc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the LAST engine's attach: from the last k_validate_graph on
start = max(i for i, n in enumerate(names) if "k_validate_graph" in n)
t0 = int(rows[start]["Start_Timestamp"])
print(f"attach + first forward of a fresh engine, {w}: kernel, queue, start offset us, duration us")
queues = {}
end = t0
for r in rows[start:]:
    n = r["Kernel_Name"]
    short = n[n.find("k_"):].split("(")[0][:64]
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, e)
    print(f"{short:66s} q{q}  {(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:9.1f}")
print(f"first kernel start to last kernel end: {(end - t0) / 1e6:.3f} ms")
PY
cat gpurun_out/r4_timeline_$TAG.txt
