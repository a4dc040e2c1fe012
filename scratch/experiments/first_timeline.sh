#!/bin/bash
# Kernel timeline (queue, start offset, duration) of the FIRST forward of a fresh engine on a workload:
#   scratch/experiments/first_timeline.sh WORKLOAD [key=value ...]   -> gpurun_out/first_timeline_WORKLOAD.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=$1
OUT=gpurun_out/ft_$W
rm -rf "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 scratch/experiments/first_once.py "$@" > gpurun_out/ft_$W.log 2>&1
echo "rc=$?"
python3 - "$W" <<'PY' > gpurun_out/first_timeline_$W.txt
import csv, glob, sys
w = sys.argv[1]
f = glob.glob(f"gpurun_out/ft_{w}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gnnvc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the forward = from the last F = 1 stage kernel's group on: the first of k_stage_f1 / k_long_f1 / k_giant_gather1 / k_lt_bytes_x
start = min(i for i, n in enumerate(names) if any(k in n for k in ("k_stage_f1<", "k_long_f1", "k_giant_gather1", "k_lt_bytes_x")))
t0 = int(rows[start]["Start_Timestamp"])
print(f"first forward of a fresh engine, {w}: kernel, queue, start offset us, duration us")
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
cat gpurun_out/first_timeline_$W.txt
