#!/bin/bash
# Kernel timeline of ONE forward of a fresh engine on the metric graph (what a score-once caller runs), from a
# rocprofv3 kernel trace of tools/pmc_probe.py --forwards 1: every gnnvc kernel in start order with its queue, start
# offset and duration — the compact-table plan's builders run on the second queue under the stage-0 kernel.
# Run through gpurun from the repo root; writes gpurun_out/first_forward_timeline.txt.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ff_trace
rm -rf "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 tools/pmc_probe.py --forwards 1 > gpurun_out/ff_trace.log 2>&1
python3 - <<'PY' > gpurun_out/first_forward_timeline.txt
import csv, glob
f = glob.glob("gpurun_out/ff_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gnnvc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the forward = from its first kernel on: the pilot run of the F = 1 stage kernel (the hand-off's kernels — row classes, the two
# plans' builders — come before it and are listed above the line)
names = [r["Kernel_Name"] for r in rows]
start = next(i for i, n in enumerate(names) if "k_stage_f1<" in n)
print("hand-off (attach) of a fresh engine, metric graph: kernel, start offset us from the forward's first kernel, duration us")
t_f = int(rows[start]["Start_Timestamp"])
for r in rows[:start]:
    n = r["Kernel_Name"]
    print(f"  {n[n.find('k_'):].split('(')[0][:60]:62s} {(int(r['Start_Timestamp']) - t_f) / 1e3:10.1f}  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f}")
t0 = int(rows[start]["Start_Timestamp"])
print("one forward of a fresh engine, metric graph (ER 10 M / 100 M): kernel, queue, start offset us, duration us")
queues = {}
end = t0
for r in rows[start:]:
    n = r["Kernel_Name"]
    short = n[n.find("k_"):].split("(")[0][:60]
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, e)
    print(f"{short:62s} q{q}  {(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:9.1f}")
print(f"first kernel start to last kernel end: {(end - t0) / 1e6:.3f} ms")
PY
cat gpurun_out/first_forward_timeline.txt
