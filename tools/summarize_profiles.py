#!/usr/bin/env python3
"""Condense gpurun_out/prof_final (tools/profile_round.sh) into small committed files under
profiles/<round>/: kernel_stats.csv (our kernels only), pmc_summary.json (per-kernel counter
means + derived traffic), bench_under_rocprof.json."""
import collections
import csv
import glob
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof_final"
rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
DST = ROOT / "profiles" / rnd
DST.mkdir(parents=True, exist_ok=True)


def short(name: str) -> str:
    if "gnnvc" not in name:
        return "calibration_copy_1GiB" if "CUDAFunctorOnSelf_add<float>" in name else ""
    n = name[name.index("k_"):]
    n = n.split("(")[0]
    return n


import os
# (gpurun merges a run's files into gpurun_out/ without removing an earlier run's: take the newest of each kind)
stats = sorted(glob.glob(str(SRC / "trace" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [[short(r[0])] + r[1:] for r in rows[1:] if "gnnvc" in r[0]]
    with open(DST / "kernel_stats.csv", "w", newline="") as f:
        csv.writer(f).writerows(keep)
log = SRC / "bench_under_rocprof.log"
if log.exists():
    lines = [l for l in log.read_text().splitlines() if l.startswith("{")]
    if lines:
        (DST / "bench_under_rocprof.json").write_text(lines[-1] + "\n")

# per kernel and counter: the value of the LAST launch of the run (steady state: the first forward runs
# without the per-graph plans, the second builds them)
agg = collections.defaultdict(dict)
newest = {}
for p in glob.glob(str(SRC / "p_*" / "*" / "*_counter_collection.csv")):
    grp = pathlib.Path(p).parts[-3]
    if grp not in newest or os.path.getmtime(p) > os.path.getmtime(newest[grp]):
        newest[grp] = p
for p in newest.values():
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        if k:
            did = int(r.get("Dispatch_Id", 0) or 0)
            cur = agg[k].get(r["Counter_Name"])
            if cur is None or did >= cur[0]:
                agg[k][r["Counter_Name"]] = (did, float(r["Counter_Value"]))
# One steady-state FORWARD: a forward of the metric graph starts with k_lt_bytes_x (the F = 1 stage's byte table, made from that
# forward's x; graphs without that plan: with the stage-0 tile kernel k_stage_f1), so the dispatches from the last such launch to
# the end of the run are exactly the last forward of the probe (kernels run one after the other under --pmc, side-stream
# kernels included).
forward = collections.defaultdict(float)
for p in newest.values():
    rows = [r for r in csv.DictReader(open(p)) if "gnnvc" in r["Kernel_Name"]]
    first_ids = sorted({int(r["Dispatch_Id"]) for r in rows if "k_lt_bytes_x" in r["Kernel_Name"]}) or \
        sorted({int(r["Dispatch_Id"]) for r in rows if "k_stage_f1<" in r["Kernel_Name"]})
    if not first_ids:
        continue
    start = first_ids[-1]
    for r in rows:
        if int(r["Dispatch_Id"]) >= start:
            forward[r["Counter_Name"]] += float(r["Counter_Value"])
summary = {}
for k, ctrs in sorted(agg.items()):
    m = {c: v for c, (_, v) in ctrs.items()}
    d = dict(m)
    if "TCC_EA0_RDREQ_sum" in m:
        # calibrated on the 1 GiB copy of the same run: every L2->fabric read request moves 128 B
        d["fabric_read_bytes"] = m["TCC_EA0_RDREQ_sum"] * 128
    if "WRITE_SIZE" in m:
        d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "fabric_read_bytes" in d and "hbm_write_bytes" in d:
        d["traffic_bytes"] = d["fabric_read_bytes"] + d["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; MFMA busy cycles sum over the 1024 SIMDs
        kernel_cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        d["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * kernel_cycles)
        d["mfma_instructions"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0   # 64 cycles per v_mfma_f32_32x32x2_f32
    summary[k] = d
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (source_hash: the identity of the kernels these counters were taken from)
meta = {"source_hash": bench.source_hash(), "probe": "tools/pmc_probe.py (metric graph), last of its forwards",
        "forward_counters": dict(forward)}
if "TCC_EA0_RDREQ_sum" in forward and "WRITE_SIZE" in forward:
    meta["forward_fabric_read_bytes"] = forward["TCC_EA0_RDREQ_sum"] * 128
    meta["forward_hbm_write_bytes"] = forward["WRITE_SIZE"] * 1024
    meta["forward_traffic_bytes"] = meta["forward_fabric_read_bytes"] + meta["forward_hbm_write_bytes"]
summary["_meta"] = meta
(DST / "pmc_summary.json").write_text(json.dumps(summary, indent=1, sort_keys=True) + "\n")
print(json.dumps({k: {c: round(v, 3) for c, v in d.items() if c in ("traffic_bytes", "fabric_read_bytes", "hbm_write_bytes", "mfma_util")} for k, d in summary.items() if k != "_meta"}, indent=1))
print(json.dumps({k: v for k, v in meta.items() if k != "forward_counters"}, indent=1))
