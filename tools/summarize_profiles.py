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
(DST / "pmc_summary.json").write_text(json.dumps(summary, indent=1, sort_keys=True) + "\n")
print(json.dumps({k: {c: round(v, 3) for c, v in d.items() if c in ("traffic_bytes", "fabric_read_bytes", "hbm_write_bytes", "mfma_util")} for k, d in summary.items()}, indent=1))
