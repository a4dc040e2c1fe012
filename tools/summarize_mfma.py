#!/usr/bin/env python3
"""gpurun_out/prof_mfma (tools/profile_mfma.sh) -> profiles/<round>/mfma_ab.json: per workload and mfma_dense setting the
forward time, and per stage kernel its average duration (rocprofv3 --kernel-trace --stats) and, from the PMC run,
matrix-core busy share, MFMA / VALU instruction counts and VALU issue share.
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
                sums the 8 XCDs)
  valu_issue  = SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES  (both in quad-cycles per the guide) — share of wave time spent
                issuing vector instructions
"""
import collections
import csv
import glob
import json
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof_mfma"
rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
DST = ROOT / "profiles" / rnd
DST.mkdir(parents=True, exist_ok=True)


def short(name):
    if "gnnvc" not in name:
        return ""
    return name[name.index("k_"):].split("(")[0].replace(" ", "")


def newest(pattern):
    hits = sorted(glob.glob(pattern), key=os.path.getmtime)
    return hits[-1] if hits else None


out = {}
for w in ("er10m", "rmat22"):
    for m in (0, 1, 2):
        tag = f"{w}_m{m}"
        entry = {"workload": w, "mfma_dense": m, "kernels": {}}
        try:
            line = [l for l in (SRC / f"bench_{tag}.json").read_text().splitlines() if l.startswith("{")][-1]
            b = json.loads(line)
            entry["ms_per_step_under_rocprof"] = b["ms_per_step"]
            entry["stage_ms"] = b["stage_ms"]
        except Exception as ex:   # noqa: BLE001
            entry["bench_error"] = str(ex)
        st = newest(str(SRC / f"trace_{tag}" / "*" / "*_kernel_stats.csv"))
        if st:
            for r in csv.DictReader(open(st)):
                k = short(r["Name"])
                if k.startswith(("k_stage_f1<", "k_stage_f16<", "k_long_", "k_giant_")):
                    # steady state: the minimum over launches is the early-exit / warm value; report average of all
                    entry["kernels"].setdefault(k, {}).update({"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                               "min_ms": float(r["MinNs"]) / 1e6})
        pc = newest(str(SRC / f"pmc_{tag}" / "*" / "*_counter_collection.csv"))
        if pc:
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            cnt = collections.defaultdict(int)
            last = {}
            for r in csv.DictReader(open(pc)):
                k = short(r["Kernel_Name"])
                if not k.startswith(("k_stage_f1<", "k_stage_f16<")):
                    continue
                did = int(r.get("Dispatch_Id", 0) or 0)
                key = (k, r["Counter_Name"])
                if key not in last or did >= last[key][0]:      # the LAST launch of the run: steady state
                    last[key] = (did, float(r["Counter_Value"]))
            for (k, c), (_, v) in last.items():
                agg[k][c] = v
            for k, c in agg.items():
                d = entry["kernels"].setdefault(k, {})
                cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
                d["pmc"] = {n: c[n] for n in sorted(c)}
                if cyc:
                    d["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
                if c.get("SQ_WAVE_CYCLES"):
                    d["valu_issue"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
                d["mfma_instructions"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 64.0   # 64 cycles per v_mfma_f32_32x32x2_f32
        out[tag] = entry
(DST / "mfma_ab.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
for tag, e in out.items():
    print(tag, "ms/step", round(e.get("ms_per_step_under_rocprof", float("nan")), 3))
    for k, d in sorted(e["kernels"].items()):
        if "mfma_busy" in d or d.get("avg_ms", 0) > 0.2:
            print("   ", k[:60], {x: (round(v, 4) if isinstance(v, float) else v) for x, v in d.items() if x != "pmc"})
