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
for w in ("er20k", "er100k", "rmat22", "powerlaw1m"):
    st = sorted(glob.glob(str(SRC / f"trace_{w}" / "*" / "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if st:
        rows = list(csv.reader(open(st[0])))
        keep = [rows[0]] + [[short(r[0])] + r[1:] for r in rows[1:] if "gnnvc" in r[0]]
        with open(DST / f"kernel_stats_{w}.csv", "w", newline="") as f:
            csv.writer(f).writerows(keep)
log = SRC / "bench_under_rocprof.log"
if log.exists():
    lines = [l for l in log.read_text().splitlines() if l.startswith("{")]
    if lines:
        (DST / "bench_under_rocprof.json").write_text(lines[-1] + "\n")



def summarise(groups: dict, what: str):
    """groups: {pass name: counter_collection.csv}.  Per kernel and counter the value of its LAST launch of the run, and the
    sums over ONE forward: the dispatches behind the run's marker kernel (tools/pmc_probe.py launches the only add<double>
    kernel of the run right in front of the forward to take: the last one = the steady state, or a fresh engine's first)."""
    agg = collections.defaultdict(dict)
    forward = collections.defaultdict(float)
    kernels_in_forward = collections.Counter()
    for p in groups.values():
        rows = list(csv.DictReader(open(p)))
        marks = sorted({int(r["Dispatch_Id"]) for r in rows if "add<double>" in r["Kernel_Name"] or "AddFunctor" in r["Kernel_Name"] and "double" in r["Kernel_Name"]})
        ours = [r for r in rows if "gnnvc" in r["Kernel_Name"]]
        if marks:
            start = marks[-1]
        else:   # (older probes: a forward of the metric graph starts with k_lt_bytes_x, graphs without that plan with k_stage_f1)
            first_ids = sorted({int(r["Dispatch_Id"]) for r in ours if "k_lt_bytes_x" in r["Kernel_Name"]}) or \
                sorted({int(r["Dispatch_Id"]) for r in ours if "k_stage_f1<" in r["Kernel_Name"]})
            if not first_ids:
                continue
            start = first_ids[-1]
        seen = set()
        for r in rows:
            k = short(r["Kernel_Name"])
            if not k:
                continue
            did = int(r.get("Dispatch_Id", 0) or 0)
            cur = agg[k].get(r["Counter_Name"])
            if cur is None or did >= cur[0]:
                agg[k][r["Counter_Name"]] = (did, float(r["Counter_Value"]))
            if did >= start and "gnnvc" in r["Kernel_Name"]:
                forward[r["Counter_Name"]] += float(r["Counter_Value"])
                if (did, k) not in seen:
                    seen.add((did, k))
        per = collections.Counter(k for _, k in seen)
        for k, c in per.items():
            kernels_in_forward[k] = max(kernels_in_forward[k], c)
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
    meta = {"source_hash": bench.source_hash(), "probe": f"tools/pmc_probe.py ({what})", "forward_counters": dict(forward),
            "kernels_in_forward": dict(kernels_in_forward)}
    if "TCC_EA0_RDREQ_sum" in forward and "WRITE_SIZE" in forward:
        meta["forward_fabric_read_bytes"] = forward["TCC_EA0_RDREQ_sum"] * 128
        meta["forward_hbm_write_bytes"] = forward["WRITE_SIZE"] * 1024
        meta["forward_traffic_bytes"] = meta["forward_fabric_read_bytes"] + meta["forward_hbm_write_bytes"]
    cal = summary.get("calibration_copy_1GiB", {})
    if "fabric_read_bytes" in cal:
        meta["calibration_copy_1GiB_read_bytes_measured"] = cal["fabric_read_bytes"]
    summary["_meta"] = meta
    return summary


sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (source_hash: the identity of the kernels these counters were taken from)

# gpurun_out/prof_final/<workload>/<pass>/...counter_collection.csv  (<workload> = er10m, er10m_plain, rmat22, rmat22_first, ...)
for wdir in sorted(p for p in SRC.iterdir() if p.is_dir() and p.name != "trace"):
    groups = {}
    for p in glob.glob(str(wdir / "p_*" / "**" / "*_counter_collection.csv"), recursive=True):
        grp = pathlib.Path(p).relative_to(wdir).parts[0]
        if grp not in groups or os.path.getmtime(p) > os.path.getmtime(groups[grp]):
            groups[grp] = p
    if not groups:
        continue
    name = wdir.name
    what = {"er10m": "metric graph, last of its forwards"}.get(name, name.replace("_plain", ", every plan off, last forward").replace("_first", ", a fresh engine's FIRST forward") if ("_plain" in name or "_first" in name) else f"{name}, last of its forwards")
    summary = summarise(groups, what)
    out = DST / ("pmc_summary.json" if name == "er10m" else f"pmc_summary_{name}.json")
    out.write_text(json.dumps(summary, indent=1, sort_keys=True) + "\n")
    meta = summary["_meta"]
    print(name, {k: v for k, v in meta.items() if k not in ("forward_counters", "kernels_in_forward")})
    print("   ", {k: {c: round(v / 1e6, 2) for c, v in d.items() if c in ("traffic_bytes", "fabric_read_bytes", "hbm_write_bytes")}
                  for k, d in summary.items() if k != "_meta" and d.get("traffic_bytes", 0) > 5e7})
