#!/bin/bash
# suite + the metric graph with / without the dense kernels' zero-term skip + the power-law graph (giant walk in one-window segments)
python -m pytest tests -q -m gpu -x > gpurun_out/r4_suite2.log 2>&1; tail -6 gpurun_out/r4_suite2.log
python bench.py --no-cpu-baseline --no-workloads > gpurun_out/r4_b1.json 2> gpurun_out/r4_b1.err
python bench.py --no-cpu-baseline --no-workloads --no-variants --no-host-path --opt dense_skip_zeros=0 > gpurun_out/r4_b1_off.json 2>> gpurun_out/r4_b1.err
python bench.py --no-cpu-baseline --no-workloads --no-variants --no-host-path --workload powerlaw1m > gpurun_out/r4_b1_pl.json 2>> gpurun_out/r4_b1.err
python bench.py --no-cpu-baseline --no-workloads --no-variants --no-host-path --workload rmat22 > gpurun_out/r4_b1_rmat22.json 2>> gpurun_out/r4_b1.err
python - <<PY
import json
for f in ("gpurun_out/r4_b1.json","gpurun_out/r4_b1_off.json","gpurun_out/r4_b1_pl.json","gpurun_out/r4_b1_rmat22.json"):
    try:
        d=json.load(open(f)); print(f, round(d["ms_per_step"],4), round(d["first_forward_ms"],4), [round(v,4) for v in d["stage_ms"]]); print({k:round(v["ms_per_forward"],4) for k,v in d["roofline"]["kernels"].items()})
    except Exception as ex:
        print(f, "failed", ex)
PY
