#!/bin/bash
# power-law: where the long rows go now that a giant stream's walk is one-window segments
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stream_sum or giant or tolerance_mode or pruned_adjacency" > gpurun_out/r4_t3.log 2>&1; tail -5 gpurun_out/r4_t3.log
for o in "" "--opt long_rows_on_main=0" "--opt long_rows_on_main=0 --opt giant_gather_first=1"; do
  python bench.py --no-cpu-baseline --no-workloads --no-variants --no-host-path --workload powerlaw1m $o > gpurun_out/r4_b2.json 2>> gpurun_out/r4_b2.err
  python - "$o" <<PY
import json,sys
d=json.load(open("gpurun_out/r4_b2.json")); print(sys.argv[1] or "default", round(d["ms_per_step"],4), round(d["first_forward_ms"],4), [round(v,4) for v in d["stage_ms"]], d["plan"]["giant_stream_segments"])
PY
done
bash scratch/experiments/r4_timeline.sh powerlaw1m > /dev/null 2>&1; tail -24 gpurun_out/r4_timeline_powerlaw1m.txt
