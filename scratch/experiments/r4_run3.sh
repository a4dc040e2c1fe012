#!/bin/bash
for o in 100 80 60 40 20; do python bench.py --no-cpu-baseline --no-workloads --no-variants --no-host-path --workload powerlaw1m --opt long_rows_main_percent=$o > gpurun_out/r4_b2.json 2>> gpurun_out/r4_b2.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r4_b2.json')); print(sys.argv[1], round(d['ms_per_step'],4), round(d['first_forward_ms'],4), [round(v,4) for v in d['stage_ms']])" $o; done
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "giant or long or hub" 2>&1 | tail -2
