cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "long or hub or power" 2>&1 | tail -2 || exit 1
for w in powerlaw1m rmat22 rmat24; do
  timeout -k 10 240 python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), 'ms', round(d['value']/1e9,2), 'G edges/s', [round(x,3) for x in d['stage_ms']])" || exit 1
done
