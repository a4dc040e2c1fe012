cd "$GRAFT_REPO_ROOT"
for t in 1024 2048 4096 8192 16384; do
  timeout -k 10 240 python bench.py --no-cpu-baseline --workload rmat22 --sorted-long-threshold $t 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rmat22 thr $t', round(d['ms_per_step'],3), [round(x,3) for x in d['stage_ms']], d['plan']['long_rows'])" || exit 1
done
for t in 512 2048 8192; do
  timeout -k 10 240 python bench.py --no-cpu-baseline --workload rmat22 --long-threshold $t 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rmat22 long-threshold $t', round(d['ms_per_step'],3), [round(x,3) for x in d['stage_ms']], d['plan']['long_rows'])" || exit 1
done
