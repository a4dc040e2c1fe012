# one GPU-box job for the end of a round: GPU tests, the profiles, the default bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1 || exit 1
timeout -k 10 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench.json || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['ms_per_step'], d['stage_ms'], d['roofline']['frac'], d['value'])"
