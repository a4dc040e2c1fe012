# rocprofv3 kernel stats of bench.py on another workload (GPU box): bash tools/experiments/profile_workload.sh rmat22
set -u
w=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python3 bench.py --no-cpu-baseline --workload $w > gpurun_out/prof_$w.log 2>&1
tail -1 gpurun_out/prof_$w.log | cut -c 1-200
