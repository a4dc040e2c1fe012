# rocprofv3 kernel stats of bench.py on another workload (GPU box): bash scratch/experiments/profile_workload.sh rmat22 [extra bench args]
set -u
w=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${TAG:-$w}
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --workload $w "$@" > gpurun_out/prof_$tag.log 2>&1
tail -1 gpurun_out/prof_$tag.log | cut -c 1-200
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/ks_$tag.csv
cp $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) gpurun_out/kt_$tag.csv
