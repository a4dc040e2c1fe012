# one GPU-box job for the end of a round: per-rank compute of the multi-GPU path, a 2-rank gloo rehearsal, the profiles
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
for P in 2 4; do timeout -k 10 200 python tools/experiments/pieces_check.py er10m $P 4 prepare 2>&1 | tail -2 | tee gpurun_out/final/pieces_prepare_$P.log || exit 1; done
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | cut -c 1-900 | tee gpurun_out/final/gloo2.log || exit 1
bash tools/profile_round.sh > gpurun_out/final/profile_round.log 2>&1 || exit 1
tail -3 gpurun_out/final/profile_round.log
