#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the exact default bench command
#   2. PMC passes (each in its own run, no trace domains) on tools/pmc_probe.py: the metric graph with every counter group,
#      every other workload of bench.py's `workloads` block with the two traffic groups, the metric graph with the plans off,
#      and the FIRST forward of the skewed graphs (what a score-once caller moves)
# Outputs land in gpurun_out/prof_final/; tools/summarize_profiles.py turns them into profiles/rN/.
#   usage: tools/profile_round.sh [all|trace|er10m|<workload> ...]     (default: all)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
mkdir -p "$OUT"
WHAT=${*:-all}
has() { [[ " $WHAT " == *" all "* || " $WHAT " == *" $1 "* ]]; }
pass() { dir=$1; name=$2; args=$3; shift 3; echo "pass $dir/$name"; rm -rf "$OUT/$dir/$name"; mkdir -p "$OUT/$dir"
         timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$dir/$name" -- python3 tools/pmc_probe.py $args > "$OUT/$dir/$name.log" 2>&1; }
if has trace; then
  rm -rf "$OUT/trace"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --no-variants --no-workloads --no-host-path > "$OUT/bench_under_rocprof.log" 2>&1
fi
for w in er20k er100k rmat22 powerlaw1m; do   # kernel stats of the other workloads' steady state (the same bench command, one workload)
  if has trace_$w || has trace; then
    rm -rf "$OUT/trace_$w"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 bench.py --workload $w --steps 50 --no-cpu-baseline --no-variants --no-workloads --no-host-path > "$OUT/bench_under_rocprof_$w.log" 2>&1
  fi
done
if has er10m; then
  pass er10m p_rd   "" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
  pass er10m p_wr   "" WRITE_SIZE TCC_REQ_sum
  pass er10m p_sec  "" TCC_READ_sum TCC_READ_SECTORS_sum TCP_TCC_READ_REQ_sum
  pass er10m p_mfma "" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
  pass er10m p_lds  "" SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  pass er10m_plain p_rd "--plain 1" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
  pass er10m_plain p_wr "--plain 1" WRITE_SIZE TCC_REQ_sum
fi
for w in er20k er100k er3m rmat22 rmat24 powerlaw1m; do
  if has $w; then
    pass $w p_rd "--workload $w" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
    pass $w p_wr "--workload $w" WRITE_SIZE TCC_REQ_sum
  fi
done
for w in rmat22 rmat24; do
  if has $w; then
    pass ${w}_first p_rd "--workload $w --first 1" TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
    pass ${w}_first p_wr "--workload $w --first 1" WRITE_SIZE TCC_REQ_sum
  fi
done
ls "$OUT"
