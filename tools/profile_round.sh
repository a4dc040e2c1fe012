#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the exact default bench command
#   2. PMC passes (each in its own run, no trace domains) on tools/pmc_probe.py
# Outputs land in gpurun_out/prof_final/; tools/summarize_profiles.py turns them into profiles/rN/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --no-variants --no-workloads --no-host-path > "$OUT/bench_under_rocprof.log" 2>&1
pass() { name=$1; shift; echo "pass $name"; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/pmc_probe.py > "$OUT/$name.log" 2>&1; }
pass p_rd   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
pass p_wr   WRITE_SIZE TCC_REQ_sum
pass p_sec  TCC_READ_sum TCC_READ_SECTORS_sum TCP_TCC_READ_REQ_sum
pass p_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY
pass p_lds  SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
ls -R "$OUT" | head -40
