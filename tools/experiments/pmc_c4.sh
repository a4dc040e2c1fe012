#!/bin/bash
# PMC passes over the two compact-table gather prototypes (GPU box, via gpurun from the repo root)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_c4
rm -rf "$OUT"; mkdir -p "$OUT"
export KSUB=3
pass() { name=$1; prog=$2; shift 2; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $prog > "$OUT/$name.log" 2>&1; }
pass w_rd  "tools/experiments/c4_wave.py 10000000 100000000 611 131072 256 1 1" TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass g_rd  "tools/experiments/c4_gather.py 10000000 100000000 6511 131072 256" TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass w_sq  "tools/experiments/c4_wave.py 10000000 100000000 611 131072 256 1 1" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
find "$OUT" -name "*counter_collection.csv" | head
