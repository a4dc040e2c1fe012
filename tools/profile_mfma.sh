#!/bin/bash
# MFMA evidence: the PLAIN path (no per-graph plans: the kernels that both gather and run the dense
# layers) with the dense layers on the VALU (mfma_dense 0), on the fp32 matrix cores everywhere (1) and in the 16-wide
# stages only (2, the default), on the metric graph and on R-MAT-22.  Per configuration: one kernel-trace run (times)
# and one PMC run (matrix-core and vector counters; counters in their own run, no trace domains).
# Run through gpurun from the repo root; tools/summarize_mfma.py condenses the output into profiles/<round>/mfma_ab.json.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_mfma
rm -rf "$OUT"; mkdir -p "$OUT"
for w in er10m rmat22; do
  for m in 0 1 2; do
    tag=${w}_m${m}
    args="--workload $w --mfma $m --no-lds-table --no-compact --no-cpu-baseline --no-variants --kernel-trace 0"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -- python3 bench.py $args --steps 10 > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
        --output-format csv -d "$OUT/pmc_$tag" -- python3 bench.py $args --steps 2 --warmup 1 > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err"
    echo "done $tag: $(head -c 160 $OUT/bench_$tag.json)"
  done
done
