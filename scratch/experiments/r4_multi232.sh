#!/bin/bash
# fuzz_multi.py case 232 under variations
cd "$GRAFT_REPO_ROOT"
for v in "" "multi_pack=0" "multi_pieces=1" "multi_push=0" "parts=1" "parts=2" "compact_gather=0" "lds_table=0" "long_row_threshold=512" "wide_tiles=0" "multi_announce=1"; do
  python scratch/experiments/fuzz_multi.py 1 232 only $v 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
