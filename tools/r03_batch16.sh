#!/bin/bash
# round-3 batch 16: Book-2 final shares: which change lost the 224 ms of batch 9 (tier variant, table row, priorities)?  Cornell shares: quorums.
set -e
out=gpurun_out/r03_batch16
mkdir -p $out
OLD="tier_auto=0,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=40,sparse_wg_percent=80"
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 python tools/share_sweep.py "" "tier_big=0" "$OLD" "$OLD,tier_big=0" "$OLD,tier_big=0,semi_stride=1,tier_priority=3" "$OLD,semi_stride=1,tier_priority=3" "tier_kernel=0" \
  "$OLD,tier_big=0,semi_stride=1,tier_priority=3,shade_threshold=32" > $out/share8_final.log 2>&1
cat $out/share8_final.log
SCENE=cornell NX=600 NY=600 NS=1000 STRIDE=8 python tools/share_sweep.py "" "shade_threshold=32" "shade_threshold=8,newpath_threshold=4" "shade_threshold=16,newpath_threshold=4,box_threshold=4" "box_threshold=1,medium_threshold=1" "steps_per_trip=4" > $out/share8_cornell.log 2>&1
cat $out/share8_cornell.log
