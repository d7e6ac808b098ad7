#!/bin/bash
# round-3 batch 17: last whole-frame knobs (headline): sparse tier with semi workgroups on, split points, list threshold; Book-1 and 1920x1080 whole
set -e
out=gpurun_out/r03_batch17
mkdir -p $out
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "" "tier_auto=0,sparse_factor_x10=35,sparse_work_percent=8" "tier_auto=0,sparse_factor_x10=30,sparse_work_percent=10" "tier_auto=0,sparse_factor_x10=50" "sparse_stride=0" \
  "heavy_factor_x10=17" "heavy_factor_x10=25" "presplit_samples=4" "presplit_samples=6,split_samples=24" "split_samples=48,presplit_samples=12" "presplit_samples=4,split_samples=16" \
  "tier_auto=0,tier1_pixels=1024,tier1_factor_x10=55,tier1_depth=2" "tier_auto=0,tier1_pixels=2048,tier1_factor_x10=40,tier1_depth=3" "tier_auto=0,tier1_depth=2" "tier_auto=0,tier1_depth=4" \
  "shade_threshold=24,newpath_threshold=20" "shade_threshold=40,newpath_threshold=28" "semi_priority=1,sparse_priority=2" > $out/share1.log 2>&1
cat $out/share1.log
SCENE=book1 NS=100 STRIDE=1 ROUNDS=4 python tools/share_sweep.py "" "semi_stride=0" "prior=0" "tier_priority=3" > $out/book1.log 2>&1
cat $out/book1.log
NX=1920 NY=1080 STRIDE=1 python tools/share_sweep.py "" "semi_stride=0" "prior=0" "tier_priority=3" > $out/share1_1920.log 2>&1
cat $out/share1_1920.log
