#!/bin/bash
# round-3 batch 15: the re-fitted table: partition tables (slowest rank, N = 1, 2, 4, 8) + a third pass on thresholds for the middle regimes
set -e
out=gpurun_out/r03_batch15
mkdir -p $out
python tools/partition_time.py > $out/partition_random_1200x800_500.log 2>&1; grep "==" $out/partition_random_1200x800_500.log
NX=1920 NY=1080 python tools/partition_time.py > $out/partition_random_1920x1080_500.log 2>&1; grep "==" $out/partition_random_1920x1080_500.log
STRIDE=4 python tools/share_sweep.py "" "shade_threshold=16,newpath_threshold=12" "shade_threshold=24,newpath_threshold=16" "steps_per_trip=8" "tier1_depth=2,tier_auto=1" "semi_stride=0" "semi_priority=0" > $out/share4.log 2>&1
cat $out/share4.log
STRIDE=2 python tools/share_sweep.py "" "shade_threshold=16,newpath_threshold=12" "shade_threshold=24,newpath_threshold=16" "semi_stride=0" "semi_priority=2" > $out/share2.log 2>&1
cat $out/share2.log
STRIDE=8 python tools/share_sweep.py "" "shade_threshold=8,newpath_threshold=8" "shade_threshold=16,newpath_threshold=12,leaf_threshold=4" "shade_threshold=16,newpath_threshold=12,diel_threshold=1" "shade_threshold=16,newpath_threshold=12,steps_per_trip=8" "shade_threshold=32,newpath_threshold=24" > $out/share8.log 2>&1
cat $out/share8.log
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 python tools/share_sweep.py "" "semi_stride=1" "semi_stride=1,semi_priority=1" "tier_priority=3" "semi_stride=1,tier_priority=3" > $out/share8_final.log 2>&1
cat $out/share8_final.log
SCENE=final NX=800 NY=800 NS=200 STRIDE=4 python tools/share_sweep.py "" "semi_stride=1" "tier_priority=3" > $out/share4_final.log 2>&1
cat $out/share4_final.log
