#!/bin/bash
# round-3 batch 14: second pass of the share sweeps around the best points of batch 13
set -e
out=gpurun_out/r03_batch14
mkdir -p $out
C="tier_auto=0,sparse_wg_percent=80"
P="$C,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60"
STRIDE=8 python tools/share_sweep.py "" "$P" "$P,tier_priority=1" "$P,tier_priority=2" "$P,tier1_depth=2" "$P,tier1_pixels=24576,tier1_factor_x10=14,heavy_factor_x10=14,sparse_factor_x10=14,sparse_work_percent=70" \
  "$P,tier1_pixels=12288,tier1_factor_x10=17" "$P,sparse_factor_x10=20" "$P,heavy_factor_x10=13" "$P,sparse_stride=0" "$P,sparse_stride=4" "$P,semi_priority=0" "$P,sparse_work_percent=80" \
  "$P,shade_threshold=16,newpath_threshold=12" "$P,shade_threshold=8,newpath_threshold=8" "$P,steps_per_trip=8" > $out/share8.log 2>&1
cat $out/share8.log
B="tier_auto=0,heavy_factor_x10=15,sparse_factor_x10=20,sparse_wg_percent=80,sparse_work_percent=20"
STRIDE=4 python tools/share_sweep.py "" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4" "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4" "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,sparse_work_percent=40" \
  "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,sparse_work_percent=40,tier_priority=1" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=0" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=15" "$B,tier1_pixels=8192,tier1_factor_x10=17,tier1_depth=4,sparse_factor_x10=17,sparse_work_percent=50" \
  "$B,tier1_pixels=4096,tier1_factor_x10=25,tier1_depth=2" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,heavy_factor_x10=13,sparse_factor_x10=17" > $out/share4.log 2>&1
cat $out/share4.log
A="tier_auto=0,heavy_factor_x10=20,sparse_factor_x10=30,sparse_wg_percent=80"
STRIDE=2 python tools/share_sweep.py "" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1" "$A,tier1_pixels=2048,tier1_factor_x10=35,tier1_depth=3,tier_priority=1" \
  "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1,semi_stride=0" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=2" \
  "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1,heavy_factor_x10=15,sparse_factor_x10=25" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1,sparse_priority=1" \
  "$A,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=4,tier_priority=1" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=0" > $out/share2.log 2>&1
cat $out/share2.log
STRIDE=1 python tools/share_sweep.py "" "tier_priority=1" "tier_priority=2" "tier_priority=1,tier1_pixels=2048,tier1_depth=4,tier_auto=0" "tier_priority=1,tier_auto=0,tier1_pixels=3072,tier1_factor_x10=35,tier1_depth=4" "tier_priority=1,tier_auto=0,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4" > $out/share1.log 2>&1
cat $out/share1.log
Q="tier_auto=0,heavy_factor_x10=20,sparse_factor_x10=30,sparse_wg_percent=80,sparse_work_percent=20,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,semi_stride=0"
NX=1920 NY=1080 STRIDE=8 python tools/share_sweep.py "" "$Q" "$Q,tier_priority=1" "$Q,tier1_pixels=16384,tier1_factor_x10=17" "$Q,heavy_factor_x10=15,sparse_factor_x10=20" "$Q,sparse_work_percent=40" "$Q,semi_stride=1" > $out/share8_1920.log 2>&1
cat $out/share8_1920.log
NX=1920 NY=1080 STRIDE=4 python tools/share_sweep.py "" "$Q" "$Q,tier_priority=1" "tier_auto=0,heavy_factor_x10=20,sparse_factor_x10=30,sparse_wg_percent=80,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1" > $out/share4_1920.log 2>&1
cat $out/share4_1920.log
