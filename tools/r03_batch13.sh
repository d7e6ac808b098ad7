#!/bin/bash
# round-3 batch 13: tier sizes for the shares of a 2- / 4- / 8-GPU run with the tier kernel beside the main kernel
set -e
out=gpurun_out/r03_batch13
mkdir -p $out
A="tier_auto=0,heavy_factor_x10=20,sparse_factor_x10=30,sparse_wg_percent=80"
STRIDE=2 python tools/share_sweep.py "" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=0" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4" \
  "$A,tier1_pixels=2048,tier1_factor_x10=35,tier1_depth=3" "$A,tier1_pixels=2048,tier1_factor_x10=40,tier1_depth=2" "$A,tier1_pixels=1536,tier1_factor_x10=45,tier1_depth=3" \
  "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=8" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_work_percent=10" \
  "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_priority=2" "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1" \
  "$A,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=40" "$A,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=4" > $out/share2.log 2>&1
cat $out/share2.log
B="tier_auto=0,heavy_factor_x10=20,sparse_factor_x10=30,sparse_wg_percent=80,sparse_work_percent=20"
STRIDE=4 python tools/share_sweep.py "" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=0" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=2" "$B,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=4" "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4" \
  "$B,tier1_pixels=2048,tier1_factor_x10=40,tier1_depth=2" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=20" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=20,heavy_factor_x10=15" "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,tier_priority=1" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_priority=1" "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=8,sparse_work_percent=40" > $out/share4.log 2>&1
cat $out/share4.log
C="tier_auto=0,sparse_wg_percent=80"
STRIDE=8 python tools/share_sweep.py "" "$C,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=40,semi_stride=0" \
  "$C,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=8,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60" \
  "$C,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=8,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60,semi_stride=0" \
  "$C,tier1_pixels=32768,tier1_factor_x10=13,tier1_depth=16,heavy_factor_x10=13,sparse_factor_x10=13,sparse_work_percent=80" \
  "$C,tier1_pixels=32768,tier1_factor_x10=12,tier1_depth=16,heavy_factor_x10=12,sparse_factor_x10=12,sparse_work_percent=90,sparse_stride=0" \
  "$C,tier1_pixels=16384,tier1_factor_x10=17,tier1_depth=8,heavy_factor_x10=13,sparse_factor_x10=17,sparse_work_percent=60" \
  "$C,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60" > $out/share8.log 2>&1
cat $out/share8.log
NX=1920 NY=1080 STRIDE=8 python tools/share_sweep.py "" "$C,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=40,semi_stride=0" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=0" "$B,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=4" "$C,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=8,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60" \
  "$B,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,semi_stride=0" > $out/share8_1920.log 2>&1
cat $out/share8_1920.log
