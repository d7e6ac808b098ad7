#!/bin/bash
# round-3 batch 5: tier sizing sweeps with the tier kernel (whole headline frame and its 1/8 share)
set -e
out=gpurun_out/r03_batch5
mkdir -p $out
B="tier_auto=0"
python tools/sweep.py --ns 500 --rounds 3 "" \
  "$B,tier1_pixels=4096,tier1_factor_x10=35,tier1_depth=4" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4" \
  "$B,tier1_pixels=8192,tier1_factor_x10=30,tier1_depth=8" \
  "$B,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=8" \
  "$B,tier1_pixels=16384,tier1_factor_x10=22,tier1_depth=16" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=25,sparse_work_percent=10" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,sparse_factor_x10=20,sparse_work_percent=15" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=2" \
  "$B,tier1_pixels=4096,tier1_factor_x10=30,tier1_depth=4,semi_stride=4" \
  "$B,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=8,semi_stride=2" \
  "$B,tier1_pixels=8192,tier1_factor_x10=25,tier1_depth=8,heavy_factor_x10=17" \
  "presplit_samples=4" "presplit_samples=4,split_samples=16" "presplit_samples=0,split_samples=16" "presplit_samples=0,split_samples=8" \
  > $out/sweep_whole.log 2>&1
cat $out/sweep_whole.log
