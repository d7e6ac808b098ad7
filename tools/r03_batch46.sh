#!/bin/bash
# round-3 batch 46: with the tail hand-off on, which of the older scheduling tiers still pay (headline whole frame)
set -e
out=gpurun_out/r03_batch46
mkdir -p $out
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "" "semi_stride=0" "sparse_stride=0" "sparse_stride=0,semi_stride=0" "tier_kernel=0" "tier_kernel=0,sparse_stride=0,semi_stride=0" "prior=0" "sparse_wg_percent=20" "sparse_wg_percent=50" "heavy_factor_x10=15" "heavy_factor_x10=30" "newpath_threshold=16" "newpath_threshold=32" "shade_threshold=24" "steps_per_trip=12" "steps_per_trip=24" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
