#!/bin/bash
# round-3 batch 45: the tail launch's shape: LDS image with / without the scene, workgroups per CU, threshold (headline whole)
set -e
out=gpurun_out/r03_batch45
mkdir -p $out
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "" "tail_lds_scene=0,tail_wgs_per_cu=4" "tail_lds_scene=0,tail_wgs_per_cu=6" "tail_lds_scene=0,tail_wgs_per_cu=6,handoff_pixels=32768" "tail_lds_scene=0,tail_wgs_per_cu=6,handoff_pixels=49152" "tail_wgs_per_cu=2" "tier_priority=0" "handoff_pixels=12288" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
