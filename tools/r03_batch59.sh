#!/bin/bash
# round-3 batch 59: raised priority for waves that are thin and no longer refilled (the decline phase of a launch)
set -e
out=gpurun_out/r03_batch59
mkdir -p $out
C=("" "thin_priority=1,thin_lanes=16" "thin_priority=1,thin_lanes=32" "thin_priority=1,thin_lanes=48" "thin_priority=2,thin_lanes=16" "thin_priority=2,thin_lanes=32" "thin_priority=3,thin_lanes=8" "thin_priority=1,thin_lanes=63")
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "${C[@]}" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
STRIDE=8 ROUNDS=4 python tools/share_sweep.py "${C[@]}" > $out/headline_8.log 2>&1; grep -v amdgpu $out/headline_8.log
STRIDE=2 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/headline_2.log 2>&1; grep -v amdgpu $out/headline_2.log
