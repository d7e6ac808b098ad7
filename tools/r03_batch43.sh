#!/bin/bash
# round-3 batch 43: with the tail hand-off on: how many parts the frame needs (headline whole, 1/8); stage quorums on the Cornell box's shares
set -e
out=gpurun_out/r03_batch43
mkdir -p $out
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "" "presplit_samples=0" "presplit_samples=0,split_samples=16" "presplit_samples=0,split_samples=8" "presplit_samples=4,split_samples=16" "presplit_samples=4" "presplit_samples=8,split_samples=24" "presplit_samples=0,split_samples=4" "presplit_samples=8,resplit_samples=96" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
STRIDE=8 ROUNDS=4 python tools/share_sweep.py "" "presplit_samples=0" "presplit_samples=0,split_samples=16" "presplit_samples=0,split_samples=8" "presplit_samples=4,split_samples=16" > $out/headline_8.log 2>&1; grep -v amdgpu $out/headline_8.log
export SCENE=cornell NX=600 NY=600 NS=1000
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "" "shade_threshold=16" "shade_threshold=8" "newpath_threshold=4" "shade_threshold=16,newpath_threshold=4" "leaf_threshold=16" "shade_threshold=8,newpath_threshold=2" "presplit_samples=0" > $out/cornell_8.log 2>&1; grep -v amdgpu $out/cornell_8.log
STRIDE=1 ROUNDS=3 python tools/share_sweep.py "" "shade_threshold=16" "newpath_threshold=4" "presplit_samples=0" > $out/cornell.log 2>&1; grep -v amdgpu $out/cornell.log
