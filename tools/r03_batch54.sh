#!/bin/bash
# round-3 batch 54: short frames (100 spp): how many parts, where to split, hand-off threshold (book1 and the random scene at 1200x800 @ 100)
set -e
out=gpurun_out/r03_batch54
mkdir -p $out
C=("" "handoff=0" "presplit_samples=0" "presplit_samples=0,split_samples=16" "presplit_samples=0,split_samples=8" "presplit_samples=4,split_samples=16" "presplit_samples=4,split_samples=12" "split_samples=24" "handoff_pixels=8192" "handoff_pixels=32768" "presplit_samples=0,split_samples=16,handoff_pixels=32768")
SCENE=book1 NS=100 STRIDE=1 ROUNDS=5 python tools/share_sweep.py "${C[@]}" > $out/book1.log 2>&1; grep -v amdgpu $out/book1.log
NS=100 STRIDE=1 ROUNDS=5 python tools/share_sweep.py "${C[@]}" > $out/random.log 2>&1; grep -v amdgpu $out/random.log
