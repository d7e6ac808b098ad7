#!/bin/bash
# round-3 batch 58: a larger hand-off threshold for the first part only (its pixels are 16 samples long)
set -e
out=gpurun_out/r03_batch58
mkdir -p $out
C=("" "handoff_first_x=2" "handoff_first_x=4" "handoff_first_x=8" "handoff_first_x=16")
STRIDE=1 ROUNDS=5 python tools/share_sweep.py "${C[@]}" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
STRIDE=8 ROUNDS=4 python tools/share_sweep.py "${C[@]}" > $out/headline_8.log 2>&1; grep -v amdgpu $out/headline_8.log
SCENE=book1 NS=100 STRIDE=1 ROUNDS=5 python tools/share_sweep.py "${C[@]}" > $out/book1.log 2>&1; grep -v amdgpu $out/book1.log
SCENE=final NX=800 NY=800 NS=200 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/final.log 2>&1; grep -v amdgpu $out/final.log
