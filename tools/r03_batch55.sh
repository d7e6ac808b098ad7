#!/bin/bash
# round-3 batch 55: two parts instead of three (the cost prior ranks the first part): where to split, every BASELINE frame
set -e
out=gpurun_out/r03_batch55
mkdir -p $out
C=("" "presplit_samples=0,split_samples=4" "presplit_samples=0,split_samples=8" "presplit_samples=0,split_samples=12" "presplit_samples=0,split_samples=16" "presplit_samples=0,split_samples=24" "presplit_samples=0,split_samples=32" "presplit_samples=4,split_samples=16")
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "${C[@]}" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
STRIDE=8 ROUNDS=4 python tools/share_sweep.py "${C[@]}" > $out/headline_8.log 2>&1; grep -v amdgpu $out/headline_8.log
STRIDE=2 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/headline_2.log 2>&1; grep -v amdgpu $out/headline_2.log
NX=1920 NY=1080 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/hd.log 2>&1; grep -v amdgpu $out/hd.log
SCENE=final NX=800 NY=800 NS=200 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/final.log 2>&1; grep -v amdgpu $out/final.log
SCENE=cornell NX=600 NY=600 NS=1000 STRIDE=1 ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/cornell.log 2>&1; grep -v amdgpu $out/cornell.log
