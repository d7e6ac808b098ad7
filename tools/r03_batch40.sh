#!/bin/bash
# round-3 batch 40: the tail launch right behind the main kernel (not behind the tier kernel's join): parity, Book-2 final, headline
set -e
out=gpurun_out/r03_batch40
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/gpu_parity.log 2>&1 || { tail -40 $out/gpu_parity.log; exit 1; }
tail -1 $out/gpu_parity.log
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "handoff=0" "" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
export SCENE=final NX=800 NY=800 NS=200
STRIDE=1 ROUNDS=3 python tools/share_sweep.py "handoff=0" "" "handoff_pixels=2048" "handoff_pixels=4096" "handoff_pixels=16384" "handoff_pixels=32768" > $out/final.log 2>&1; grep -v amdgpu $out/final.log
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "handoff=0" "" "handoff_pixels=8192" > $out/final_8.log 2>&1; grep -v amdgpu $out/final_8.log
STRIDE=2 ROUNDS=3 python tools/share_sweep.py "handoff=0" "" > $out/final_2.log 2>&1; grep -v amdgpu $out/final_2.log
