#!/bin/bash
# round-3 batch 38: tail hand-off, polls by time: period on the headline frame, on Book-2 final (whole and 1/8), on the headline's 1/8
set -e
out=gpurun_out/r03_batch38
mkdir -p $out
P="handoff_poll_us"
STRIDE=1 ROUNDS=4 python tools/share_sweep.py "handoff=0" "$P=100" "$P=250" "$P=500" "" "$P=2000" "$P=4000" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
STRIDE=8 ROUNDS=4 python tools/share_sweep.py "handoff=0" "$P=100" "$P=250" "$P=500" "" "$P=2000" > $out/headline_8.log 2>&1; grep -v amdgpu $out/headline_8.log
export SCENE=final NX=800 NY=800 NS=200
STRIDE=1 ROUNDS=3 python tools/share_sweep.py "handoff=0" "$P=100" "$P=250" "$P=500" "" "$P=2000" "$P=4000" "$P=500,handoff_pixels=4096" "$P=500,handoff_pixels=16384" > $out/final.log 2>&1; grep -v amdgpu $out/final.log
STRIDE=8 ROUNDS=3 python tools/share_sweep.py "handoff=0" "$P=100" "$P=250" "$P=500" "" "$P=2000" "$P=500,handoff_pixels=512" "$P=500,handoff_pixels=8192" > $out/final_8.log 2>&1; grep -v amdgpu $out/final_8.log
