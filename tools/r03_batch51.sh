#!/bin/bash
# round-3 batch 51: tier-1 waves linger and take handed-off pixels while the main kernel runs: parity, then whole frame and shares
set -e
out=gpurun_out/r03_batch51
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "handoff or split_frame" > $out/gpu_parity.log 2>&1 || { tail -40 $out/gpu_parity.log; exit 1; }
tail -1 $out/gpu_parity.log
timeout -k 10 300 python tools/sweep.py --ns 500 --rounds 3 "handoff_linger=0" "" "handoff_linger_wgs=128" "handoff_linger_wgs=512" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log | cut -c1-250
for st in 2 4 8; do
  STRIDE=$st ROUNDS=3 timeout -k 10 300 python tools/share_sweep.py "handoff_linger=0" "" > $out/share_$st.log 2>&1; grep -v amdgpu $out/share_$st.log
done
SCENE=final NX=800 NY=800 NS=200 STRIDE=1 ROUNDS=2 timeout -k 10 300 python tools/share_sweep.py "handoff_linger=0" "" "handoff_linger_wgs=128" "handoff_linger_wgs=512" > $out/final.log 2>&1; grep -v amdgpu $out/final.log
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 ROUNDS=2 timeout -k 10 300 python tools/share_sweep.py "handoff_linger=0" "" "handoff_linger_wgs=128" "handoff_linger_wgs=512" > $out/final_8.log 2>&1; grep -v amdgpu $out/final_8.log
