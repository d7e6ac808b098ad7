#!/bin/bash
# round-3 batch 33: tail hand-off threshold on shares of the headline frame (rank 0 of 2 / 4 / 8) and with smaller tiers
set -e
out=gpurun_out/r03_batch33
mkdir -p $out
for st in 2 4 8; do
  STRIDE=$st ROUNDS=4 python tools/share_sweep.py "handoff=0" "" "handoff_pixels=2048" "handoff_pixels=4096" "handoff_pixels=8192" "handoff_pixels=12288" "handoff_pixels=16384" "handoff_pixels=24576" "handoff_pixels=32768" > $out/share_$st.log 2>&1
  grep -v amdgpu $out/share_$st.log
done
