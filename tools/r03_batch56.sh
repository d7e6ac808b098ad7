#!/bin/bash
# round-3 batch 56: the tail launch's shape on shares (where it takes 6 - 10 ms): LDS image without the scene, more workgroups per CU, thresholds
set -e
out=gpurun_out/r03_batch56
mkdir -p $out
W="tail_lds_scene=0,tail_wgs_per_cu=6"
C=("" "$W" "$W,handoff_pixels=12288" "$W,handoff_pixels=24576" "$W,handoff_pixels=36864" "handoff_pixels=12288" "handoff_pixels=24576" "tail_lds_scene=0,tail_wgs_per_cu=8")
for st in 2 4 8 1; do
  STRIDE=$st ROUNDS=3 python tools/share_sweep.py "${C[@]}" > $out/share_$st.log 2>&1; grep -v amdgpu $out/share_$st.log
done
