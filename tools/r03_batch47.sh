#!/bin/bash
# round-3 batch 47: cycles per ray in the tier loops (diagnostic build): Book-2 final, the Cornell box, the headline frame
set -e
out=gpurun_out/r03_batch47
mkdir -p $out
export RT_LIB_OVERRIDE=$PWD/accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so
SCENE=final NX=800 NY=800 NS=200 python tools/diag_tier_pace.py > $out/final.log 2>&1; grep -v amdgpu $out/final.log
SCENE=cornell NX=600 NY=600 NS=200 python tools/diag_tier_pace.py > $out/cornell.log 2>&1; grep -v amdgpu $out/cornell.log
python tools/diag_tier_pace.py > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log
