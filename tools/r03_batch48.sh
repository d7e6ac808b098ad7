#!/bin/bash
# round-3 batch 48: box faces shared out over the lanes of a tier wave: parity; tier pace; Book-2 final and the Cornell box, whole and shares
set -e
out=gpurun_out/r03_batch48
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
RT_LIB_OVERRIDE=$PWD/accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so SCENE=final NX=800 NY=800 NS=200 python tools/diag_tier_pace.py > $out/pace_final.log 2>&1; grep -v amdgpu $out/pace_final.log
RT_LIB_OVERRIDE=$PWD/accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so SCENE=cornell NX=600 NY=600 NS=200 python tools/diag_tier_pace.py > $out/pace_cornell.log 2>&1; grep -v amdgpu $out/pace_cornell.log
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py 1 2 4 8 > $out/partition_final.log 2>&1; grep "==" $out/partition_final.log
SCENE=cornell NX=600 NY=600 NS=1000 python tools/partition_time.py 1 2 4 8 > $out/partition_cornell.log 2>&1; grep "==" $out/partition_cornell.log
SCENE=cornell_smoke NX=600 NY=600 NS=1000 python tools/partition_time.py 1 8 > $out/partition_smoke.log 2>&1; grep "==" $out/partition_smoke.log
