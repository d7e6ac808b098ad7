#!/bin/bash
# round-3 batch 11: the whole GPU test suite on the new defaults, then the partition tables of every BASELINE configuration
set -e
out=gpurun_out/r03_batch11
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -3 $out/gpu_tests.log
python tools/partition_time.py > $out/partition_random_1200x800_500.log 2>&1; grep "==" $out/partition_random_1200x800_500.log
NX=1920 NY=1080 python tools/partition_time.py > $out/partition_random_1920x1080_500.log 2>&1; grep "==" $out/partition_random_1920x1080_500.log
SCENE=cornell NX=600 NY=600 NS=1000 python tools/partition_time.py > $out/partition_cornell_600x600_1000.log 2>&1; grep "==" $out/partition_cornell_600x600_1000.log
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py > $out/partition_final_800x800_200.log 2>&1; grep "==" $out/partition_final_800x800_200.log
D=accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so
RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py > $out/tier_pace.txt 2>&1
STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py >> $out/tier_pace.txt 2>&1
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py >> $out/tier_pace.txt 2>&1
cat $out/tier_pace.txt
