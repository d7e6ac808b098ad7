#!/bin/bash
# round-3 batch 4: who ends the last launch?  (diagnostic build: wave-end histogram + the pixels of the last waves)
set -e
out=gpurun_out/r03_batch4
mkdir -p $out
D=accelerated-ray-tracer_amd/lib/librt_mi355x_diag.so
RT_LIB_OVERRIDE=$D python tools/diag_wave_ends.py 500 > $out/wave_ends_whole.txt 2>&1
STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_wave_ends.py 500 > $out/wave_ends_eighth.txt 2>&1
RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py > $out/tier_pace.txt 2>&1
STRIDE=8 RT_LIB_OVERRIDE=$D python tools/diag_tier_pace.py >> $out/tier_pace.txt 2>&1
tail -22 $out/wave_ends_whole.txt; tail -22 $out/wave_ends_eighth.txt; cat $out/tier_pace.txt
