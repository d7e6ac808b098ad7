#!/bin/bash
# round-3 batch 19: main_trim A/B in one process per share
set -e
out=gpurun_out/r03_batch19
mkdir -p $out
for st in 8 4 2; do STRIDE=$st ROUNDS=4 python tools/share_sweep.py "" "main_trim=1" "main_trim=1,tier1_depth=2,tier_auto=1" >> $out/shares.log 2>&1; done
NX=1920 NY=1080 STRIDE=8 ROUNDS=4 python tools/share_sweep.py "" "main_trim=1" >> $out/shares.log 2>&1
NX=1920 NY=1080 STRIDE=4 ROUNDS=4 python tools/share_sweep.py "" "main_trim=1" >> $out/shares.log 2>&1
grep -v "^/opt" $out/shares.log
