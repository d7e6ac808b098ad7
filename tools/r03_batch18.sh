#!/bin/bash
# round-3 batch 18: surplus main workgroups leave (main_wgs): shares of the headline frame, then the tier grid's size
set -e
out=gpurun_out/r03_batch18
mkdir -p $out
for st in 8 4 2 1; do STRIDE=$st python tools/share_sweep.py "" "tier1_depth=2,tier_auto=1" >> $out/shares.log 2>&1; done
NX=1920 NY=1080 STRIDE=8 python tools/share_sweep.py "" >> $out/shares.log 2>&1
SCENE=final NX=800 NY=800 NS=200 STRIDE=8 python tools/share_sweep.py "" >> $out/shares.log 2>&1
grep -v "^/opt" $out/shares.log
python tools/partition_time.py 8 4 > $out/partition.log 2>&1; grep "==" $out/partition.log
