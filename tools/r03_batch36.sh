#!/bin/bash
# round-3 batch 36: the re-fitted share rows on every rank (headline, 1920x1080); Book-2 final's shares with the hand-off
set -e
out=gpurun_out/r03_batch36
mkdir -p $out
python tools/partition_time.py 1 2 4 8 > $out/partition.log 2>&1; grep "==" $out/partition.log
NX=1920 NY=1080 python tools/partition_time.py 1 2 4 8 > $out/partition_hd.log 2>&1; grep "==" $out/partition_hd.log
SCENE=final NX=800 NY=800 NS=200 python tools/partition_time.py 1 2 4 8 > $out/partition_final.log 2>&1; grep "==" $out/partition_final.log
SCENE=final NX=800 NY=800 NS=200 RT_OPTS=handoff=0 python tools/partition_time.py 1 2 4 8 > $out/partition_final_off.log 2>&1; grep "==" $out/partition_final_off.log
SCENE=bouncing NS=500 python tools/partition_time.py 1 8 > $out/partition_bouncing.log 2>&1; grep "==" $out/partition_bouncing.log
