#!/bin/bash
# round-3 batch 12: stage thresholds on a 1/8 share (latency regime: the machine is not full, waiting for a stage's quorum costs chain time)
set -e
out=gpurun_out/r03_batch12
mkdir -p $out
for o in "" "shade_threshold=16,newpath_threshold=12" "shade_threshold=8,newpath_threshold=8" "shade_threshold=4,newpath_threshold=4,leaf_threshold=4" "shade_threshold=1,newpath_threshold=1,leaf_threshold=1,diel_threshold=1" \
   "shade_threshold=8,newpath_threshold=8,steps_per_trip=6" "shade_threshold=8,newpath_threshold=8,steps_per_trip=24" "shade_threshold=8,newpath_threshold=8,leaf_threshold=2" \
   "shade_threshold=8,newpath_threshold=8,wg_per_cu=2,threads=256" "shade_threshold=8,newpath_threshold=8,wg_per_cu=4,threads=256"; do
  echo "== RT_OPTS=$o" >> $out/partition8.log
  RT_OPTS=$o python tools/partition_time.py 8 2>&1 | grep "==" >> $out/partition8.log
done
cat $out/partition8.log
