#!/bin/bash
# round-3 batch 10: defaults for general scenes (Book-2 final, Cornell): semi workgroups, tier kernel, tier sizes
set -e
out=gpurun_out/r03_batch10
mkdir -p $out
OLD="tier_auto=0,tier1_factor_x10=70,tier1_pixels=256,tier1_depth=1"
python tools/sweep.py --scene final --nx 800 --ny 800 --ns 200 --rounds 2 "" "semi_stride=0" "semi_stride=0,tier_kernel=0" "$OLD" "$OLD,semi_stride=0" "$OLD,semi_stride=0,tier_kernel=0" \
   "semi_priority=0" "tier_auto=0,tier1_factor_x10=45,tier1_pixels=4096,tier1_depth=4,sparse_work_percent=10" "tier_auto=0,tier1_factor_x10=45,tier1_pixels=8192,tier1_depth=4,sparse_work_percent=20,semi_stride=0" \
   "tier_auto=0,tier1_factor_x10=45,tier1_pixels=8192,tier1_depth=8,sparse_work_percent=20,sparse_stride=0" \
   > $out/sweep_final.log 2>&1
cat $out/sweep_final.log
python tools/sweep.py --scene cornell --nx 600 --ny 600 --ns 1000 --rounds 2 "" "semi_stride=0" "semi_stride=0,tier_kernel=0" "tier_kernel=0" "lpt=0" > $out/sweep_cornell.log 2>&1
cat $out/sweep_cornell.log
for o in "" "semi_stride=0" "tier_kernel=0,semi_stride=0" "lpt=0"; do
  echo "== cornell RT_OPTS=$o" >> $out/partition_cornell.log
  SCENE=cornell NX=600 NY=600 NS=1000 RT_OPTS=$o python tools/partition_time.py 8 2>&1 | grep "==" >> $out/partition_cornell.log
done
cat $out/partition_cornell.log
