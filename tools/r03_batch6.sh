#!/bin/bash
# round-3 batch 6: priorities.  Whole frame: tier 3 on workgroups of its own at raised priority.  1/8 share: sparse / tier priority.
set -e
out=gpurun_out/r03_batch6
mkdir -p $out
python tools/sweep.py --ns 500 --rounds 3 "" "prior=0" \
  "semi_stride=1" "semi_stride=1,semi_priority=1" "semi_stride=1,semi_priority=2" "semi_stride=1,semi_priority=3" \
  "semi_stride=2,semi_priority=2" "semi_stride=1,semi_priority=2,heavy_factor_x10=17" "semi_stride=1,semi_priority=2,heavy_factor_x10=25" \
  "tier_priority=2" "tier_priority=1" "sparse_priority=2" "sparse_priority=1" \
  > $out/sweep_whole.log 2>&1
cat $out/sweep_whole.log
for o in "" "sparse_priority=0" "sparse_priority=1" "sparse_priority=1,tier_priority=2" "sparse_priority=0,tier_priority=1" "semi_stride=1,semi_priority=2" "sparse_priority=1,semi_stride=1,semi_priority=1"; do
  echo "== RT_OPTS=$o" >> $out/partition8.log
  RT_OPTS=$o python tools/partition_time.py 8 2>&1 | grep "==" >> $out/partition8.log
done
cat $out/partition8.log
