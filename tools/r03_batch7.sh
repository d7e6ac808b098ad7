#!/bin/bash
# round-3 batch 7: first run of the group kernel: parity, then whole-frame and 1/8-share sweeps
set -e
out=gpurun_out/r03_batch7
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -3 $out/gpu_tests.log
python tools/sweep.py --ns 500 --rounds 3 "" "group_kernel=0" "group_lanes=16" "group_depth=1" "group_depth=4" \
  "tier_auto=0,sparse_factor_x10=30,sparse_work_percent=10" "tier_auto=0,sparse_factor_x10=25,sparse_work_percent=15" "tier_auto=0,sparse_factor_x10=20,sparse_work_percent=25,group_depth=4" \
  "tier_auto=0,sparse_factor_x10=30,sparse_work_percent=10,tier1_factor_x10=60" "tier_auto=0,sparse_factor_x10=25,sparse_work_percent=15,tier1_factor_x10=80,tier1_depth=1" \
  > $out/sweep_whole.log 2>&1
cat $out/sweep_whole.log
for o in "" "group_kernel=0" "group_lanes=16" "group_depth=1" "group_depth=4" "tier_auto=0,tier1_pixels=2048,tier1_factor_x10=35,tier1_depth=2,heavy_factor_x10=14,sparse_factor_x10=14,sparse_work_percent=60,group_depth=4" "tier_auto=0,tier1_pixels=1024,tier1_factor_x10=45,tier1_depth=1,heavy_factor_x10=13,sparse_factor_x10=13,sparse_work_percent=70,group_depth=4"; do
  echo "== RT_OPTS=$o" >> $out/partition8.log
  RT_OPTS=$o python tools/partition_time.py 8 2>&1 | grep "==" >> $out/partition8.log
done
cat $out/partition8.log
