#!/bin/bash
# round-3 batch 8: next-node prefetch A/B (whole frame, Book-1, 1/8 share); timelines with / without the cost prior; N = 8 sparse tuning;
# the general tier kernel on Cornell / Book-2 shares
set -e
out=gpurun_out/r03_batch9
mkdir -p $out
export TMPDIR=/tmp
PF=accelerated-ray-tracer_amd/lib/pf/librt_mi355x.so
for round in 1 2; do
  echo "== shipped" >> $out/ab_prefetch.log; python tools/sweep.py --ns 500 --rounds 3 "" >> $out/ab_prefetch.log 2>&1
  echo "== prefetch" >> $out/ab_prefetch.log; RT_LIB_OVERRIDE=$PF python tools/sweep.py --ns 500 --rounds 3 "" >> $out/ab_prefetch.log 2>&1
done
echo "== shipped book1" >> $out/ab_prefetch.log; python tools/sweep.py --scene book1 --ns 100 --rounds 3 "" >> $out/ab_prefetch.log 2>&1
echo "== prefetch book1" >> $out/ab_prefetch.log; RT_LIB_OVERRIDE=$PF python tools/sweep.py --scene book1 --ns 100 --rounds 3 "" >> $out/ab_prefetch.log 2>&1
echo "== shipped final" >> $out/ab_prefetch.log; python tools/sweep.py --scene final --nx 800 --ny 800 --ns 100 --rounds 2 "" >> $out/ab_prefetch.log 2>&1
echo "== prefetch final" >> $out/ab_prefetch.log; RT_LIB_OVERRIDE=$PF python tools/sweep.py --scene final --nx 800 --ny 800 --ns 100 --rounds 2 "" >> $out/ab_prefetch.log 2>&1
grep -E "^==|min" $out/ab_prefetch.log
T="tier_auto=0,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=15,sparse_work_percent=40,sparse_wg_percent=80"
for o in "" "$T,sparse_factor_x10=15" "$T,sparse_factor_x10=15,sparse_stride=4" "$T,sparse_factor_x10=15,sparse_stride=16" "$T,sparse_factor_x10=20" "$T,sparse_factor_x10=15,sparse_priority=0,tier_priority=1" \
   "tier_auto=0,tier1_pixels=4096,tier1_factor_x10=25,tier1_depth=2,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=40,sparse_wg_percent=80" \
   "tier_auto=0,tier1_pixels=16384,tier1_factor_x10=15,tier1_depth=8,heavy_factor_x10=15,sparse_factor_x10=15,sparse_work_percent=60,sparse_wg_percent=80" \
   "$T,sparse_factor_x10=15,semi_stride=0" \
   "tier_auto=0,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=12,sparse_factor_x10=12,sparse_stride=4,sparse_work_percent=60,sparse_wg_percent=80" \
   "tier_auto=0,tier1_pixels=8192,tier1_factor_x10=22,tier1_depth=4,heavy_factor_x10=12,sparse_factor_x10=12,sparse_stride=4,sparse_work_percent=60,sparse_wg_percent=80,semi_stride=0" \
   "tier_auto=0,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=13,sparse_factor_x10=13,sparse_stride=8,sparse_work_percent=60,sparse_wg_percent=90" \
   "tier_auto=0,tier1_pixels=8192,tier1_factor_x10=20,tier1_depth=4,heavy_factor_x10=11,sparse_factor_x10=11,sparse_stride=2,sparse_work_percent=70,sparse_wg_percent=80"; do
  echo "== RT_OPTS=$o" >> $out/partition8.log
  RT_OPTS=$o python tools/partition_time.py 8 2>&1 | grep "==" >> $out/partition8.log
done
cat $out/partition8.log
for sc in "cornell 600 600 1000" "final 800 800 200"; do
  set -- $sc
  for o in "" "tier_kernel=0"; do
    echo "== $1 RT_OPTS=$o" >> $out/partition_general.log
    SCENE=$1 NX=$2 NY=$3 NS=$4 RT_OPTS=$o python tools/partition_time.py 1 8 2>&1 | grep "==" >> $out/partition_general.log
  done
done
cat $out/partition_general.log
