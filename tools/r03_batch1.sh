#!/bin/bash
# round-3 experiment batch 1 (one GPU box): stream concurrency micro-test; cost of the tier code's register budget in the bulk
set -e
out=gpurun_out/r03_batch1
mkdir -p $out
[ -s $out/concurrent_kernels.txt ] || ./tools/ubench/concurrent_kernels.bin > $out/concurrent_kernels.txt 2>&1
NT=accelerated-ray-tracer_amd/lib/nt/librt_mi355x.so
NT5=accelerated-ray-tracer_amd/lib/nt5/librt_mi355x.so
OFF="tier0_auto=0,tier0_pixels=0,tier1_pixels=0"
for round in 1 2; do
  echo "== shipped" >> $out/ab_notiers.log
  python tools/sweep.py --ns 500 --rounds 3 "" "$OFF" >> $out/ab_notiers.log 2>&1
  echo "== no-tier build (94 VGPR)" >> $out/ab_notiers.log
  RT_LIB_OVERRIDE=$NT python tools/sweep.py --ns 500 --rounds 3 "$OFF" >> $out/ab_notiers.log 2>&1
  echo "== no-tier build, 5 waves per SIMD (2 x 640)" >> $out/ab_notiers.log
  RT_LIB_OVERRIDE=$NT5 python tools/sweep.py --ns 500 --rounds 3 "$OFF" "$OFF,threads=640" >> $out/ab_notiers.log 2>&1
done
cat $out/concurrent_kernels.txt
grep -E "^==|min" $out/ab_notiers.log
