#!/bin/bash
# round-3 batch 21: the widened one-fma box test in the walk loop: parity suite, then A/B against the previous build (lib/old)
set -e
out=gpurun_out/r03_batch21
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -2 $out/gpu_tests.log
OLD=accelerated-ray-tracer_amd/lib/old/librt_mi355x.so
for round in 1 2; do
  echo "== new" >> $out/ab.log; python tools/sweep.py --ns 500 --rounds 3 "" >> $out/ab.log 2>&1
  echo "== old" >> $out/ab.log; RT_LIB_OVERRIDE=$OLD python tools/sweep.py --ns 500 --rounds 3 "" >> $out/ab.log 2>&1
done
for sc in "book1 1200 800 100" "cornell_smoke 600 600 1000" "final 800 800 200" "random_scene 1920 1080 500"; do
  set -- $sc
  echo "== new $1" >> $out/ab.log; python tools/sweep.py --scene $1 --nx $2 --ny $3 --ns $4 --rounds 2 "" >> $out/ab.log 2>&1
  echo "== old $1" >> $out/ab.log; RT_LIB_OVERRIDE=$OLD python tools/sweep.py --scene $1 --nx $2 --ny $3 --ns $4 --rounds 2 "" >> $out/ab.log 2>&1
done
for st in 8 2; do
  echo "== new share $st" >> $out/ab.log; STRIDE=$st python tools/share_sweep.py "" >> $out/ab.log 2>&1
  echo "== old share $st" >> $out/ab.log; RT_LIB_OVERRIDE=$OLD STRIDE=$st python tools/share_sweep.py "" >> $out/ab.log 2>&1
done
grep -E "^==|min|defaults" $out/ab.log
