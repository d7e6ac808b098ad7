#!/bin/bash
# round-3 batch 2: first run of the tier kernel + cost prior: parity tests, then headline A/B and the N = 8 share
set -e
out=gpurun_out/r03_batch2
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; exit 1; }
tail -3 $out/gpu_tests.log
python tools/sweep.py --ns 500 --rounds 3 "" "prior=0" "tier_kernel=0" "tier1_depth=1" "tier1_depth=2" > $out/ab_headline.log 2>&1
cat $out/ab_headline.log
python tools/partition_time.py 1 8 > $out/partition_random.log 2>&1
grep "==" $out/partition_random.log
