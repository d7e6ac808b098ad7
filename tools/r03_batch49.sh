#!/bin/bash
# round-3 batch 49: GPU suite with the hand-off parity test; headline / shares check of the committed build
set -e
out=gpurun_out/r03_batch49
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
python tools/partition_time.py 1 2 4 8 > $out/partition.log 2>&1; grep "==" $out/partition.log
