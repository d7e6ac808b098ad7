#!/bin/bash
# round-3 batch 26: tail hand-off keyed on the count of lanes out of work: parity, threshold sweep on the headline
set -e
out=gpurun_out/r03_batch26
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/gpu_parity.log 2>&1 || { tail -40 $out/gpu_parity.log; exit 1; }
tail -2 $out/gpu_parity.log
timeout -k 10 400 python tools/sweep.py --ns 500 --rounds 3 "handoff=0" "handoff_pixels=0" "handoff_pixels=2048" "handoff_pixels=4096" "" "handoff_pixels=16384" "handoff_pixels=32768" "handoff_pixels=65536" "handoff_poll=16" > $out/headline.log 2>&1; cat $out/headline.log
