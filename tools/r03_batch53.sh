#!/bin/bash
# round-3 batch 53: the hand-off queue's overflow guard: GPU suite with the new test; headline check
set -e
out=gpurun_out/r03_batch53
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -1 $out/gpu_tests.log
timeout -k 10 300 python tools/sweep.py --ns 500 --rounds 3 "handoff=0" "" "handoff_pixels=16777216,handoff_poll_us=1" > $out/headline.log 2>&1; grep -v amdgpu $out/headline.log | cut -c1-230
