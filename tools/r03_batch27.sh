#!/bin/bash
# round-3 batch 27: tail hand-off, the count consumed one round later (no wait on the atomic): overhead (threshold 0) and thresholds
set -e
out=gpurun_out/r03_batch27
mkdir -p $out
timeout -k 10 400 python tools/sweep.py --ns 500 --rounds 4 "handoff=0" "handoff_pixels=0" "handoff_pixels=0,handoff_poll=16" "handoff_pixels=4096" "" "handoff_pixels=16384" "handoff_pixels=32768" "handoff_poll=16" > $out/headline.log 2>&1; cat $out/headline.log
