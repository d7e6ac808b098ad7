#!/bin/bash
# round-3 batch 29: tail hand-off with its counters in their own cache line: overhead of each way of learning the count (threshold 0)
set -e
out=gpurun_out/r03_batch29
mkdir -p $out
timeout -k 10 400 python tools/sweep.py --ns 500 --rounds 4 "handoff=0" "handoff_pixels=0,handoff_poll=16" "handoff_pixels=0,handoff_poll=3" "handoff_pixels=0,handoff_poll=5" "handoff_pixels=0,handoff_poll=16,handoff_debug=2" "handoff_poll=3" "handoff_poll=5" "handoff_poll=5,handoff_pixels=16384" "handoff_poll=16,handoff_debug=2,handoff_pixels=16384" > $out/headline.log 2>&1; cat $out/headline.log
