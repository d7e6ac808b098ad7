#!/bin/bash
# round-3 batch 25: tail hand-off: which part of the mechanism costs the main kernel its time (threshold 0 = no hand-off happens)
set -e
out=gpurun_out/r03_batch25
mkdir -p $out
timeout -k 10 400 python tools/sweep.py --ns 500 --rounds 3 "handoff=0" "handoff_pixels=0" "handoff_pixels=0,handoff_debug=2" "handoff_pixels=0,handoff_debug=4" "handoff_pixels=0,handoff_debug=6" "handoff_pixels=0,handoff_poll=6" "handoff_poll=6" "handoff_poll=8" > $out/headline.log 2>&1; cat $out/headline.log
