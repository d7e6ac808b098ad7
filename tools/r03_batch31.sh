#!/bin/bash
# round-3 batch 31: tail hand-off: poll period x threshold, second pass
set -e
out=gpurun_out/r03_batch31
mkdir -p $out
timeout -k 10 600 python tools/sweep.py --ns 500 --rounds 4 "handoff=0" "handoff_poll=5,handoff_pixels=16384" "handoff_poll=6,handoff_pixels=16384" "handoff_poll=7,handoff_pixels=16384" "handoff_poll=8,handoff_pixels=16384" "handoff_poll=6,handoff_pixels=12288" "handoff_poll=6,handoff_pixels=24576" "handoff_poll=7,handoff_pixels=24576" "handoff_poll=7,handoff_pixels=32768" "handoff_poll=6,handoff_pixels=8192" > $out/headline.log 2>&1; cat $out/headline.log
