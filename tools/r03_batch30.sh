#!/bin/bash
# round-3 batch 30: tail hand-off, polls only once a lane of the wave has found its queue empty: poll period x threshold
set -e
out=gpurun_out/r03_batch30
mkdir -p $out
timeout -k 10 600 python tools/sweep.py --ns 500 --rounds 4 "handoff=0" "handoff_pixels=0,handoff_poll=3" "handoff_poll=2,handoff_pixels=16384" "handoff_poll=3,handoff_pixels=16384" "handoff_poll=4,handoff_pixels=16384" "handoff_poll=5,handoff_pixels=16384" "handoff_poll=3,handoff_pixels=8192" "handoff_poll=3,handoff_pixels=24576" "handoff_poll=3,handoff_pixels=32768" "handoff_poll=3,handoff_pixels=49152" > $out/headline.log 2>&1; cat $out/headline.log
