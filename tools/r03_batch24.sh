#!/bin/bash
# round-3 batch 24: tail hand-off: where the time goes (threshold 0 = the mechanism without a hand-off), per-kernel times
set -e
out=gpurun_out/r03_batch24
mkdir -p $out
timeout -k 10 300 python tools/sweep.py --ns 500 --rounds 3 "handoff=0" "handoff_pixels=0" "handoff_pixels=256" "handoff_pixels=2048" "" > $out/headline.log 2>&1; cat $out/headline.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/one_frame.py > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profile.py $out/trace > $out/trace_summary.txt 2>&1 || true
head -30 $out/trace_summary.txt
