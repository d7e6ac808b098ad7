#!/bin/bash
# round-3 batch 3: per-launch timelines (headline whole, headline 1/8 share; with and without the cost prior)
set -e
out=gpurun_out/r03_batch3
mkdir -p $out
export TMPDIR=/tmp
for cfg in "whole:1:" "whole_noprior:1:prior=0" "eighth:8:" "eighth_noprior:8:prior=0"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; stride=${rest%%:*}; opts=${rest#*:}
  STRIDE=$stride RT_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/one_frame.py > $out/$tag.log 2>&1
  python3 tools/timeline_from_trace.py $out/trace_$tag > $out/timeline_$tag.txt 2>&1 || true
  tail -1 $out/$tag.log; cat $out/timeline_$tag.txt
  rm -rf $out/trace_$tag
done
