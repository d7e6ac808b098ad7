#!/bin/bash
# round-3 batch 44: per-launch timeline of the headline frame and its 1/8 share with the final hand-off defaults
set -e
out=gpurun_out/r03_batch44
mkdir -p $out
export TMPDIR=/tmp
for cfg in "whole:1:" "eighth:8:" "hd_eighth:8:"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; stride=${rest%%:*}; opts=${rest#*:}
  if [ $tag = hd_eighth ]; then export NX=1920 NY=1080; fi
  STRIDE=$stride RT_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/one_frame.py > $out/$tag.log 2>&1
  python3 tools/timeline_from_trace.py $out/trace_$tag > $out/timeline_$tag.txt 2>&1 || true
  echo "== $tag"; grep -E "^(main|tier|kernel|prior|rank|collect)" $out/timeline_$tag.txt
  rm -rf $out/trace_$tag
done
