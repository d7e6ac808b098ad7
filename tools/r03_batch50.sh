#!/bin/bash
# round-3 batch 50: per-launch timelines of the headline frame's 1/2 and 1/4 shares
set -e
out=gpurun_out/r03_batch50
mkdir -p $out
export TMPDIR=/tmp
for cfg in "half:2:" "quarter:4:"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; stride=${rest%%:*}; opts=${rest#*:}
  STRIDE=$stride RT_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/one_frame.py > $out/$tag.log 2>&1
  python3 tools/timeline_from_trace.py $out/trace_$tag > $out/timeline_$tag.txt 2>&1 || true
  echo "== $tag"; grep -E "^(main|tier) " $out/timeline_$tag.txt
  rm -rf $out/trace_$tag
done
