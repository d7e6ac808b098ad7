#!/bin/bash
# round-3 batch 39: per-launch timelines of Book-2 final's 1/8 share and whole frame with and without the tail hand-off
set -e
out=gpurun_out/r03_batch39
mkdir -p $out
export TMPDIR=/tmp SCENE=final NX=800 NY=800 NS=200
for cfg in "eighth_off:8:handoff=0" "eighth_on:8:" "whole_off:1:handoff=0" "whole_on:1:"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; stride=${rest%%:*}; opts=${rest#*:}
  STRIDE=$stride RT_OPTS=$opts rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/one_frame.py > $out/$tag.log 2>&1
  python3 tools/timeline_from_trace.py $out/trace_$tag > $out/timeline_$tag.txt 2>&1 || true
  echo "== $tag"; grep -v "^W2026" $out/$tag.log | tail -1; grep -E "^(main|tier|kernel)" $out/timeline_$tag.txt
  rm -rf $out/trace_$tag
done
