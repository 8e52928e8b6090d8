#!/bin/bash
# round 4, call j: what an SDP flush costs by arena size (C5's heuristic leg at three flush budgets)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4j; mkdir -p $OUT
for gb in 64 115 230; do
  C4GPU_SDP_GB=$gb timeout 600 python tools/trace_c5_heuristic.py 256 $OUT/c5t_$gb > $OUT/c5_trace_$gb.txt 2>&1
  rm -f $OUT/c5t_$gb/*.fa
  echo "== budget $gb GB"; grep -E '^== default|sdp flush|c4gpu sdp' $OUT/c5_trace_$gb.txt | head -6
  grep -h 'sdp arena\|sdp round' $OUT/c5t_$gb/default.err | head -12
done
