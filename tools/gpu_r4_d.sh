#!/bin/bash
# round 4, call d: which part of the staged packed score pass is wrong?  (three builds: all of it, without the profile, without the stage)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4d; mkdir -p $OUT
for n in FULL NOPROF NOSTAGE; do
  C4GPU_LIB=$ROOT/exonerate_amd/alt/libc4gpu_$n.so C4GPU_TRACE=1 timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -m gpu -x -q -k "test_windowed_region_pass_matches_oracle and est2genome-600-6000-32-3" > $OUT/$n.log 2>&1
  echo "== $n"; grep -h 'score pass\|passed\|failed' $OUT/$n.log | head -12
done
