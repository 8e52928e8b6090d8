#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4n; mkdir -p $OUT
for v in dp0; do
  C4GPU_LIB=$ROOT/exonerate_amd/alt/libc4gpu_$v.so timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests_dbg/test_zz_debug.py -m gpu -q -x > $OUT/dbg_$v.log 2>&1
  echo "after kv, $v:"; grep -n 'DEBUG RESULTS' -A2 $OUT/dbg_$v.log | tail -2
done
