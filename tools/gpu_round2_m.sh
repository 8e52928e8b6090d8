set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/sdp_repeat.log
for i in 1 2 3 4 5 6 7 8; do
  MALLOC_CHECK_=3 timeout 300 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider >> gpurun_out/sdp_repeat.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/sdp_repeat.log
done
grep -n "rc=\|passed\|failed\|Fatal\|malloc\|free()\|corrupt" gpurun_out/sdp_repeat.log | head -40
(cd /tmp && timeout 600 python $ROOT/tools/bench_sdp.py 100 > $ROOT/gpurun_out/sdp_bench.md 2> $ROOT/gpurun_out/sdp_bench.err)
cat gpurun_out/sdp_bench.md; tail -3 gpurun_out/sdp_bench.err
