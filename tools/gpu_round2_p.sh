set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/sdp_repeat3.log
for i in $(seq 1 6); do
  timeout 300 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider >> gpurun_out/sdp_repeat3.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/sdp_repeat3.log
done
grep -c "rc=0" gpurun_out/sdp_repeat3.log; grep -n "rc=[1-9]\|Fatal\|failed" gpurun_out/sdp_repeat3.log | head
timeout 1800 python -m pytest tests/test_integration_gpu.py -m gpu -q -p no:cacheprovider -k "alphabet or sdp or c1 or byte_identical" > gpurun_out/pytest_gpu_p_int.log 2>&1
tail -8 gpurun_out/pytest_gpu_p_int.log
(cd /tmp && timeout 600 python $ROOT/tools/bench_sdp.py 100 > $ROOT/gpurun_out/sdp_bench.md 2> $ROOT/gpurun_out/sdp_bench.err)
cat gpurun_out/sdp_bench.md; tail -3 gpurun_out/sdp_bench.err
