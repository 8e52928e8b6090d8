set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/sdp_repeat4.log
for i in 1 2 3; do
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_sdp.py -m gpu -q -p no:cacheprovider >> gpurun_out/sdp_repeat4.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/sdp_repeat4.log
done
grep -n "rc=\|passed\|failed\|Fatal" gpurun_out/sdp_repeat4.log | head; grep -n "AssertionError:" gpurun_out/sdp_repeat4.log | head -5
timeout 1800 python -m pytest tests/test_integration_gpu.py tests/test_integration_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k "sdp or c1 or heuristic" > gpurun_out/pytest_gpu_r_int.log 2>&1
tail -4 gpurun_out/pytest_gpu_r_int.log
(cd /tmp && timeout 600 python $ROOT/tools/bench_sdp.py 100 > $ROOT/gpurun_out/sdp_bench.md 2> $ROOT/gpurun_out/sdp_bench.err)
cat gpurun_out/sdp_bench.md | head -8
(cd /tmp && timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --no-revcomp --no-cpu-baseline > $ROOT/gpurun_out/bench_r.json 2>/dev/null)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'staging_ms', d.get('staging_ms'), 'incl', d.get('value_incl_staging'))
PY
