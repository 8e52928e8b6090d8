set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_b.log 2>&1
tail -40 gpurun_out/pytest_gpu_b.log
(cd /tmp && timeout 900 python $ROOT/bench.py --steps 3 --warmup 1 > $ROOT/gpurun_out/bench_r02_b.json 2> $ROOT/gpurun_out/bench_r02_b.err)
tail -c 3000 gpurun_out/bench_r02_b.json
