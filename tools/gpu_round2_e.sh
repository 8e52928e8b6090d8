set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_e.log 2>&1
tail -12 gpurun_out/pytest_gpu_e.log
bash tools/profile_round.sh r02_e > gpurun_out/prof_r02_e.log 2>&1
cd $ROOT
tail -c 700 gpurun_out/prof_r02_e/bench.json
for K in 11 12 13 14; do
  (cd /tmp && C4GPU_SEED_KSHIFT=$K C4GPU_TRACE=1 timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --no-revcomp --no-cpu-baseline > $ROOT/gpurun_out/bench_k$K.json 2> $ROOT/gpurun_out/bench_k$K.err)
  python -c "
import json;d=json.loads(open('gpurun_out/bench_k$K.json').read().strip().splitlines()[-1]);print('K',$K,d['value'],d['ms_per_step'],d['kernel_ms'])"
done
(cd /tmp && timeout 1200 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/configs_windowed.md 2> $ROOT/gpurun_out/configs_windowed.err)
(cd /tmp && C4GPU_WINDOWED=0 timeout 1200 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/configs_onepass.md 2> $ROOT/gpurun_out/configs_onepass.err)
cat gpurun_out/configs_windowed.md gpurun_out/configs_onepass.md
(cd /tmp && timeout 1500 python $ROOT/tools/bench_heuristic.py 16 > $ROOT/gpurun_out/heuristic.md 2> $ROOT/gpurun_out/heuristic.err)
cat gpurun_out/heuristic.md; tail -3 gpurun_out/heuristic.err
