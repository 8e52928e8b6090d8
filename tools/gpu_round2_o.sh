set -u
ROOT=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_o.log 2>&1
tail -6 gpurun_out/pytest_gpu_o.log
bash tools/profile_round.sh r02_o > gpurun_out/prof_r02_o.log 2>&1
cd $ROOT
tail -c 900 gpurun_out/prof_r02_o/bench.json
(cd /tmp && timeout 1200 python $ROOT/tools/bench_configs.py > $ROOT/gpurun_out/configs_o.md 2> $ROOT/gpurun_out/configs_o.err)
cat gpurun_out/configs_o.md
(cd /tmp && timeout 1500 python $ROOT/tools/bench_heuristic.py 32 > $ROOT/gpurun_out/heuristic_o.md 2> $ROOT/gpurun_out/heuristic_o.err)
cat gpurun_out/heuristic_o.md; tail -3 gpurun_out/heuristic_o.err
(cd /tmp && timeout 900 python $ROOT/tools/bench_dropin.py 32 32 1 /tmp/dropin > $ROOT/gpurun_out/dropin_o.md 2> $ROOT/gpurun_out/dropin_o.err)
cat gpurun_out/dropin_o.md
