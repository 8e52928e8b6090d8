#!/bin/bash
# whole GPU suite + smoke on the current tree (usage through gpurun: bash tools/gpu_suite_only.sh <tag>), then the dump-interval sweep
TAG=${1:-suite}
mkdir -p gpurun_out/$TAG
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -16 gpurun_out/$TAG/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -1 gpurun_out/$TAG/smoke.log
bash tools/gpu_kshift_sweep.sh
