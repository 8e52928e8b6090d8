# round 3: the multi-GPU plumbing on RCCL with world size 1 + the default bench line
set -u
mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r3e/pytest.log 2>&1; echo "dist rc=$?"; tail -15 gpurun_out/r3e/pytest.log
python bench.py --steps 3 --warmup 1 --no-revcomp > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3e/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3e/bench.json').read().strip().splitlines()[-1])
print("ms_per_step %.1f" % d["ms_per_step"], d["work_queue"], d.get("cpu_baseline",{}).get("value"))
PY
