#!/bin/bash
# after the read-back and retire changes: headline step + both strands, the drop-in's config 5 heuristic leg, quick suites
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 --no-configs --no-cpu-baseline > gpurun_out/dl_0.json 2> gpurun_out/dl_0.err
python - <<P
import json
d=json.loads(open("gpurun_out/dl_0.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"],1), "value %.3e" % d["value"], "revcomp", {k:(round(v,1) if isinstance(v,float) else v) for k,v in d.get("revcomp",{}).items() if k in ("ms_per_step","value")}, "per step", d.get("rank0_step_ms"))
P
python tools/gpu_c5_cold_trace.py
grep "c4gpu sdp:\|c4gpu seed:\|c4gpu hsp:" gpurun_out/c5_cold_trace.log | tail -3
python -m pytest tests/test_gpu_sdp.py tests/test_gpu_hsp.py tests/test_gpu_seed.py tests/test_gpu_stage.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
