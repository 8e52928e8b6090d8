#!/bin/bash
# the default bench line twice on one box (run-to-run spread of the step and of config 5's heuristic leg)
mkdir -p gpurun_out/bench2
for k in 1 2; do
  timeout 900 python bench.py > gpurun_out/bench2/bench_$k.json 2> gpurun_out/bench2/bench_$k.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/bench2/bench_$k.json").read().strip().splitlines()[-1])
print("run $k", round(d["ms_per_step"], 1), "%.3e" % d["value"], "revcomp", (d.get("revcomp") or {}).get("value"), {k: (v.get("value"), v.get("wall_s"), v.get("wall_s_runs"), v.get("slow_run_trace")) for k, v in d["configs"].items()})
PY
done
