#!/usr/bin/env python3
"""tests/golden/stale_tails_pair214.json: the oracle's sub-optimal loop (two paths, -D 32, threshold 300) over pair 214 of the
north-star batch (1 kb x 100 kb), what tests/test_gpu_parity.py::test_stale_sub_alignment_tails_are_recomputed compares the device
with -- 45 s of CPU here instead of on the GPU box in every run of the suite."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import exonerate_amd as ex
from exonerate_amd import workloads
import oracle_lib
model = ex.Model("est2genome")
q, t = workloads.est2genome_pairs(1, 1000, 100000, first=214)[0]
exp = oracle_lib.find_paths_subopt(model.c, model.params, q, t, 32, 300, 2)
json.dump({"pair": 214, "dpmemory": 32, "threshold": 300, "max_paths": 2, "alignments": [d for d, _ in exp]},
          open(os.path.join(ROOT, "tests", "golden", "stale_tails_pair214.json"), "w"), indent=1)
print(len(exp), "alignments")
