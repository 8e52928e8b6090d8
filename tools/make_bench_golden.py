#!/usr/bin/env python3
"""tests/golden/bench_configs.json: what bench.py's `configs` block checks its sampled alignments against — score,
region and operations of every 64th pair of BASELINE.json's configurations 2, 3 and 5 (exonerate_amd.workloads.bench_config)
from the CPU oracle (oracle/c4_oracle.c, pinned on the reference's own vectors: tests/test_oracle_golden.py).  Pairs against a
shared contig go through the size-independent window property (tests/test_gpu_configs.py): a local alignment whose path
lies inside a window of the contig is the window's alignment shifted by the window's offset; the window is the planted
gene with a margin.  Run in the build container (CPU only); tests/test_bench_golden.py checks the file against the
oracle again, tests/test_gpu_configs.py checks the device against it."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import exonerate_amd as ex
from exonerate_amd import workloads
import oracle_lib

MARGIN = {"c3": 1500, "c5": 1000}


def expected(name, every=64):
    model_name, pairs, places = workloads.bench_config(name)
    model = ex.Model(model_name)
    out = []
    for k in range(0, len(pairs), every):
        q, t = pairs[k]
        if places is None:
            exp = oracle_lib.find_path(model.c, model.params, q, t, dpmemory=32)
            w0 = 0
        else:
            g0, g1 = places[k]
            w0, w1 = max(0, g0 - MARGIN[name]), min(len(t), g1 + MARGIN[name])
            exp = oracle_lib.find_path(model.c, model.params, q, t[w0:w1], dpmemory=32)
        r = exp["region"]
        out.append({"pair": k, "score": exp["score"], "region": [r[0], r[1] + w0, r[2], r[3]], "ops": exp["ops"]})
    return {"model": model_name, "pairs": len(pairs), "every": every, "sample": out}


if __name__ == "__main__":
    doc = {name: expected(name) for name in ("c2", "c3", "c5")}
    path = os.path.join(ROOT, "tests", "golden", "bench_configs.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    print(path, {k: len(v["sample"]) for k, v in doc.items()})
