#!/usr/bin/env python3
"""tests/golden/bench_configs.json: what bench.py's `configs` block checks its sampled alignments against — score,
region and operations of every 64th pair of BASELINE.json's configurations 2, 3 and 5 (exonerate_amd.workloads.bench_config)
from the REFERENCE ITSELF (oracle/_ref/refdump --cmd golden: the reference's Optimal_find_path on the pair, as for every
other vector under tests/golden/; `--source oracle` writes the same records from oracle/c4_oracle.c instead, and
tests/test_bench_golden.py holds the two equal).  Pairs against a
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

REF_FLAGS = {"est2genome": ("--withsplice", "yes")}

MARGIN = {"c3": 1500, "c5": 1000}


def windows(name, every=64):
    """(pair index, query, the target or its window, the window's offset) of every `every`th pair."""
    model_name, pairs, places = workloads.bench_config(name)
    out = []
    for k in range(0, len(pairs), every):
        q, t = pairs[k]
        w0, w1 = 0, len(t)
        if places is not None:
            g0, g1 = places[k]
            w0, w1 = max(0, g0 - MARGIN[name]), min(len(t), g1 + MARGIN[name])
        out.append((k, q, t[w0:w1], w0))
    return model_name, len(pairs), out


def expected(name, every=64, source="reference"):
    model_name, n_pairs, wins = windows(name, every)
    if source == "reference":
        import make_golden
        text = lambda x: x.decode() if isinstance(x, bytes) else x
        recs = make_golden.run(model_name, [("p%d" % k, text(q), text(t)) for k, q, t, _ in wins], 32,
                               REF_FLAGS.get(model_name, ()))
    else:
        model = ex.Model(model_name)
        recs = [oracle_lib.find_path(model.c, model.params, q, t, dpmemory=32) for _, q, t, _ in wins]
    out = []
    for (k, _, _, w0), exp in zip(wins, recs):
        r = exp["region"]
        out.append({"pair": k, "score": exp["score"], "region": [r[0], r[1] + w0, r[2], r[3]], "ops": exp["ops"]})
    return {"model": model_name, "pairs": n_pairs, "every": every, "source": source, "sample": out}


if __name__ == "__main__":
    source = "oracle" if "--source=oracle" in sys.argv or sys.argv[1:3] == ["--source", "oracle"] else "reference"
    doc = {name: expected(name, source=source) for name in ("c2", "c3", "c5")}
    path = os.path.join(ROOT, "tests", "golden", "bench_configs.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    print(path, {k: len(v["sample"]) for k, v in doc.items()})
