"""tests/golden/bench_configs.json (what bench.py's `configs` block checks its sampled alignments against) was written by
tools/make_bench_golden.py from the reference itself (refdump --cmd golden on each sampled pair or window); the oracle gives
the same records: the cheap configuration is regenerated in full, one record of each of the others."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_bench_config_fixture_is_the_oracles():
    import make_bench_golden as mk
    import exonerate_amd as ex
    from exonerate_amd import workloads
    import oracle_lib
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_configs.json")))
    assert set(doc) == {"c2", "c3", "c5"}
    assert all(doc[k]["source"] == "reference" for k in doc)
    got = json.loads(json.dumps(mk.expected("c2", source="oracle")))
    assert got["sample"] == doc["c2"]["sample"] and got["pairs"] == doc["c2"]["pairs"]
    for name in ("c3",):                       # (c5's 10 Mb chromosome takes a minute to generate: its records are checked on the GPU box)
        model_name, pairs, places = workloads.bench_config(name)
        model = ex.Model(model_name)
        rec = doc[name]["sample"][1]
        q, t = pairs[rec["pair"]]
        g0, g1 = places[rec["pair"]]
        w0, w1 = max(0, g0 - mk.MARGIN[name]), min(len(t), g1 + mk.MARGIN[name])
        exp = oracle_lib.find_path(model.c, model.params, q, t[w0:w1], dpmemory=32)
        r = exp["region"]
        assert rec == {"pair": rec["pair"], "score": exp["score"], "region": [r[0], r[1] + w0, r[2], r[3]], "ops": exp["ops"]}
