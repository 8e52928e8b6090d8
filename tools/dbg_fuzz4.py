import os, sys, random, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import exonerate_amd as ex, oracle_lib
import test_library_fuzz_gpu as lf
from golden_util import PARAM_VARIANTS, apply_flags
rng = random.Random(9000 + 4)
case = None
for _ in range(3):
    mt = rng.choice(lf.MODELS)
    variant = rng.choice([None, None, "altparams", "tightintron", "invertedintron", "posgap"])
    params = ex.default_params() if variant is None else apply_flags(ex.default_params(), PARAM_VARIANTS[variant])
    model = ex.Model(mt, params=params)
    pairs = lf._pairs(rng, mt)
    dpm = rng.choice([0, 1, 32]); thr = rng.choice([-987654321, 50, 200]); rounds = rng.choice([1, 1, 2, 3])
    if mt == "protein2genome" and variant == "tightintron":
        case = (mt, variant, model, pairs, dpm, thr, rounds)
mt, variant, model, pairs, dpm, thr, rounds = case
print("case", mt, variant, [(len(q), len(t)) for q, t in pairs], dpm, thr, rounds)
json.dump({"pairs": pairs, "dpm": dpm, "thr": thr, "rounds": rounds}, open("/tmp/fuzz4_case.json", "w"))
if len(sys.argv) > 1 and sys.argv[1] == "gpu":
    eng = ex.Engine(0)
    for env in ({}, {"C4GPU_MW": "0"}, {"C4GPU_LOCAL_EXACT": "0"}, {"C4GPU_PACK": "0"}, {"C4GPU_FORCE_SEQUENTIAL": "1"}):
        for k in ("C4GPU_MW", "C4GPU_LOCAL_EXACT", "C4GPU_PACK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for d in (dpm, 32):
            got = eng.find_path(model, pairs, dpmemory=d, threshold=max(thr, 40))
            sc = eng.find_score(model, pairs)
            exp = [oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=d, threshold=max(thr, 40)) for q, t in pairs]
            osc = [oracle_lib.find_score(model.c, model.params, q.encode(), t.encode()) for q, t in pairs]
            print(env, "dpm", d, "scores gpu", sc, "oracle", osc, "paths", [(a.score if a else None) for a in got], [(e["score"] if e else None) for e in exp],
                  "OK" if [(a.as_dict() if a else None) for a in got] == exp else "DIFF")
