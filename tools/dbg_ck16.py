import os, random, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import exonerate_amd as ex
from test_gpu_kernel_variants import _seeded_pairs, _rand, _mutate
dpm = 32
rng = random.Random(4100 + dpm)
model = ex.Model("est2genome")
pairs = []
sizes = [(900, 30000), (400, 52000), (1000, 9000), (640, 30000), (130, 20000), (777, 41000), (1300, 7000), (190, 2500),
         (64, 3000), (1000, 100000)]
for ql, tl in sizes:
    pairs += _seeded_pairs(rng, "est2genome", ql, tl, 1)
q = _rand(rng, 800)
pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
pairs.append((_rand(rng, 500), _rand(rng, 25000)))
os.environ["C4GPU_TRACE"] = "1"
eng = ex.Engine(0)
res = {}
for ck in ("0", "1"):
    os.environ["C4GPU_CK16"] = ck
    sys.stderr.write("=== CK16=%s\n" % ck)
    res[ck] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=dpm, threshold=30)]
for k, (a, b) in enumerate(zip(res["0"], res["1"])):
    print(k, len(pairs[k][0]), len(pairs[k][1]), "same" if a == b else "DIFF", a and a["score"], a and a["region"] if a else None)
