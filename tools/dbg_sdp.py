"""Debug helper for the device SDP tests: the batches of tests/test_gpu_sdp.py one by one, with progress on stderr."""
import faulthandler, os, random, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import exonerate_amd as ex
import oracle_lib
import test_gpu_sdp as T
from test_library_fuzz_gpu import CODON

eng = ex.Engine(0)
rng = random.Random(12)
params = ex.default_params()
AA = T.AA
for mt, match, adv, w in (("affine:local", "dna2dna", (1, 1), 11), ("protein2dna", "protein2dna", (1, 3), 4)):
    model = ex.Model(mt, params=params)
    pairs, hsps = [], []
    for k in range(12):
        if mt == "affine:local":
            q = "".join(rng.choice("ACGT") for _ in range(rng.randint(300, 900)))
            t = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 800))) + T._mut(rng, q, 0.06, "ACGT") + \
                "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 800)))
            h = T._hsps(params, match, q, t, w)
        else:
            q = "".join(rng.choice(AA) for _ in range(rng.randint(80, 250)))
            coding = "".join(rng.choice(CODON[a]) for a in T._mut(rng, q, 0.05, AA))
            p = rng.randint(10, len(coding) - 10)
            coding = coding[:p] + rng.choice("ACGT") + coding[p:]
            t = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 500))) + coding + \
                "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 500)))
            rev = {c: a for a, cs in CODON.items() for c in cs}
            words = {}
            for i in range(len(q) - w + 1):
                words.setdefault(q[i:i + w], []).append(i)
            seeds = []
            for j in range(len(t) - 3 * w + 1):
                word = "".join(rev.get(t[j + 3 * x:j + 3 * x + 3], "X") for x in range(w))
                seeds += [(i, j) for i in words.get(word, ())]
            h = oracle_lib.hsp_set(params, match, q.encode(), t.encode(), w, 20, 30, seeds)
        if h:
            pairs.append((q, t)); hsps.append(h)
    print(mt, "pairs", len(pairs), "hsps", [len(h) for h in hsps], file=sys.stderr, flush=True)
    for i in range(len(pairs)):
        print("  single", i, len(pairs[i][0]), len(pairs[i][1]), len(hsps[i]), file=sys.stderr, flush=True)
        got = eng.sdp(model, pairs[i:i + 1], hsps[i:i + 1], adv[0], adv[1], 50, 50, 4)
        ub, exp = oracle_lib.sdp(model.c, model.params, pairs[i][0].encode(), pairs[i][1].encode(), hsps[i], adv[0], adv[1], 50, True, 50, 4)
        same = [a.as_dict() for a in got[0]] == exp
        print("    ->", len(got[0]), len(exp), "same" if same else "DIFFERENT", file=sys.stderr, flush=True)
    print(" batch", file=sys.stderr, flush=True)
    got = eng.sdp(model, pairs, hsps, adv[0], adv[1], 50, 50, 4)
    print(" batch done", [len(g) for g in got], file=sys.stderr, flush=True)
eng.close()
print("all done", file=sys.stderr)
