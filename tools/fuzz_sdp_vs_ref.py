#!/usr/bin/env python3
"""Differential fuzz of the oracle's SDP restatement against the REFERENCE ITSELF (oracle/_ref/refdump --cmd sdp).
Build container only (needs oracle/_ref); nothing is written.  usage: tools/fuzz_sdp_vs_ref.py [first seed] [seeds]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exonerate_amd as ex
import oracle_lib
import make_golden as mg
from golden_util import PARAM_VARIANTS, apply_flags

MODELS = [("affine:local", None, (1, 1), ("--dnawordlen", "10")), ("affine:local:protein", ex.ALPHABET_PROTEIN, (1, 1), ("--proteinwordlen", "5")),
          ("est2genome", None, (1, 1), ("--dnawordlen", "10")), ("protein2dna", None, (1, 3), ("--proteinwordlen", "4")),
          ("protein2genome", None, (1, 3), ("--proteinwordlen", "4"))]


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = total = 0
    for seed in range(first, first + count):
        rng = random.Random(4242 + seed)
        mname, alpha, adv, extra = MODELS[seed % len(MODELS)]
        variant = rng.choice([None, "altparams", "posgap", "posgap"])
        drop = rng.choice([50, 50, 12, 25, 100])
        thr = rng.choice([30, 40, 80])
        hspthr = rng.choice([None, "20", "40"])
        flags = list(extra) + ["--extensionthreshold", str(drop), "--suboptmax", "5", "--suboptthreshold", str(thr)]
        if variant:
            flags += list(PARAM_VARIANTS[variant])
        if hspthr:
            flags += ["--dnahspthreshold", hspthr, "--proteinhspthreshold", hspthr]
        cases = mg.sdp_cases(mname, rng.choice([6, 10]), seed * 7 + 1)
        recs = mg.run_sdp(mname, cases, flags)
        par, recs = recs[0]["params"], recs[1:]
        params = ex.default_params() if variant is None else apply_flags(ex.default_params(), PARAM_VARIANTS[variant])
        mt = mname.replace(":protein", "")
        model = ex.Model(mt, query_alphabet=alpha, target_alphabet=alpha if mname.endswith(":protein") else None, params=params)
        for r in recs:
            ub, got = oracle_lib.sdp(model.c, model.params, r["query"].encode(), r["target"].encode(), r["hsps"], adv[0], adv[1],
                                     par["dropoff"], True, par["threshold"], 5, qid=r["id"])
            exp = [(a["path_score"], a["region"], a["ops"]) for a in r["alignments"]]
            g = [(a["score"], a["region"], a["ops"]) for a in got]
            total += 1
            if g != exp:
                bad += 1
                print("DIFF seed", seed, mname, variant, drop, thr, r["id"], len(exp), len(g))
    print("pairs", total, "differences", bad)


if __name__ == "__main__":
    main()
