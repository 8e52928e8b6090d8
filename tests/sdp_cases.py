"""Seeded random SDP cases shared by tests/test_gpu_sdp.py (the kernels, on the MI355X) and tests/test_sdp_sim.py (the same
per-lane code driven by CPU loops): pairs with HSPs grown from every shared word by the oracle's HSPset restatement
(pinned on reference HSPs)."""
import random

import exonerate_amd as ex
import oracle_lib
from golden_util import PARAM_VARIANTS, apply_flags

AA = "ARNDCQEGHILKMFPSTWYV"
FAMILY = {"dna": "affine", "protein": "affine", "p2d": "protein2dna", "e2g": "est2genome", "p2g": "protein2genome"}


def mut(rng, s, rate, alphabet):
    out = []
    for c in s:
        x = rng.random()
        if x < rate:
            out.append(rng.choice(alphabet))
        elif x < rate * 1.3:
            continue
        elif x < rate * 1.6:
            out.append(c + rng.choice(alphabet))
        else:
            out.append(c)
    return "".join(out)


def seeded_fuzz_cases(seed, rounds=2, pairs_per_round=10, qmax=500):
    """The three boundary-free families with non-default penalties, matrices, --extensionthreshold and thresholds, HSPs
    grown from every shared word (lowered HSP thresholds: many weak seeds)."""
    from test_library_fuzz_gpu import CODON
    rng = random.Random(500 + seed)
    rev = {c: a for a, cs in CODON.items() for c in cs}
    out = []
    for _ in range(rounds):
        variant = rng.choice([None, "altparams", "posgap"])
        params = ex.default_params() if variant is None else apply_flags(ex.default_params(), PARAM_VARIANTS[variant])
        kind = rng.choice(["dna", "protein", "p2d"])
        dropoff, threshold = rng.choice([12, 50, 120]), rng.choice([30, 80])
        if kind == "dna":
            model, match, adv, w, alpha = ex.Model("affine:local", params=params), "dna2dna", (1, 1), 10, "ACGT"
        elif kind == "protein":
            model = ex.Model("affine:local", query_alphabet=ex.ALPHABET_PROTEIN, target_alphabet=ex.ALPHABET_PROTEIN, params=params)
            match, adv, w, alpha = "protein2protein", (1, 1), 4, AA
        else:
            model, match, adv, w, alpha = ex.Model("protein2dna", params=params), "protein2dna", (1, 3), 4, AA
        pairs, hsps = [], []
        for k in range(pairs_per_round):
            q = "".join(rng.choice(alpha) for _ in range(rng.randint(60, qmax)))
            body = mut(rng, q, rng.choice([0.03, 0.1, 0.2]), alpha)
            if kind == "p2d":
                body = "".join(rng.choice(CODON[a]) for a in body)
                if rng.random() < 0.5:
                    p = rng.randint(5, len(body) - 5)
                    body = body[:p] + rng.choice("ACGT") + body[p:]
            flank = "ACGT" if kind != "protein" else AA
            t = "".join(rng.choice(flank) for _ in range(rng.randint(300, 900))) + body + \
                "".join(rng.choice(flank) for _ in range(rng.randint(0, 900)))
            if rng.random() < 0.3:
                t += body[len(body) // 3:]
            words = {}
            for i in range(len(q) - w + 1):
                words.setdefault(q[i:i + w], []).append(i)
            if kind == "p2d":
                seeds = []
                for j in range(len(t) - 3 * w + 1):
                    word = "".join(rev.get(t[j + 3 * x:j + 3 * x + 3], "X") for x in range(w))
                    seeds += [(i, j) for i in words.get(word, ()) if j - 3 * i + len(q) >= 0]
            else:
                seeds = [(i, j) for j in range(len(t) - w + 1) for i in words.get(t[j:j + w], ())]
            h = oracle_lib.hsp_set(params, match, q.encode(), t.encode(), w, rng.choice([10, 30]), rng.choice([15, 30]), seeds)
            if h:
                pairs.append((q, t)); hsps.append(h)
        if pairs:
            out.append(dict(model=model, kind=kind, variant=variant, pairs=pairs, hsps=hsps, adv=adv, dropoff=dropoff, threshold=threshold))
    return out


def boundary_fuzz_cases(seed, rounds=2, pairs_per_round=8, qmax=700):
    """The boundary flavour (est2genome, protein2genome): genes with several introns on either strand sense, a second copy
    of the gene, indels and frameshifts, non-default penalties and intron windows (so that stored span seeds expire),
    lowered --extensionthreshold; every shared word a word hit."""
    from test_library_fuzz_gpu import CODON
    rng = random.Random(900 + seed)
    rev = {c: a for a, cs in CODON.items() for c in cs}
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    out = []
    for _ in range(rounds):
        variant = rng.choice([None, "altparams", "tightintron"])
        params = ex.default_params() if variant is None else apply_flags(ex.default_params(), PARAM_VARIANTS[variant])
        kind = rng.choice(["e2g", "p2g"])
        dropoff, threshold = rng.choice([15, 50, 90]), rng.choice([40, 100])
        model = ex.Model("est2genome" if kind == "e2g" else "protein2genome", params=params)
        adv, w = ((1, 1), 10) if kind == "e2g" else ((1, 3), 4)
        pairs, hsps = [], []
        for k in range(pairs_per_round):
            if kind == "e2g":
                q = dna(rng.randint(150, qmax))
                cuts = sorted(rng.sample(range(30, len(q) - 30), rng.randint(1, 3)))
                revs = rng.random() < 0.3
                gene, last = "", 0
                for c in cuts + [len(q)]:
                    gene += mut(rng, q[last:c], 0.03, "ACGT")
                    if c < len(q):
                        gene += ("CT" if revs else "GT") + dna(rng.choice([30, 60, 150, 400])) + ("AC" if revs else "AG")
                    last = c
            else:
                q = "".join(rng.choice(AA) for _ in range(rng.randint(60, max(61, qmax * 220 // 700))))
                coding = "".join(rng.choice(CODON[a]) for a in mut(rng, q, 0.04, AA))
                cuts = sorted(rng.sample(range(20, len(coding) - 20), rng.randint(1, 3)))
                gene, last = "", 0
                for c in cuts + [len(coding)]:
                    gene += coding[last:c]
                    if c < len(coding):
                        gene += "GT" + dna(rng.choice([30, 60, 150, 400])) + "AG"
                    last = c
                if rng.random() < 0.3:
                    p = rng.randint(10, len(gene) - 10)
                    gene = gene[:p] + rng.choice("ACGT") + gene[p:]
            t = dna(rng.randint(300, 900)) + gene + dna(rng.randint(50, 600))
            if rng.random() < 0.3:
                t += gene[len(gene) // 3:] + dna(40)
            words = {}
            for i in range(len(q) - w + 1):
                words.setdefault(q[i:i + w], []).append(i)
            if kind == "p2g":
                seeds = []
                for j in range(len(t) - 3 * w + 1):
                    word = "".join(rev.get(t[j + 3 * x:j + 3 * x + 3], "X") for x in range(w))
                    seeds += [(i, j) for i in words.get(word, ()) if j - 3 * i + len(q) >= 0]
                h = oracle_lib.hsp_set(params, "protein2dna", q.encode(), t.encode(), w, 20, rng.choice([15, 30]), seeds)
            else:
                seeds = [(i, j) for j in range(len(t) - w + 1) for i in words.get(t[j:j + w], ())]
                h = oracle_lib.hsp_set(params, "dna2dna", q.encode(), t.encode(), w, 30, rng.choice([20, 40]), seeds)
            if h:
                pairs.append((q, t)); hsps.append(h)
        if pairs:
            out.append(dict(model=model, kind=kind, variant=variant, pairs=pairs, hsps=hsps, adv=adv, dropoff=dropoff, threshold=threshold))
    return out
