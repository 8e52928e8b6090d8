#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE ITSELF (oracle/_ref/refdump,
built from /root/reference by `make -C oracle ref`).  Runs only in the build container; the vectors it
writes are small, committed, and are what pins the oracle (oracle/c4_oracle.c) and the HIP engine.

Inputs are seeded synthetic sequences (generator below, no reference code) plus the hard-coded inputs of
the reference's own model known-answer tests (src/model/affine.test.c:33-38, est2genome.test.c:24-37,
protein2dna.test.c:58-70, protein2genome.test.c:62-75) so those KAT scores (-151/18/32/18, 157, 134, 125) are
part of the vectors (records kat_affine, kat_est2genome, kat_protein2dna, kat_protein2genome).
"""
import json, os, random, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDUMP = os.path.join(ROOT, "oracle", "_ref", "refdump")
OUT = os.path.join(ROOT, "tests", "golden")
AA = "ARNDCQEGHILKMFPSTWYV"
CODON = {}
_ncbi = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
for i, a in enumerate("TCAG"):
    for j, b in enumerate("TCAG"):
        for k, c in enumerate("TCAG"):
            CODON.setdefault(_ncbi[i * 16 + j * 4 + k], []).append(a + b + c)


def rand_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate, alphabet):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice(alphabet))
        elif r < 2 * rate / 3:
            out.append(ch)
            out.append(rng.choice(alphabet))
        elif r < rate:
            pass
        else:
            out.append(ch)
    return "".join(out)


def dna_pairs(rng, n):
    cases = []
    for k in range(n):
        ql = rng.choice([0, 1, 2, 5, 13, 40, 77, 150, 300])
        q = rand_dna(rng, ql, "ACGT" if k % 4 else "ACGTN")
        mode = k % 5
        if mode == 0:
            t = rand_dna(rng, rng.choice([0, 1, 3, 30, 200]))
        elif mode == 1:
            t = rand_dna(rng, rng.randint(0, 40)) + mutate(rng, q, 0.1, "ACGT") + rand_dna(rng, rng.randint(0, 40))
        elif mode == 2:
            t = q
        elif mode == 3:
            t = "N" * rng.randint(1, 60)
        else:
            t = mutate(rng, q, 0.3, "ACGT")
        cases.append(("dna%03d" % k, q, t))
    return cases


def protein_pairs(rng, n):
    cases = []
    for k in range(n):
        ql = rng.choice([1, 2, 7, 30, 90, 200])
        q = rand_dna(rng, ql, AA)
        t = rand_dna(rng, rng.randint(0, 30), AA) + mutate(rng, q, 0.15, AA) + rand_dna(rng, rng.randint(0, 30), AA)
        if k % 6 == 5:
            t = rand_dna(rng, rng.randint(1, 50), AA)
        cases.append(("prot%03d" % k, q, t))
    return cases


def est_pairs(rng, n):
    """cDNA vs genomic with GT..AG introns (some on the reverse gene strand: CT..AC)."""
    cases = []
    for k in range(n):
        ql = rng.choice([30, 60, 120, 200, 320])
        q = rand_dna(rng, ql)
        nint = rng.choice([0, 1, 1, 2, 3])
        cuts = sorted(rng.sample(range(5, max(6, ql - 5)), min(nint, max(0, ql - 11)))) if ql > 12 else []
        rev = (k % 3 == 2)
        pieces, last = [], 0
        for c in cuts + [ql]:
            pieces.append(q[last:c])
            last = c
        t = rand_dna(rng, rng.randint(0, 50))
        for i, ex in enumerate(pieces):
            t += mutate(rng, ex, 0.03, "ACGT")
            if i + 1 < len(pieces):
                ilen = rng.choice([20, 35, 60, 150, 400])
                body = rand_dna(rng, max(0, ilen - 4))
                t += ("CT" + body + "AC") if rev else ("GT" + body + "AG")
        t += rand_dna(rng, rng.randint(0, 50))
        if k % 7 == 6:
            t = t.replace("A", "N", 3)
        cases.append(("est%03d" % k, q, t))
    return cases


def p2d_pairs(rng, n):
    cases = []
    for k in range(n):
        ql = rng.choice([3, 10, 40, 90, 160])
        q = rand_dna(rng, ql, AA)
        coding = "".join(rng.choice(CODON[a]) for a in mutate(rng, q, 0.08, AA))
        if k % 3 == 1 and len(coding) > 20:      # frameshift
            p = rng.randint(5, len(coding) - 5)
            coding = coding[:p] + rng.choice("ACGT") + coding[p:]
        if k % 3 == 2 and len(coding) > 20:
            p = rng.randint(5, len(coding) - 5)
            coding = coding[:p] + coding[p + 2:]
        t = rand_dna(rng, rng.randint(0, 60)) + coding + rand_dna(rng, rng.randint(0, 60))
        if k % 8 == 7:
            t = rand_dna(rng, rng.randint(0, 5))
        cases.append(("p2d%03d" % k, q, t))
    return cases


def p2g_pairs(rng, n):
    """protein vs genomic: coding sequence interrupted by GT..AG introns at codon phase 0, 1 or 2."""
    cases = []
    for k in range(n):
        ql = rng.choice([8, 25, 60, 120])
        q = rand_dna(rng, ql, AA)
        coding = "".join(rng.choice(CODON[a]) for a in mutate(rng, q, 0.05, AA))
        nint = rng.choice([0, 1, 1, 2, 3])
        cuts = sorted(rng.sample(range(4, max(5, len(coding) - 4)), min(nint, max(0, len(coding) - 9)))) \
            if len(coding) > 10 else []
        t = rand_dna(rng, rng.randint(0, 40))
        last = 0
        for c in cuts:
            t += coding[last:c] + "GT" + rand_dna(rng, rng.choice([26, 40, 90, 200])) + "AG"
            last = c
        t += coding[last:] + rand_dna(rng, rng.randint(0, 40))
        if k % 5 == 4 and len(t) > 30:          # frameshift as well
            p = rng.randint(10, len(t) - 10)
            t = t[:p] + rng.choice("ACGT") + t[p:]
        cases.append(("p2g%03d" % k, q, t))
    return cases


KAT_AFFINE = ("kat_affine", "MEEPQSDPSVEPPLSQETFSDLWKLL",
              "PENNVLSPLPSQAMDDLMLSPDDIEQWFTEDPGPEHSCETFDIWKWCPIECDFLNVISEPNEPIPSQ")
KAT_E2G = ("kat_est2genome", "CGATCGATCGNATCGATCGATC" "CATCTATCTAGCGAGCGATCTA",
           "CGATCGATCGATCGATCGATC" "GT" + "N" * 20 + "N" * 47 * 3 + "N" * 27 + "AG" + "CATCTATCTANNNGCGAGCGATCTA")


# src/model/protein2dna.test.c:58-70 (score 134) and protein2genome.test.c:62-75 (score 125): inputs only
KAT_P2D = ("kat_protein2dna", "NNNNNNMADQLTEQIAEFKEAFSLFDKDG" "TVHNC" "X" "WYFSGRW" "DGTITT",
           "ATGGCTGACCAGCTGACTGAGGAGCAGATT" "GCAGAGTTCNAAGGAGGCCTTCTCCCTCTTT" "GACAAGGATGGA"
           "NNACTGTCCATAATTGC" "TGGTACTTCAGCGGTCGATGG" "GATGGCACTCTGACCACC")
KAT_P2G = ("kat_protein2genome", "MADQLTEQIAEFKEAFSLFDKDGDGTITT",
           "ATGGCTGACCAGCTGACTGAGCAGATT" "GCAGAGTTCAA" "GT" + "N" * 42 + "AG" + "GGAGGCCTTCTCCCTCTTT"
           "GACAAGGATGGAGATGGCACTATTACCACC")


def repeat_pairs(rng, n, kind):
    """Targets holding several diverged copies of the query: successive sub-optimal alignments
    (GAM_Result_exhaustive_create's loop) each find the next copy, or a second path through the same one."""
    cases = []
    for k in range(n):
        if kind == "dna":
            q = rand_dna(rng, rng.choice([25, 40, 60, 90]))
            copies = [mutate(rng, q, r, "ACGT") for r in rng.sample([0.0, 0.05, 0.1, 0.2, 0.3], rng.randint(1, 3))]
            if k % 4 == 3:
                copies.append(q[len(q) // 3:])          # partial copy
            t = rand_dna(rng, rng.randint(0, 20))
            for c in copies:
                t += c + rand_dna(rng, rng.randint(0, 25))
            if k % 5 == 4:
                t = q + q                                # exact tandem repeat: ties everywhere
        elif kind == "est":
            base = est_pairs(rng, 2)
            q = base[0][1]
            t = base[0][2] + rand_dna(rng, 30) + mutate(rng, base[0][2], 0.05, "ACGT")
            if k % 3 == 2:
                t += rand_dna(rng, 10) + q               # an intron-less (processed) copy
        elif kind == "p2d":
            base = p2d_pairs(rng, 1)[0]
            q = base[1]
            t = base[2] + rand_dna(rng, 21) + mutate(rng, base[2], 0.04, "ACGT")
        else:
            base = p2g_pairs(rng, 1)[0]
            q = base[1]
            t = base[2] + rand_dna(rng, 21) + mutate(rng, base[2], 0.03, "ACGT")
        cases.append(("%ssub%03d" % (kind, k), q, t))
    return cases


sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from golden_util import PARAM_VARIANTS          # the flag lists live next to the code that mirrors them in the tests
ALT_FLAGS = PARAM_VARIANTS["altparams"]


def run(model, cases, dpmemory, extra=()):
    # (a case may carry a fourth entry: the query's CDS annotation (cds_start, cds_length) -- exonerate's --annotation)
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        for c in cases:
            f.write("%s\t%s\t%s" % c[:3] + ("\t%d:%d" % c[3] if len(c) > 3 else "") + "\n")
        path = f.name
    cmd = [REFDUMP, "--cmd", "golden", "--model", model, "--input", path, "-D", str(dpmemory)] + list(extra)
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    os.unlink(path)
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(recs) == len(cases), (model, len(recs), len(cases))
    for r, c in zip(recs, cases):
        assert r["id"] == c[0]
        r["query"], r["target"], r["dpmemory"] = c[1], c[2], dpmemory
        if len(c) > 3:
            r["cds"] = list(c[3])
    return recs


def run_span(cases, match_state, span_state, model="est2genome"):
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        for cid, q, t in cases:
            f.write("%s\t%s\t%s\n" % (cid, q, t))
        path = f.name
    cmd = [REFDUMP, "--cmd", "span", "--model", model, "--input", path, "--derived", "%d,%d" % (match_state, span_state)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    os.unlink(path)
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(recs) == len(cases)
    for r, (cid, q, t) in zip(recs, cases):
        assert r["id"] == cid
        r["query"], r["target"] = q, t
    return recs


def hsp_cases(kind, n, seed):
    """Pairs with shared words and every shared-word position as a seed, in target-scan order (the order the FSM scan
    reports them): `kind` dna2dna / protein2protein / protein2dna.  Plus the inputs of src/comparison/hspset.test.c."""
    rng = random.Random(seed)
    w = 12 if kind == "dna2dna" else 6
    cases = []
    for k in range(n):
        if kind == "dna2dna":
            q = rand_dna(rng, rng.choice([60, 150, 400]), "ACGT" if k % 5 else "ACGTN")
            t = rand_dna(rng, rng.randint(0, 80)) + mutate(rng, q, rng.choice([0.02, 0.06, 0.12]), "ACGT") + rand_dna(rng, rng.randint(0, 80))
            if k % 4 == 3:
                t += mutate(rng, q[len(q) // 3:], 0.04, "ACGT")          # a second copy on other diagonals
            trans = lambda s: s
            step = 1
        else:
            q = rand_dna(rng, rng.choice([30, 80, 200]), AA)
            m = mutate(rng, q, rng.choice([0.03, 0.1]), AA)
            if kind == "protein2protein":
                t = rand_dna(rng, rng.randint(0, 30), AA) + m + rand_dna(rng, rng.randint(0, 30), AA)
                trans = lambda s: s
                step = 1
            else:
                t = rand_dna(rng, rng.randint(0, 40)) + "".join(rng.choice(CODON[a]) for a in m) + rand_dna(rng, rng.randint(0, 40))
                rev = {c: a for a, cs in CODON.items() for c in cs}
                trans = lambda s: "".join(rev.get(s[i:i + 3], "X") for i in range(0, len(s) - 2, 3))
                step = 3
        seeds = []
        words = {}
        for i in range(len(q) - w + 1):
            words.setdefault(q[i:i + w], []).append(i)
        for j in range(0, len(t) - w * step + 1):
            word = trans(t[j:j + w * step])
            for i in words.get(word, ()):
                seeds.append("%d:%d" % (i, j))
        if seeds:
            cases.append(("hsp_%s%03d" % (kind[:3], k), q, t, ",".join(seeds)))
    if kind == "dna2dna":       # hspset.test.c:49-66
        cases.append(("kat_d2d", "AAAAGTGAGAGAGAGAGAGAGGCGAAAAAAAAAACCCCCCCCCCACCCCGCGA",
                      "TTTTGTGAGAGTGTGAGAGAGGCGTTTTTTTTTTCCCCCCCCCCTCCCCGCCT", "8:8,36:36"))
    if kind == "protein2dna":
        cases.append(("kat_p2d", "PNKDEGSCPIECDFLCRHQYISDP",
                      "ACGTACGTACGTACGAGTGCGTGCCCCCTTNNNTGTGACTACATCTGCAAAACGTACGTACGT", "8:24"))
    return cases


def seed_cases(kind, n, seed):
    """Records for the seeder's walk (refdump --cmd seeds): several queries per seeder -- some share words, so that a word's seed
    list holds seeds of several queries -- and one target that holds mutated copies of them, residues outside the alphabet
    (N / X: the walk's reset), lower case (soft-masked: Sequence_mask) and, for DNA, a low-complexity stretch whose words repeat."""
    rng = random.Random(seed)
    cases = []
    for k in range(n):
        if kind == "dna2dna":
            q0 = rand_dna(rng, rng.choice([40, 90, 200]))
            qs = [q0, rand_dna(rng, rng.randint(20, 60)) + q0[len(q0) // 3:len(q0) // 3 + 30] + rand_dna(rng, rng.randint(0, 40))]
            if k % 3 == 0:
                qs.append(rand_dna(rng, 25) + "ACACACACACACACACACAC" + rand_dna(rng, 10))
            t = rand_dna(rng, rng.randint(0, 60)) + mutate(rng, q0, rng.choice([0.0, 0.03, 0.08]), "ACGT") + "NNN" + rand_dna(rng, rng.randint(5, 80))
            t += qs[1][5:] + rand_dna(rng, 17).lower() + mutate(rng, q0[10:], 0.02, "ACGT")
            if k % 3 == 0:
                t += "ACACACACACACACACACACACACAC" + rand_dna(rng, 9)
            if k % 4 == 1:
                t = t[:30] + t[30:70].lower() + t[70:]
        else:
            q0 = rand_dna(rng, rng.choice([24, 35]), AA)
            qs = [q0, rand_dna(rng, 8, AA) + q0[6:20] + rand_dna(rng, 6, AA)]
            t = rand_dna(rng, rng.randint(0, 20), AA) + mutate(rng, q0, rng.choice([0.0, 0.08]), AA) + "X" + rand_dna(rng, rng.randint(3, 25), AA)
            t += qs[1][2:] + rand_dna(rng, 6, AA).lower()
        cases.append(("seeds_%s%03d" % (kind[:3], k), qs, t))
    return cases


def run_seeds(kind, cases, extra=()):
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        for cid, qs, t in cases:
            f.write("%s\t%s\t%s\n" % (cid, ",".join(qs), t))
        path = f.name
    out = subprocess.run([REFDUMP, "--cmd", "seeds", "--model", kind, "--input", path] + list(extra),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    os.unlink(path)
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(recs) == len(cases)
    for r, (cid, qs, t) in zip(recs, cases):
        assert r["id"] == cid
        r["queries"], r["target"] = qs, t
    return recs


def run_hsp(kind, cases, extra=()):
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        for cid, q, t, seeds in cases:
            f.write("%s\t%s\t%s\t%s\n" % (cid, q, t, seeds))
        path = f.name
    out = subprocess.run([REFDUMP, "--cmd", "hsp", "--model", kind, "--input", path] + list(extra),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    os.unlink(path)
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    params, recs = lines[0]["params"], lines[1:]
    assert len(recs) == len(cases)
    for r, (cid, q, t, seeds) in zip(recs, cases):
        assert r["id"] == cid
        r["query"], r["target"] = q, t
        r["seeds"] = [[int(x) for x in sd.split(":")] for sd in seeds.split(",")]
    return [{"params": params}] + recs


def word_hits(q, t, w, kind):
    """Every shared word of the pair in target-scan order (what the seeder's FSM scan reports)."""
    if kind == "protein2dna":
        rev = {c: a for a, cs in CODON.items() for c in cs}
        trans = lambda x: "".join(rev.get(x[i:i + 3], "X") for i in range(0, len(x) - 2, 3))
        step = 3
    else:
        trans = lambda x: x
        step = 1
    words = {}
    for i in range(len(q) - w + 1):
        words.setdefault(q[i:i + w], []).append(i)
    seeds = []
    for j in range(0, len(t) - w * step + 1):
        for i in words.get(trans(t[j:j + w * step]), ()):
            seeds.append("%d:%d" % (i, j))
    return seeds


def indels(rng, s, alphabet, every, unit=1):
    """Deletions and insertions of 1-4 units roughly every `every` positions."""
    out, i = [], 0
    while i < len(s):
        step = rng.randint(every // 2, every * 3 // 2)
        out.append(s[i:i + step])
        i += step
        if i < len(s):
            k = rng.randint(1, 4) * unit
            if rng.random() < 0.5:
                i += k
            else:
                out.append(rand_dna(rng, k, alphabet))
    return "".join(out)


def sdp_cases(model, n, seed):
    """Pairs for SDP (sdp.c:743): the gapped model families with indels / introns / frameshifts between the HSPs, a second
    copy of the gene in some targets (later alignments of the pair), every shared word as a word hit."""
    rng = random.Random(seed)
    cases = []
    base = []
    if model.startswith("affine") and model.endswith(":protein"):
        kind, w = "protein2protein", 5
        for k in range(n):
            q = rand_dna(rng, rng.choice([40, 90, 200, 350]), AA)
            body = indels(rng, mutate(rng, q, rng.choice([0.05, 0.15]), AA), AA, rng.choice([25, 60]))
            t = rand_dna(rng, rng.randint(0, 30), AA) + body + rand_dna(rng, rng.randint(0, 30), AA)
            if k % 4 == 3:
                t += rand_dna(rng, 10, AA) + mutate(rng, q[len(q) // 3:], 0.1, AA)
            base.append(("prot%03d" % k, q, t))
    elif model.startswith("affine"):
        kind, w = "dna2dna", 10
        for k in range(n):
            q = rand_dna(rng, rng.choice([60, 150, 300, 500]), "ACGT" if k % 5 else "ACGTN")
            body = indels(rng, mutate(rng, q, rng.choice([0.02, 0.06, 0.1]), "ACGT"), "ACGT", rng.choice([30, 70, 150]))
            t = rand_dna(rng, rng.randint(0, 60)) + body + rand_dna(rng, rng.randint(0, 60))
            if k % 4 == 3:
                t += rand_dna(rng, 25) + indels(rng, mutate(rng, q[len(q) // 4:], 0.05, "ACGT"), "ACGT", 80)
            base.append(("dna%03d" % k, q, t))
    elif model == "est2genome":
        kind, w = "dna2dna", 10
        for cid, q, t in est_pairs(rng, n):
            if rng.random() < 0.5:
                t = indels(rng, t, "ACGT", 120)
            base.append((cid, q, t))
    elif model == "protein2dna":
        kind, w = "protein2dna", 4
        base = p2d_pairs(rng, n)
    else:
        kind, w = "protein2dna", 4
        base = p2g_pairs(rng, n)
    for k, (cid, q, t) in enumerate(base):
        if k % 4 == 3 and kind != "protein2protein" and not model.startswith("affine"):
            t = t + rand_dna(rng, 30) + t[len(t) // 4:]                   # second copy: later alignments of the pair
        seeds = word_hits(q, t, w, kind)
        if seeds:
            cases.append(("sdp_" + cid, q, t, ",".join(seeds)))
    return cases


def run_sdp(model, cases, extra=()):
    with tempfile.NamedTemporaryFile("w", suffix=".tsv", delete=False) as f:
        for cid, q, t, seeds in cases:
            f.write("%s\t%s\t%s\t%s\n" % (cid, q, t, seeds))
        path = f.name
    out = subprocess.run([REFDUMP, "--cmd", "sdp", "--model", model, "--input", path] + list(extra),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    os.unlink(path)
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    params, recs = lines[0]["params"], lines[1:]
    assert len(recs) == len(cases)
    for r, (cid, q, t, seeds) in zip(recs, cases):
        assert r["id"] == cid
        r["query"], r["target"] = q, t
    return [{"params": params}] + recs


def main():
    rng = random.Random(20260928)
    sets = []
    dna = dna_pairs(rng, 30)
    prot = protein_pairs(rng, 18) + [KAT_AFFINE]
    est = est_pairs(rng, 28) + [KAT_E2G]
    p2d = p2d_pairs(rng, 24) + [KAT_P2D]
    p2g = p2g_pairs(random.Random(99), 26) + [KAT_P2G]
    for scope in ("local", "global", "bestfit", "overlap"):
        # empty sequences are rejected by the reference's Sequence_create for non-local global DP
        d = [c for c in dna if len(c[1]) > 0 and len(c[2]) > 0]
        sets.append(("affine_%s_dna" % scope, "affine:%s" % scope, d, 32, ()))
        sets.append(("affine_%s_protein" % scope, "affine:%s:protein" % scope,
                     [c for c in prot if len(c[2]) > 0], 32, ()))
    sets.append(("est2genome", "est2genome", est, 32, ("--withsplice", "yes")))
    sets.append(("protein2dna", "protein2dna", [c for c in p2d if len(c[2]) > 0], 32, ()))
    # the same inputs with a tiny traceback budget force the reduced-space route (region ->
    # checkpoints -> recursion -> continuation sub-alignments) on small inputs: -D 0 => every
    # region larger than 6x the max advance goes through checkpoints.
    big = [c for c in dna if len(c[1]) >= 13 and len(c[2]) >= 13]
    sets.append(("affine_local_dna_D0", "affine:local", big, 0, ()))
    sets.append(("affine_global_dna_D0", "affine:global", big, 0, ()))
    sets.append(("est2genome_D0", "est2genome", est, 0, ()))
    sets.append(("protein2dna_D0", "protein2dna", [c for c in p2d if len(c[2]) > 30], 0, ()))
    sets.append(("protein2genome", "protein2genome", p2g, 32, ()))
    sets.append(("protein2genome_D0", "protein2genome", p2g, 0, ()))
    # a larger est2genome case that takes the reduced-space route at the DEFAULT -D 32
    rng2 = random.Random(7)
    q = rand_dna(rng2, 700)
    t = rand_dna(rng2, 300) + q[:250] + "GT" + rand_dna(rng2, 700) + "AG" + q[250:520] + "GT" + \
        rand_dna(rng2, 1500) + "AG" + q[520:] + rand_dna(rng2, 400)
    sets.append(("est2genome_big", "est2genome", [("estbig0", q, t)], 32, ()))
    # the remaining accelerated model types: ungapped (dna, protein, protein vs dna) and the bestfit forms
    sets.append(("ungapped_dna", "ungapped", [c for c in dna if len(c[1]) > 0 and len(c[2]) > 0], 32, ()))
    sets.append(("ungapped_protein", "ungapped:protein", [c for c in prot if len(c[1]) > 0 and len(c[2]) > 0], 32, ()))
    sets.append(("ungapped_dna_D0", "ungapped", big, 0, ()))
    sets.append(("protein2dna_bestfit", "protein2dna:bestfit", [c for c in p2d if len(c[2]) > 30], 32, ()))
    sets.append(("protein2dna_bestfit_D0", "protein2dna:bestfit", [c for c in p2d if len(c[2]) > 30], 0, ()))
    sets.append(("protein2genome_bestfit", "protein2genome:bestfit", p2g, 32, ()))
    sets.append(("protein2genome_bestfit_D0", "protein2genome:bestfit", p2g, 0, ()))
    # --forcegtag: splice sites off GT..AG (CT..AC on the reverse strand) become impossible (splice.c:334-336)
    fg = ("--forcegtag", "yes")
    sets.append(("est2genome_forcegtag", "est2genome", est, 32, fg + ("--withsplice", "yes")))
    sets.append(("est2genome_forcegtag_D0", "est2genome", est, 0, fg))
    sets.append(("protein2genome_forcegtag", "protein2genome", p2g, 32, fg))
    # BSDP's derived models (C4_DerivedModel_create, heuristic.c:242-330) on rectangles of the size BSDP gives
    # them: start terminal (model start scope -> CORNER), end terminal, join (CORNER -> CORNER)
    small = random.Random(515)
    sd = [("drv%02d" % k, rand_dna(small, small.randint(2, 48)), rand_dna(small, small.randint(2, 48))) for k in range(10)]
    sd += [("drvsim%02d" % k, q, mutate(small, q, 0.15, "ACGT")) for k, q in
           enumerate(rand_dna(small, small.randint(8, 48)) for _ in range(10))]
    sd = [c for c in sd if len(c[2]) >= 2]
    se = [c for c in est if 2 <= len(c[1]) <= 60 and len(c[2]) <= 400][:8] + sd[:8]
    sp = [c for c in p2d if 2 <= len(c[1]) <= 40 and 6 <= len(c[2]) <= 200][:12]
    sg = [c for c in p2g if 2 <= len(c[1]) <= 40 and 6 <= len(c[2]) <= 400][:10]
    for tag, model, cases, match_states in (("affine_local", "affine:local", sd, (2,)), ("est2genome", "est2genome", se, (2, 5)),
                                            ("protein2dna", "protein2dna", sp, (2,)), ("protein2genome", "protein2genome", sg, (2,))):
        for ms in match_states:
            sfx = "" if len(match_states) == 1 else ("_fwd" if ms == 2 else "_rev")
            sets.append(("derived_%s%s_start" % (tag, sfx), model, cases, 32, ("--derived", "0,%d,0,4" % ms)))
            sets.append(("derived_%s%s_end" % (tag, sfx), model, cases, 32, ("--derived", "%d,1,4,0" % ms)))
            sets.append(("derived_%s%s_join" % (tag, sfx), model, cases, 32, ("--derived", "%d,%d,4,4" % (ms, ms))))
    # sub-optimal alignments (SubOpt blocking, src/c4/subopt.c) through the GAM loop
    so = ("--suboptmax", "6", "--suboptthreshold", "30")
    rs = random.Random(4242)
    sub_dna, sub_est = repeat_pairs(rs, 16, "dna"), repeat_pairs(rs, 10, "est")
    sub_p2d, sub_p2g = repeat_pairs(rs, 8, "p2d"), repeat_pairs(rs, 8, "p2g")
    sub_p2d = [c for c in sub_p2d if len(c[2]) > 30]
    sets.append(("affine_local_dna_subopt", "affine:local", sub_dna, 32, so))
    sets.append(("affine_local_dna_subopt_D0", "affine:local", sub_dna, 0, so))
    sets.append(("affine_global_dna_subopt", "affine:global", sub_dna, 32, ("--suboptmax", "3", "--suboptthreshold", "-100000")))
    sets.append(("est2genome_subopt", "est2genome", sub_est, 32, so))
    sets.append(("est2genome_subopt_D0", "est2genome", sub_est, 0, so))
    sets.append(("protein2dna_subopt", "protein2dna", sub_p2d, 32, so))
    sets.append(("protein2dna_subopt_D0", "protein2dna", sub_p2d, 0, so))
    sets.append(("protein2genome_subopt", "protein2genome", sub_p2g, 32, so))
    sets.append(("protein2genome_subopt_D0", "protein2genome", sub_p2g, 0, so))
    # non-default scoring parameters, parsed by the reference's own ArgumentSets (affine.c:24-49, intron.c:24-32,
    # frameshift.c, match.c): penalties, intron length window, pam250 / identity matrices (submat.c:296-307).
    # The flag lists are tests/golden_util.py:PARAM_VARIANTS; the matrices travel in scoring_data_alt.json.
    alt = tuple(ALT_FLAGS)
    sets.append(("affine_local_dna_altparams", "affine:local", d, 32, alt))
    sets.append(("affine_local_dna_altparams_D0", "affine:local", big, 0, alt))
    sets.append(("affine_global_protein_altparams", "affine:global:protein", [c for c in prot if len(c[2]) > 0], 32, alt))
    sets.append(("est2genome_altparams", "est2genome", est, 32, alt + ("--withsplice", "yes")))
    sets.append(("est2genome_altparams_D0", "est2genome", est, 0, alt))
    sets.append(("protein2dna_altparams", "protein2dna", [c for c in p2d if len(c[2]) > 0], 32, alt))
    sets.append(("protein2dna_altparams_D0", "protein2dna", [c for c in p2d if len(c[2]) > 30], 0, alt))
    sets.append(("protein2genome_altparams", "protein2genome", p2g, 32, alt))
    sets.append(("protein2genome_altparams_D0", "protein2genome", p2g, 0, alt))
    sets.append(("est2genome_altparams_subopt", "est2genome", sub_est, 32, alt + so))
    # extreme magnitudes / degenerate windows on the local-scope models (12 cases each, both routes)
    base_cases = {"affine_local_dna": ("affine:local", d[:14]), "est2genome": ("est2genome", est[:12] + [KAT_E2G]),
                  "protein2dna": ("protein2dna", [c for c in p2d if len(c[2]) > 30][:12]),
                  "protein2genome": ("protein2genome", p2g[:12] + [KAT_P2G])}
    for tag, bases in (("hugegap", ("affine_local_dna", "est2genome", "protein2dna", "protein2genome")),
                       ("hugeintron", ("est2genome", "protein2dna", "protein2genome")),
                       ("tightintron", ("est2genome", "protein2genome")),
                       ("invertedintron", ("est2genome", "protein2genome")),
                       ("posgap", ("affine_local_dna", "est2genome", "protein2dna", "protein2genome"))):
        for b in bases:
            model, cases = base_cases[b]
            sets.append(("%s_%s" % (b, tag), model, cases, 32, tuple(PARAM_VARIANTS[tag])))
            sets.append(("%s_%s_D0" % (b, tag), model, [c for c in cases if len(c[1]) >= 13 and len(c[2]) >= 13], 0,
                         tuple(PARAM_VARIANTS[tag])))
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    # SDP (sdp.c:743, scheduler.c:859-1065): alignments of GAM_Result_SDP_create's loop from the reference's own HSPs
    sub = ("--suboptmax", "4", "--suboptthreshold", "40")
    for name, model, n, extra in (
            ("sdp_affine_local", "affine:local", 24, ("--dnawordlen", "10") + sub),
            ("sdp_affine_local_protein", "affine:local:protein", 16, ("--proteinwordlen", "5") + sub),
            ("sdp_est2genome", "est2genome", 20, ("--dnawordlen", "10") + sub),
            ("sdp_est2genome_drop", "est2genome", 14, ("--dnawordlen", "10", "--extensionthreshold", "12", "--dnahspthreshold", "30") + sub),
            ("sdp_est2genome_altparams", "est2genome", 14, tuple(ALT_FLAGS) + ("--dnawordlen", "10", "--singlepass", "no",
                                                                        "--dnahspthreshold", "8") + sub),
            ("sdp_protein2dna", "protein2dna", 16, ("--proteinwordlen", "4") + sub),
            ("sdp_protein2genome", "protein2genome", 20, ("--proteinwordlen", "4") + sub),
            ("sdp_protein2genome_altparams", "protein2genome", 14, tuple(ALT_FLAGS) + ("--proteinwordlen", "4", "--extensionthreshold", "25") + sub)):
        if only and name not in only:
            continue
        recs = run_sdp(model, sdp_cases(model, n, 77 + len(name)), extra)
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs) - 1, "pairs,", sum(len(r["hsps"]) for r in recs[1:]), "HSPs,",
              sum(len(r["alignments"]) for r in recs[1:]), "alignments", recs[0]["params"])
    # HSP seeding (hspset.c:933): per-seed HSPs and whole-set HSP lists, default and lowered thresholds / dropoffs
    for name, kind, extra in (("hsp_dna2dna", "dna2dna", ()), ("hsp_dna2dna_low", "dna2dna", ("--dnahspthreshold", "20", "--dnahspdropoff", "8")),
                              ("hsp_protein2protein", "protein2protein", ()),
                              ("hsp_protein2dna", "protein2dna", ("--proteinhspthreshold", "12")),
                              ("hsp_protein2dna_drop", "protein2dna", ("--proteinhspdropoff", "5", "--proteinhspthreshold", "5"))):
        if only and name not in only:
            continue
        recs = run_hsp(kind, hsp_cases(kind, 14, 31 + len(name)), extra)
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs) - 1, "pairs,", sum(len(r["seeds"]) for r in recs[1:]), "seeds,", sum(len(r["set"]) for r in recs[1:]), "HSPs")
    # the seeder's automaton walk (seeder.c:649-720,852-915): word tables off the reference's automaton + the calls its walk made
    for name, kind, n, extra in (("seeds_dna2dna", "dna2dna", 8, ()), ("seeds_dna2dna_w9", "dna2dna", 5, ("--dnawordlen", "9")),
                                 ("seeds_dna2dna_compact", "dna2dna", 4, ("--forcefsm", "compact")),
                                 ("seeds_dna2dna_hood", "dna2dna", 3, ("--dnawordlen", "8", "--dnawordlimit", "5")),
                                 ("seeds_protein2protein", "protein2protein", 3, ("--proteinwordlimit", "2")),
                                 ("seeds_protein2protein_compact", "protein2protein", 2, ("--proteinwordlimit", "1", "--forcefsm", "compact"))):
        if only and name not in only:
            continue
        recs = run_seeds(kind, seed_cases(kind, n, 77 + len(name)), extra)
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs), "seeders,", sum(len(r["words"]) for r in recs), "words,", sum(sum(len(w[2]) for w in r["words"]) for r in recs),
              "neighbour links,", sum(len(r["expected"]) for r in recs), "seeds", [r["automaton"] for r in recs][:1])
    if not only or "scoring_data_alt" in only:
        out = subprocess.run([REFDUMP, "--cmd", "data"] + ALT_FLAGS, stdout=subprocess.PIPE, check=True).stdout.decode()
        dd = json.loads(out)
        # dump_data prints whatever matrices are loaded under fixed key names
        keep = {"dnasubmat:identity": dd["nucleic"], "proteinsubmat:pam250": dd["blosum62"]}
        with open(os.path.join(OUT, "scoring_data_alt.json"), "w") as f:
            json.dump(keep, f, separators=(",", ":"))
    # span models (BSDP): src DP reporting END cells, dst DP starting from them (refdump --cmd span)
    sr = random.Random(909)
    span_cases = []
    for k in range(8):
        q = rand_dna(sr, sr.randint(12, 30))
        c = len(q) // 2
        rev = k % 2
        t = rand_dna(sr, sr.randint(0, 6)) + q[:c] + ("CT" if rev else "GT") + rand_dna(sr, sr.randint(30, 70)) + \
            ("AC" if rev else "AG") + mutate(sr, q[c:], 0.05, "ACGT") + rand_dna(sr, sr.randint(0, 6))
        span_cases.append(("span%02d" % k, q, t))
    for name, ms, ss in (("span_est2genome_fwd", 2, 8), ("span_est2genome_rev", 5, 9)):
        if only and name not in only:
            continue
        recs = run_span(span_cases, ms, ss)
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs), "dst scores", [r["dst_score"] for r in recs])
    # protein2genome's three spans (intron in phase 0, 1, 2: span states 10, 11, 12)
    pr = random.Random(910)
    p2g_span_cases = []
    for k in range(9):
        q = "".join(pr.choice(AA) for _ in range(pr.randint(5, 11)))
        coding = "".join(pr.choice(CODON[a]) for a in mutate(pr, q, 0.05, AA))
        c = 3 * pr.randint(1, len(q) - 2) + (k % 3)
        t = rand_dna(pr, pr.choice([0, 0, 3])) + coding[:c] + "GT" + rand_dna(pr, pr.randint(30, 60)) + "AG" + \
            coding[c:] + rand_dna(pr, pr.choice([0, 0, 2]))
        p2g_span_cases.append(("pspan%02d" % k, q, t))
    for name, ms, ss in (("span_protein2genome_phase0", 2, 10), ("span_protein2genome_phase1", 2, 11),
                         ("span_protein2genome_phase2", 2, 12)):
        if only and name not in only:
            continue
        recs = run_span(p2g_span_cases, ms, ss, "protein2genome")
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs), "dst scores", [r["dst_score"] for r in recs])
    # --annotation (match.c:276-281): DNA queries with a CDS annotation -- no 1:1 DNA match inside it.  CDS in the middle, at
    # either end, over the whole query, past its end, one residue long
    def annotated(cases, seed):
        ar = random.Random(seed)
        out = []
        for k, (cid, q, t) in enumerate(cases):
            n = len(q)
            kind = k % 6
            cds = ((n // 3, max(1, n // 3)), (0, max(1, n // 4)), (max(0, n - max(1, n // 4)), max(1, n // 4)), (0, n + 5),
                   (ar.randint(0, max(0, n - 1)), 1), (n // 2, n))[kind]
            out.append((cid, q, t, cds))
        return out
    sets.append(("est2genome_annot", "est2genome", annotated([c for c in est if len(c[1]) >= 30][:14], 41), 32, ()))
    sets.append(("est2genome_annot_D0", "est2genome", annotated([c for c in est if len(c[1]) >= 30 and len(c[2]) >= 13][:12], 42), 0, ()))
    sets.append(("affine_local_dna_annot", "affine:local", annotated([c for c in d if len(c[1]) >= 13 and len(c[2]) >= 13][:14], 43), 32, ()))
    sets.append(("affine_local_dna_annot_D0", "affine:local", annotated([c for c in d if len(c[1]) >= 13 and len(c[2]) >= 13][:12], 44), 0, ()))
    for name, model, cases, dpm, extra in sets:
        if only and name not in only:
            continue
        recs = run(model, cases, dpm, extra)
        if name.endswith("_subopt_D0"):      # the point sets are those of the -D 32 twin: keep the files small
            for r in recs:
                for a in r.get("subopt", []):
                    a.pop("points", None)
        with open(os.path.join(OUT, name + ".jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r, separators=(",", ":")) + "\n")
        print(name, len(recs), "scores", [r["score"] for r in recs[:6]])


if __name__ == "__main__":
    main()
