"""Kernel variants the default inputs never select, each held to the same bar as the default ones:

* the UNPACKED FIND_REGION kernels (two region-start slots per state; taken when bits(Q) + bits(T) > 31, e.g.
  300 aa x 10 Mb) — forced with C4GPU_PACK=0 over the reference's vector sets and seeded pairs, one-wave,
  4-wave and 8-wave forms;
* the GENERAL score / region kernels (every Layout validity mask kept) that local models fall back to when the
  scoring parameters are too large for the local-scope shortcuts (Engine::local_exact) — forced with
  C4GPU_LOCAL_EXACT=0, and reached for real with penalties of hundreds of millions;
* BASELINE config 5's shape at its real size: proteins against ONE 10 Mb contig (unpacked kernels by
  necessity), checked through the window property against the oracle.
"""
import random
import pytest

import exonerate_amd as ex
from exonerate_amd import _abi, workloads
import oracle_lib
from golden_util import SETS, SUBOPT_SETS, PARAM_VARIANTS, apply_flags, load_set, expected, set_params

TABLE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
CODON = {}
for _i, _a in enumerate("TCAG"):
    for _j, _b in enumerate("TCAG"):
        for _k, _c in enumerate("TCAG"):
            CODON.setdefault(TABLE[_i * 16 + _j * 4 + _k], []).append(_a + _b + _c)

pytestmark = pytest.mark.gpu

# sets whose alignments go through the region pass on small inputs (-D 0), plus the larger default-route one
_EXTREME = ("hugegap", "hugeintron", "tightintron", "invertedintron", "posgap")
REGION_SETS = sorted(n for n in SETS if n.endswith("_D0") and not any(t in n for t in _EXTREME)) + ["est2genome_big"]


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def _model(name):
    mt, qa, ta = SETS[name] if name in SETS else SUBOPT_SETS[name]
    return ex.Model(mt, qa, ta, params=set_params(_abi.load(), name))


def _check_set(eng, name):
    model = _model(name)
    recs = load_set(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    assert eng.find_score(model, pairs) == [r["score"] for r in recs]
    alns = eng.find_path(model, pairs, dpmemory=recs[0]["dpmemory"])
    for rec, aln in zip(recs, alns):
        if "path_score" not in rec:
            assert aln is None, rec["id"]
        else:
            assert aln is not None and aln.as_dict(rec["id"]) == expected(rec), rec["id"]


@pytest.mark.parametrize("name", REGION_SETS)
def test_unpacked_region_kernels_match_reference_vectors(eng, monkeypatch, name):
    monkeypatch.setenv("C4GPU_PACK", "0")
    _check_set(eng, name)


@pytest.mark.parametrize("name", sorted(n for n in SETS if ("local" in n or "est2genome" in n or n.startswith("protein2")
                                                         or n.startswith("ungapped"))
                                        and not any(t in n for t in _EXTREME if not t.startswith("huge"))))
def test_general_kernels_match_reference_vectors(eng, monkeypatch, name):
    """Local models on the kernels that keep every validity mask (what very large penalties select)."""
    monkeypatch.setenv("C4GPU_LOCAL_EXACT", "0")
    _check_set(eng, name)


@pytest.mark.parametrize("name", ["est2genome_subopt", "est2genome_subopt_D0", "affine_local_dna_subopt_D0",
                                  "protein2genome_subopt_D0", "est2genome_altparams_subopt"])
@pytest.mark.parametrize("switch", ["C4GPU_PACK", "C4GPU_LOCAL_EXACT"])
def test_blocking_kernels_of_both_variants(eng, monkeypatch, name, switch):
    """The sub-optimal loop (blocked MATCH cells) through the unpacked / general kernels."""
    monkeypatch.setenv(switch, "0")
    model = _model(name)
    recs = load_set(name)
    found = eng.find_all_paths(model, [(r["query"], r["target"]) for r in recs], dpmemory=recs[0]["dpmemory"],
                               threshold=recs[0]["threshold"], max_paths=6)
    for rec, alns in zip(recs, found):
        assert [(a.score, list(a.region), [list(o) for o in a.ops], a.vulgar(rec["id"])) for a in alns] == \
               [(e["path_score"], e["region"], e["ops"], e["vulgar"]) for e in rec["subopt"]], rec["id"]


def _rand(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _mutate(rng, s, rate, alpha="ACGT"):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice(alpha))
        elif r < 2 * rate / 3:
            out.append(ch + rng.choice(alpha))
        elif r >= rate:
            out.append(ch)
    return "".join(out)


def _seeded_pairs(rng, model_type, qlen, tlen, n):
    pairs = []
    for k in range(n):
        q = _rand(rng, qlen + 3 * k)
        if model_type == "est2genome":
            c = qlen // 3
            t = (_rand(rng, 120) + _mutate(rng, q[:c], 0.03) + "GT" + _rand(rng, tlen // 3) + "AG" +
                 _mutate(rng, q[c:], 0.03) + _rand(rng, 200))
        else:
            t = _rand(rng, 50) + _mutate(rng, q, 0.08) + _rand(rng, max(0, tlen - qlen))
        pairs.append((q, t))
    return pairs


@pytest.mark.parametrize("switch", ["C4GPU_PACK", "C4GPU_LOCAL_EXACT"])
@pytest.mark.parametrize("model_type,qlen,tlen,n", [
    ("est2genome", 300, 3000, 5),       # one strip: one-wave kernels
    ("est2genome", 1300, 5000, 2),      # > 1 024 rows: cooperating waves + HBM carry rows between super-strips
    ("affine:local", 900, 1100, 6),     # several strips, many jobs: 4 cooperating waves per job
    ("affine:local", 2100, 2300, 2),
])
def test_seeded_pairs_on_unpacked_and_general_kernels(eng, monkeypatch, switch, model_type, qlen, tlen, n):
    monkeypatch.setenv(switch, "0")
    rng = random.Random(qlen * 7 + len(switch))
    model = ex.Model(model_type)
    pairs = _seeded_pairs(rng, model_type, qlen, tlen, n)
    scores = eng.find_score(model, pairs)
    alns = eng.find_path(model, pairs, dpmemory=1)
    for (q, t), s_, a in zip(pairs, scores, alns):
        assert s_ == oracle_lib.find_score(model.c, model.params, q.encode(), t.encode())
        assert a.as_dict() == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=1)


@pytest.mark.parametrize("kind", ["hugegap", "hugeintron", "tightintron", "invertedintron", "posgap"])
@pytest.mark.parametrize("model_type", ["affine:local", "est2genome", "protein2dna", "protein2genome"])
def test_extreme_parameters_match_oracle(eng, model_type, kind):
    """User-settable penalties at magnitudes where "unset" (-987654321) and real scores are no longer far apart,
    degenerate intron windows, rewards instead of penalties — on inputs of several strips (the reference's own
    small vectors with these flags are part of SETS; the oracle is pinned on them).  The magnitude guard
    (Engine::local_exact) must route the huge ones to kernels that stay exact."""
    import zlib
    rng = random.Random(zlib.crc32(("%s %s" % (model_type, kind)).encode()))
    params = apply_flags(ex.default_params(), PARAM_VARIANTS[kind])
    model = ex.Model(model_type, params=params)
    pairs = []
    for k in range(4):
        if model_type.startswith("protein"):
            aa = "ARNDCQEGHILKMFPSTWYV"
            q = _rand(rng, rng.choice([20, 90, 150, 320]), aa)
            coding = "".join(rng.choice(CODON[x]) for x in _mutate(rng, q, 0.06, aa))
            if "genome" in model_type and len(coding) > 60:
                c = rng.randint(10, len(coding) - 10)
                coding = coding[:c] + "GT" + _rand(rng, rng.choice([40, 75, 300])) + "AG" + coding[c:]
            if k % 2:
                c = rng.randint(5, len(coding) - 5)
                coding = coding[:c] + "A" + coding[c:]
            t = _rand(rng, rng.randint(0, 60)) + coding + _rand(rng, rng.randint(0, 60))
        else:
            q = _rand(rng, rng.choice([40, 130, 300, 700]))
            if model_type == "est2genome":
                c = len(q) // 2
                t = _rand(rng, 30) + _mutate(rng, q[:c], 0.04) + "GT" + _rand(rng, rng.choice([40, 74, 400])) + "AG" + \
                    _mutate(rng, q[c:], 0.04) + _rand(rng, 40)
            else:
                t = _rand(rng, rng.randint(0, 30)) + _mutate(rng, q, 0.12) + _rand(rng, rng.randint(0, 30))
        pairs.append((q, t))
    scores = eng.find_score(model, pairs)
    for dpm in (32, 0):
        alns = eng.find_path(model, pairs, dpmemory=dpm)
        for (q, t), s_, a in zip(pairs, scores, alns):
            assert s_ == oracle_lib.find_score(model.c, model.params, q.encode(), t.encode()), (kind, dpm)
            exp = oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dpm)
            assert (a.as_dict() if a else None) == exp, (kind, dpm)


def test_c5_protein2genome_against_a_10mb_contig(eng):
    """BASELINE config 5's exhaustive shape at full size: proteins vs ONE 10 Mb chromosome.  9 + 24 bits of
    coordinates do not fit the packed region-start slot: these launches run the unpacked kernels (8 cooperating
    waves per job for the 16-job launch, 4 for the 272-job one).  Parity through the window property: a local
    alignment whose path lies inside a window of the contig is the window's alignment shifted by its offset."""
    proteins, contig, places = workloads.protein_vs_contig(16, 300, 10000000, seed=20260936, introns=True)
    model = ex.Model("protein2genome")
    alns = eng.find_path(model, [(p, contig) for p in proteins], dpmemory=32)
    margin = 1000
    for p, (g0, g1), a in zip(proteins, places, alns):
        w0, w1 = max(0, g0 - margin), min(len(contig), g1 + margin)
        exp = oracle_lib.find_path(model.c, model.params, p, contig[w0:w1], dpmemory=32)
        assert a is not None and exp is not None
        assert a.score == exp["score"]
        assert [list(o) for o in a.ops] == exp["ops"]
        r = exp["region"]
        assert list(a.region) == [r[0], r[1] + w0, r[2], r[3]]
    # the same 16 proteins 17 times over: a launch with more jobs than CUs takes the 4-wave kernels
    many = eng.find_path(model, [(p, contig) for p in proteins] * 17, dpmemory=32)
    for k, a in enumerate(many):
        assert a.as_dict() == alns[k % 16].as_dict(), k


@pytest.mark.parametrize("model_type,dpm", [("est2genome", 32), ("est2genome", 0), ("affine:local", 1), ("protein2genome", 32)])
def test_find_path_over_regions_of_resident_pairs(eng, model_type, dpm):
    """c4gpu_batch_run_regions: Optimal_find_path with a region argument (what --refine region asks for: the heuristic
    alignment's box grown by 32) for every pair of a resident batch in one go, against the oracle's region form."""
    import zlib
    rng = random.Random(zlib.crc32(model_type.encode()) + dpm)
    model = ex.Model(model_type)
    pairs = []
    for k in range(6):
        if model_type.startswith("protein"):
            aa = "ARNDCQEGHILKMFPSTWYV"
            q = _rand(rng, 150 + 20 * k, aa)
            coding = "".join(rng.choice(CODON[x]) for x in _mutate(rng, q, 0.05, aa))
            c = len(coding) // 2
            t = _rand(rng, 900) + coding[:c] + "GT" + _rand(rng, 300) + "AG" + coding[c:] + _rand(rng, 700)
        else:
            pairs_k = _seeded_pairs(rng, model_type, 500 + 40 * k, 6000, 1)[0]
            q, t = pairs_k
        pairs.append((q, t))
    full = eng.find_path(model, pairs, dpmemory=dpm)
    regions = []
    for (q, t), a in zip(pairs, full):
        qs, ts, ql, tl = a.region
        r0, r1 = max(0, qs - 32), max(0, ts - 32)
        regions.append((r0, r1, min(len(q), qs + ql + 32) - r0, min(len(t), ts + tl + 32) - r1))
    regions[1] = (0, 0, len(pairs[1][0]), len(pairs[1][1]) // 3)           # a region that cuts the gene
    batch = ex.ResidentBatch(eng, model, pairs)
    batch.run_regions(regions, dpmemory=dpm, threshold=0)
    got = [batch.alignment(i) for i in range(len(pairs))]
    batch.close()
    for (q, t), r, a in zip(pairs, regions, got):
        exp = oracle_lib.find_path_region(model.c, model.params, q.encode(), t.encode(), r, dpmemory=dpm, threshold=0)
        assert (a.as_dict() if a else None) == exp, r
    assert got[0].as_dict() == full[0].as_dict()                           # the box holds the whole alignment


# (the 1 300-row est2genome case -- 10 s of oracle per run -- takes the smallest and the largest interval only: the GPU suite's time)
@pytest.mark.parametrize("model_type,qlen,tlen,dpm,kshift", [
    (m, q, t, d, k) for (m, q, t, d) in [("est2genome", 600, 6000, 32), ("est2genome", 1300, 5000, 1), ("affine:local", 700, 2500, 1),
                                         ("affine:local", 2100, 2300, 32), ("protein2dna", 300, 3000, 32),
                                         ("protein2genome", 330, 5000, 32)]
    for k in ("3", "5", "7") if not (q == 1300 and k == "5")])
def test_windowed_region_pass_matches_oracle(eng, monkeypatch, capfd, model_type, qlen, tlen, dpm, kshift):
    """FIND_REGION in two passes (score pass that dumps the DP state every 2^kshift columns, region-start payload
    passes over one dump interval at a time, walking left from the end cell): same scores, end cells and region starts
    as the one-pass kernel — checked through the alignments, against the oracle, with tiny dump intervals so that a
    path crosses many of them (and introns jump over dumped columns)."""
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", kshift)
    monkeypatch.setenv("C4GPU_TRACE", "1")
    import zlib
    rng = random.Random(zlib.crc32(("%s %d %s" % (model_type, qlen, kshift)).encode()))
    model = ex.Model(model_type)
    pairs = []
    for k in range(4):
        if model_type.startswith("protein"):
            aa = "ARNDCQEGHILKMFPSTWYV"
            q = _rand(rng, qlen + 7 * k, aa)
            coding = "".join(rng.choice(CODON[x]) for x in _mutate(rng, q, 0.05, aa))
            if "genome" in model_type:
                c1, c2 = len(coding) // 3, 2 * len(coding) // 3 + 1
                coding = coding[:c1] + "GT" + _rand(rng, 150 + 40 * k) + "AG" + coding[c1:c2] + "GT" + _rand(rng, 700) + "AG" + coding[c2:]
            t = _rand(rng, tlen // 3) + coding + _rand(rng, tlen // 2)
        else:
            q, t = _seeded_pairs(rng, model_type, qlen + 11 * k, tlen, 1)[0]
        pairs.append((q, t))
    pairs.append((pairs[0][0], _rand(rng, tlen)))                       # unrelated: a short alignment anywhere
    alns = eng.find_path(model, pairs, dpmemory=dpm, threshold=20)
    err = capfd.readouterr().err
    assert "windowed region pass" in err, "these inputs no longer take the two-pass route:\n" + err[-800:]
    # the score pass of the scheme is the packed 16-bit kernel (two jobs per lane) for the family that has one
    assert ("kernel kpk16" in err) == (model_type == "est2genome"), err[-800:]
    for (q, t), a in zip(pairs, alns):
        exp = oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dpm, threshold=20)
        assert (a.as_dict() if a else None) == exp


def test_window_kernel_on_two_and_four_waves_agree(eng, monkeypatch, capfd):
    """The region windows run on two cooperating waves per job by default (est2genome), C4GPU_WIN_NW=4 keeps four: same
    alignments on pairs whose windows need several super-strips in either form and whose later windows are longer than
    the first ones (the strip carry rows are laid out for the longest window of any hop)."""
    rng = random.Random(3131)
    model = ex.Model("est2genome")
    pairs = []
    for ql, tl in [(1000, 100000), (1000, 70000), (900, 52000), (640, 41000), (1000, 33000)]:
        pairs.append(_seeded_pairs(rng, "est2genome", ql, tl, 1)[0])
    q = _rand(rng, 1000)                                      # an intron of 30 kb: the second window is as high as the first
    pairs.append((q, _rand(rng, 5000) + _mutate(rng, q[:150], 0.03) + "GT" + _rand(rng, 30000) + "AG" + _mutate(rng, q[150:], 0.03) + _rand(rng, 9000)))
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_WIN16", "0")                # the 32-bit windows (the packed ones: test_packed_16_bit_region_windows_..)
    res = {}
    for nw in ("2", "4"):
        monkeypatch.setenv("C4GPU_WIN_NW", nw)
        res[nw] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=50)]
        err = capfd.readouterr().err
        assert ("kernel kmw2_est2genome_region_local_pack_seed2" in err) == (nw == "2"), err[-1500:]
        assert ("kernel kmw_est2genome_region_local_pack_seed2" in err) == (nw == "4"), err[-1500:]
    assert res["2"] == res["4"] and all(r is not None for r in res["2"])
    q, t = pairs[5]
    assert res["2"][5] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=32, threshold=50)


def test_windowed_and_one_pass_region_agree_at_full_size(eng, monkeypatch):
    """1 kb x 100 kb est2genome pairs of the north-star batch: the two-pass route (default) and the one-pass kernel
    (C4GPU_WINDOWED=0) give identical alignments; pair 0 also against the oracle."""
    model = ex.Model("est2genome")
    pairs = workloads.est2genome_pairs(24, 1000, 100000, first=100)
    two = eng.find_path(model, pairs, dpmemory=32)
    monkeypatch.setenv("C4GPU_WINDOWED", "0")
    one = eng.find_path(model, pairs, dpmemory=32)
    assert [a.as_dict() for a in two] == [a.as_dict() for a in one]
    q, t = pairs[0]
    assert two[0].as_dict() == oracle_lib.find_path(model.c, model.params, q, t, dpmemory=32)


def test_window_hop_budget_covers_paths_across_the_whole_window(eng, monkeypatch, capfd):
    """Paths that run back over the whole 100 kb window (a 90 kb intron; reverse-complemented cDNAs, whose chance alignments
    chain hits over 93 kb) need 13-14 windows at 8 192 columns per dump: the hop budget follows the longest way back, no
    pair is left to the one-pass kernel, and the alignments are the ones a budget of ONE hop gives (every pair finished by
    the one-pass kernel then)."""
    rng = random.Random(4242)
    model = ex.Model("est2genome")
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    pairs = []
    for k in range(3):
        q = _rand(rng, 1000 - 7 * k)
        pairs.append((q.encode(), (_rand(rng, 1500 + 300 * k) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 90000) + "AG" +
                                   _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)).encode()))
    for q, t in workloads.est2genome_pairs(5, 1000, 100000, first=200):
        pairs.append((q.translate(comp)[::-1], t))
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "13")         # the default spacing, and the two-pass route whatever earlier tests measured
    full = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=100)]
    err = capfd.readouterr().err
    assert "windowed region pass" in err and " 0 of %d paths left to the one-pass kernel" % len(pairs) in err, err[-1500:]
    assert all(a is not None for a in full[:3]) and all(a["region"][3] > 90000 for a in full[:3])
    monkeypatch.setenv("C4GPU_WINDOW_HOPS", "1")
    one = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=100)]
    err = capfd.readouterr().err
    assert " 0 of" not in err.split("windowed region pass")[-1].split("\n")[0], err[-1500:]
    assert full == one


def test_packed_16_bit_score_pass_agrees_with_the_32_bit_pass(eng, monkeypatch, capfd):
    """The score pass with column dumps runs two jobs per lane in packed 16-bit halves (c4_viterbi16_kernel.h) where every
    score fits; C4GPU_PK16=0 keeps the 32-bit kernel.  Same alignments either way on a ragged batch with an odd number of
    jobs (two jobs of a lane of different sizes), tiny dump intervals, and an intron longer than the 15-bit length counter
    (45 000 columns: the counter saturates, the intron stays valid); one pair also against the oracle."""
    rng = random.Random(77)
    model = ex.Model("est2genome")
    pairs = []
    for k, (ql, tl) in enumerate([(900, 30000), (400, 52000), (1000, 9000), (640, 30000), (1000, 100000), (130, 20000), (777, 41000)]):
        q, t = _seeded_pairs(rng, "est2genome", ql, tl, 1)[0]
        pairs.append((q, t))
    q = _rand(rng, 800)
    pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
    pairs.append((_rand(rng, 500), _rand(rng, 25000)))                         # unrelated; nine jobs: the last lane pair is half empty
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_PK16_NW8", "0")      # (small launches would take the eight-wave shape of the score pass)
    res = {}
    # 1, 3, 4: the forms of the packed pass (c4_viterbi16_kernel.h: VAR 1 = default, VAR 0 = every instruction its own asm
    # statement, VAR 2 = progress counters between the cooperating waves instead of a barrier per chunk)
    for pk in ("1", "3", "4", "0"):
        monkeypatch.setenv("C4GPU_PK16", pk)
        res[pk] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=20)]
        err = capfd.readouterr().err
        assert "windowed region pass" in err
        # the default form writes its dumps as 16-bit rows for the packed windows behind it (kpk16d; C4GPU_WIN16=0: kpk16b)
        # (... from LDS-fed column loops where queries and residue codes allow it: kpk16f, the test below)
        assert ("kpk16f_est2genome" in err) == (pk == "1") and ("kpk16_est2genome" in err) == (pk == "3"), err[-1500:]
        assert ("kwin16_est2genome" in err) == (pk == "1"), err[-1500:]
        assert ("kpk16c_est2genome" in err) == (pk == "4"), err[-1500:]
    assert res["1"] == res["0"] and res["3"] == res["0"] and res["4"] == res["0"]
    ops = [model.c.transitions[t].label for t, n in res["1"][7]["ops"] if n >= 45000]
    assert ops == [6], "the long intron is not in the alignment"              # C4_Label_INTRON
    q, t = pairs[2]
    assert res["1"][2] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=32, threshold=20)
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "6")
    monkeypatch.setenv("C4GPU_PK16", "1")
    small = pairs[:4] + pairs[5:6]
    a = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16", "3")
    a2 = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16", "4")
    a4 = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16", "0")
    b = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    assert a == b and a2 == b and a4 == b


def test_staged_packed_score_pass_agrees_with_the_plain_one(eng, monkeypatch, capfd):
    """The packed score pass feeds its column loop from LDS alone (c4_viterbi16_kernel.h, IO 1: column stage refilled per chunk,
    query profile per strip, every strip boundary a ring, the waves three chunks apart) where every query fits the four strips
    of a workgroup and the targets hold at most six residue codes; C4GPU_PK16_IO=0 keeps the form that loads per step.  Same
    alignments either way and as the 32-bit pass on a ragged batch with an odd number of jobs (two jobs of a lane with
    different target lengths and offsets: the stage clamps each), N in queries and targets (five codes), a 45 000-column
    intron, queries of 1 to 1 023 rows (one to four strips, idle waves), with the default and with tiny dump intervals; a
    seventh residue code in a target or a query of 1 024 rows sends the launch to the plain form."""
    rng = random.Random(4242)
    model = ex.Model("est2genome")
    pairs = []
    for k, (ql, tl) in enumerate([(1023, 30000), (400, 52000), (1000, 9000), (640, 30011), (1000, 100000), (130, 20000), (777, 41000),
                                  (64, 5000), (257, 12345), (1, 3000)]):
        q, t = _seeded_pairs(rng, "est2genome", ql, tl, 1)[0]
        if k % 2:                                            # N in the target and in the query
            i, j = rng.randrange(len(t) - 40), rng.randrange(max(1, len(q) - 3))
            t = t[:i] + "N" * 7 + t[i + 7:]
            q = q[:j] + "N" + q[j + 1:]
        pairs.append((q, t))
    q = _rand(rng, 800)
    pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_PK16_NW8", "0")      # (a launch this small would take the eight-wave shape: its own test below)
    res = {}
    # C4GPU_PK16_IO: 2 (the default) = the LDS-fed form with progress counters between its cooperating waves (kpk16f), 1 = the same
    # with a barrier per chunk (kpk16e), 0 = the form that loads per step (kpk16d)
    for io, pk in (("1", "1"), ("2", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("C4GPU_PK16_IO", io)
        monkeypatch.setenv("C4GPU_PK16", pk)
        res[io, pk] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=20)]
        err = capfd.readouterr().err
        assert ("kpk16e_est2genome" in err) == (io == "1" and pk == "1"), err[-1500:]
        assert ("kpk16f_est2genome" in err) == (io == "2"), err[-1500:]
        assert ("kpk16d_est2genome" in err) == (io == "0"), err[-1500:]
        if io in ("1", "2") and pk == "1":
            assert "kernel kpk16%s_est2genome: 3 workgroups per CU" % ("e" if io == "1" else "f") in err, err[-1500:]
    assert res["1", "1"] == res["2", "1"] == res["0", "1"] == res["1", "0"]
    assert sum(1 for a in res["1", "1"] if a) >= 9
    for k in (2, 7, 9):
        q, t = pairs[k]
        assert res["1", "1"][k] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=32, threshold=20)
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "6")
    small = pairs[:4] + pairs[5:10]
    monkeypatch.delenv("C4GPU_PK16_IO"); monkeypatch.setenv("C4GPU_PK16", "1")
    a = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    assert "kpk16f_est2genome" in capfd.readouterr().err
    monkeypatch.setenv("C4GPU_PK16", "0")
    b = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    assert a == b
    monkeypatch.delenv("C4GPU_SEED_KSHIFT")
    monkeypatch.setenv("C4GPU_PK16", "1")
    # six residue codes (A C G T N R): the profile is full
    q, t = pairs[4]
    six = pairs[:4] + [(q, t[:2000] + "RRAR" + t[2004:])] + pairs[5:]
    e = [x.as_dict() if x else None for x in eng.find_path(model, six, dpmemory=32, threshold=20)]
    assert "kpk16f_est2genome" in capfd.readouterr().err
    monkeypatch.setenv("C4GPU_PK16", "0")
    assert e == [x.as_dict() if x else None for x in eng.find_path(model, six, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16", "1")
    capfd.readouterr()
    # seven residue codes (A C G T N R Y): no profile for them; a query of 1 024 rows: five strips
    q, t = pairs[3]
    many = pairs[:3] + [(q, t[:100] + "RY" + t[102:])] + pairs[4:]
    c = [x.as_dict() if x else None for x in eng.find_path(model, many, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    # (round 5: seven and eight codes take the staged form with the eight-code profile, kpk16i; C4GPU_PK16_C8=0: the per-step form)
    assert "kpk16i_est2genome" in err and "kpk16e_est2genome" not in err and "kpk16f_est2genome" not in err, err[-1500:]
    monkeypatch.setenv("C4GPU_PK16_C8", "0")
    assert c == [x.as_dict() if x else None for x in eng.find_path(model, many, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    assert "kpk16d_est2genome" in err and "kpk16i_est2genome" not in err, err[-1500:]
    monkeypatch.delenv("C4GPU_PK16_C8")
    # nine codes (K, M beside them): no code table at all -- the 32-bit dumps and windows behind the packed score pass
    q9, t9 = pairs[3]
    nine = pairs[:3] + [(q9, t9[:100] + "RYKM" + t9[104:])] + pairs[4:]
    c9 = [x.as_dict() if x else None for x in eng.find_path(model, nine, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    assert "kpk16i_est2genome" not in err and "kwin16" not in err, err[-1500:]
    monkeypatch.setenv("C4GPU_PK16", "0")
    assert c9 == [x.as_dict() if x else None for x in eng.find_path(model, nine, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16", "1")
    capfd.readouterr()
    q, t = _seeded_pairs(rng, "est2genome", 1024, 100000, 1)[0]           # a target long enough for the windowed route
    tall = pairs + [(q, t)]
    d = [x.as_dict() if x else None for x in eng.find_path(model, tall, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    # (five strips of 256 rows = four of 384: the six-rows-per-lane form of the staged pass, tests/test_gpu_pk16_edges.py)
    assert "kpk16h_est2genome" in err and "kpk16e_est2genome" not in err and "kpk16f_est2genome" not in err, err[-1500:]
    monkeypatch.setenv("C4GPU_PK16_R6", "0")
    assert d == [x.as_dict() if x else None for x in eng.find_path(model, tall, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    assert "kpk16d_est2genome" in err and "kpk16h_est2genome" not in err, err[-1500:]
    monkeypatch.delenv("C4GPU_PK16_R6")
    monkeypatch.setenv("C4GPU_PK16", "0")
    assert c == [x.as_dict() if x else None for x in eng.find_path(model, many, dpmemory=32, threshold=20)]
    assert d == [x.as_dict() if x else None for x in eng.find_path(model, tall, dpmemory=32, threshold=20)]



def test_staged_packed_score_pass_on_eight_waves_of_two_rows(eng, monkeypatch, capfd):
    """A launch with at most one pair of jobs per compute unit (the 512-pair shard of a strong-scaled run, and every smaller
    call) runs the LDS-fed packed score pass on EIGHT cooperating waves of two rows per lane (kpk16g: the same 1 024 rows per
    workgroup, twice the waves); C4GPU_PK16_NW8=0 keeps four waves of four rows, =1 forces eight.  Ragged batch, odd job count,
    one to eight strips of 128 rows (idle waves), N in queries and targets, a 45 000-column intron, tiny dump intervals: the
    same alignments as the four-wave form and the 32-bit pass, and EVERY pair the oracle finishes in seconds against the
    oracle."""
    rng = random.Random(777)
    model = ex.Model("est2genome")
    pairs = []
    for k, (ql, tl) in enumerate([(1023, 30000), (400, 52000), (1000, 9000), (640, 30011), (1000, 100000), (130, 20000), (897, 41000),
                                  (64, 5000), (257, 12345), (1, 3000), (128, 7000), (129, 7001), (1022, 4000)]):
        q, t = _seeded_pairs(rng, "est2genome", ql, tl, 1)[0]
        if k % 2:
            i, j = rng.randrange(len(t) - 40), rng.randrange(max(1, len(q) - 3))
            t = t[:i] + "N" * 7 + t[i + 7:]
            q = q[:j] + "N" + q[j + 1:]
        pairs.append((q, t))
    q = _rand(rng, 800)
    pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
    monkeypatch.setenv("C4GPU_TRACE", "1")
    res = {}
    for nw8, pk in ((None, "1"), ("1", "1"), ("0", "1"), ("0", "0")):
        if nw8 is None:
            monkeypatch.delenv("C4GPU_PK16_NW8", raising=False)
        else:
            monkeypatch.setenv("C4GPU_PK16_NW8", nw8)
        monkeypatch.setenv("C4GPU_PK16", pk)
        res[nw8, pk] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=20)]
        err = capfd.readouterr().err
        assert ("kpk16g_est2genome" in err) == (nw8 in (None, "1") and pk == "1"), err[-1500:]
        assert ("kpk16f_est2genome" in err) == (nw8 == "0" and pk == "1"), err[-1500:]
    assert res[None, "1"] == res["1", "1"] == res["0", "1"] == res["0", "0"]
    checked = 0
    for k, (q, t) in enumerate(pairs):
        if (len(q) + 1) * (len(t) + 1) <= 1.3e7:
            assert res[None, "1"][k] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=32, threshold=20), k
            checked += 1
    assert checked >= 8
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "6")
    monkeypatch.delenv("C4GPU_PK16_NW8", raising=False)
    monkeypatch.setenv("C4GPU_PK16", "1")
    small = pairs[:4] + pairs[5:13]
    a = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]
    assert "kpk16g_est2genome" in capfd.readouterr().err
    monkeypatch.setenv("C4GPU_PK16", "0")
    assert a == [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=20)]

def test_packed_16_bit_region_windows_agree_with_the_32_bit_windows(eng, monkeypatch, capfd):
    """The region windows run two windows per lane in packed 16-bit halves (c4_win16_kernel.h) behind the packed score pass,
    from its 16-bit dumps, each chain over the component of the state its END was entered from (one strand of est2genome);
    C4GPU_WIN16=0 keeps the 32-bit windows.  Same alignments either way, in every shape of the packed kernel, on a ragged
    batch with an odd number of jobs on either strand (reverse-complemented pairs: the gene on the other strand, introns
    CT..AC, END entered from the reverse match state), chains of one to thirteen windows, an intron longer than the 15-bit
    length counter, a pair below the threshold; then with tiny dump intervals (a path crosses dozens of dumps, introns jump
    over dumped columns); pairs also against the oracle."""
    rng = random.Random(1606)
    model = ex.Model("est2genome")
    comp = str.maketrans("ACGTN", "TGCAN")
    pairs = []
    for k, (ql, tl) in enumerate([(900, 36000), (400, 52000), (1000, 9000 + 33000), (640, 33000), (1000, 100000), (450, 34000),
                                  (777, 41000), (1000, 70000), (520, 66000)]):
        q, t = _seeded_pairs(rng, "est2genome", ql, tl, 1)[0]
        if k % 3 == 1: q, t = q.translate(comp)[::-1], t.translate(comp)[::-1]       # the gene on the reverse strand
        pairs.append((q, t))
    q = _rand(rng, 800)
    pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
    q = _rand(rng, 990)                                    # a 90 kb intron on the reverse strand: thirteen windows back
    t = _rand(rng, 1500) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 90000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)
    pairs.append((q.translate(comp)[::-1], t.translate(comp)[::-1]))
    pairs.append((_rand(rng, 500), _rand(rng, 35000)))                         # unrelated: below the threshold
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_PK16_NW8", "0")      # (small launches would take the eight-wave shape of the score pass)
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "13")
    res = {}
    shapes = ("1", "2", "3", "4", "5", "6", "7", "8", "9", "10")     # 10: two waves at three per SIMD; 1: chosen by the jobs; 5 ... 8: the strips of a window on 4 / 8 / 4 / 2 cooperating waves; 9: one wave
    for w in shapes + ("0",):
        monkeypatch.setenv("C4GPU_WIN16", w)
        res[w] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=100)]
        err = capfd.readouterr().err
        assert "windowed region pass" in err and " 0 of " in err.split("windowed region pass")[-1].split("\n")[0], err[-1500:]
        # (the score pass behind them writes 16-bit dumps: kpk16f, its LDS-fed form, for these queries and residue codes)
        assert ("kwin16_est2genome" in err) == (w != "0") and ("kpk16f_est2genome" in err) == (w != "0"), err[-1500:]
        assert ("kmw2_est2genome_region_local_pack_seed2" in err) == (w == "0"), err[-1500:]
    for w in shapes:
        assert res[w] == res["0"], w
    assert all(r is not None for r in res["0"][:-1])          # (the unrelated pair has a chance alignment above the threshold or not)
    # both strands were there: an intron-labelled run of a reverse-strand pair goes through the reverse intron state
    strands = set()
    for r in res["1"][:-1]:
        if r is None: continue
        states = {model.c.transitions[t].output for t, n in r["ops"]}
        strands.add("rev" if 9 in states or 5 in states else "fwd")
    assert strands == {"fwd", "rev"}, strands
    assert res["1"][10]["region"][3] > 90000 and res["1"][9]["region"][3] > 45000
    for k in (2, 3):
        q, t = pairs[k]
        assert res["1"][k] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=32, threshold=100)
    for kshift in ("6", "9"):
        monkeypatch.setenv("C4GPU_SEED_KSHIFT", kshift)
        small = [pairs[0], pairs[1], pairs[3], pairs[5], pairs[11]]
        got = {}
        for w in ("1", "3", "5", "6", "9", "0"):
            monkeypatch.setenv("C4GPU_WIN16", w)
            got[w] = [x.as_dict() if x else None for x in eng.find_path(model, small, dpmemory=32, threshold=100)]
            err = capfd.readouterr().err
            assert ("kwin16_est2genome" in err) == (w != "0"), err[-1500:]
        assert all(got[w] == got["0"] for w in ("1", "3", "5", "6", "9")), kshift


@pytest.mark.parametrize("model_type,dpm", [("affine:local", 32), ("affine:local", 1), ("affine:local", 0),
                                            ("est2genome", 32), ("est2genome", 1), ("est2genome", 0),
                                            ("protein2dna", 1), ("protein2genome", 1)])
def test_device_route_of_the_sub_alignments_gives_the_host_route_results(eng, monkeypatch, capfd, model_type, dpm):
    """Optimal_find_path_reduced_space (optimal.c:160-345): the sub-alignment jobs of a checkpoint pass are listed, run,
    verified and stitched on the device (fused_reduced_paths); C4GPU_FUSED=0 keeps the host route.  Same alignments either
    way on ragged batches, one pair also against the oracle; at -D 32 every pair finishes on the device route (at -D 0 a
    section can itself need checkpoints: those pairs are flagged and take the host route, recursion included)."""
    rng = random.Random(20260929 + dpm)
    model = ex.Model(model_type)
    if model_type.startswith("protein"):
        proteins, contig, _ = workloads.protein_vs_contig(6, 120, 60000 if model_type == "protein2genome" else 9000, seed=511 + dpm,
                                                          introns=model_type == "protein2genome")
        pairs = [(p, contig) for p in proteins]
        enc = lambda s: s
    else:
        pairs = []
        for ql, tl in [(1000, 1100), (300, 5000), (77, 900), (640, 2000), (1200, 1300), (1000, 12000), (150, 150)]:
            pairs += _seeded_pairs(rng, model_type, ql, tl, 1)
        pairs.append((_rand(rng, 400), _rand(rng, 700)))                      # unrelated: below the threshold
        enc = lambda s: s.encode()
    monkeypatch.setenv("C4GPU_TRACE", "1")
    res = {}
    # C4GPU_CONT_FREE=0: the continuation kernels that keep the row-0 validity mask (c4_viterbi_kernel.h, CONT && LOCAL)
    for sw, free in (("1", "1"), ("1", "0"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("C4GPU_FUSED", sw)
        monkeypatch.setenv("C4GPU_CONT_FREE", free)
        res[sw + free] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=dpm, threshold=30)]
        err = capfd.readouterr().err
        finished = [ln for ln in err.splitlines() if "pairs finished on the device route" in ln]
        if sw == "0":
            assert not finished
        else:
            assert finished, err[-2000:]
            assert ("_ckpt_cont_local," in err) == (free == "1") and ("_path_cont_local" in err) == (free == "1"), err[-2000:]
            if dpm == 32:
                words = finished[0].split("fused:")[1].split()
                assert words[0] == words[2], finished[0]                      # "N of N pairs finished ..."
    assert res["11"] == res["00"] and res["10"] == res["00"] and res["01"] == res["00"]
    res["1"] = res["11"]
    assert any(r is not None for r in res["1"])
    k = next(i for i, r in enumerate(res["1"]) if r is not None)
    q, t = pairs[k]
    assert res["1"][k] == oracle_lib.find_path(model.c, model.params, enc(q), enc(t), dpmemory=dpm, threshold=30)


@pytest.mark.parametrize("dpm", [32, 1, 0])
def test_packed_16_bit_checkpoint_pass_agrees_with_the_32_bit_pass(eng, monkeypatch, capfd, dpm):
    """The checkpoint pass of the device route (Viterbi_Checkpoint_process viterbi.c:605-631, traceback :537-601) runs two jobs
    per lane in packed 16-bit halves (c4_ckpt16_kernel.h) wherever score, checkpoint payload and intron length fit;
    C4GPU_CK16=0 keeps the 32-bit kernel.  Same alignments either way on a ragged batch with an odd number of jobs (the two
    jobs of a lane differ in rows, columns and checkpoint columns; queries of one to seven strips), with an intron longer than
    the 15-bit length counter (45 000 columns: its saturated shadow is "equivalent", final_cell_equiv) and with every
    shape of the packed kernel; pairs also against the oracle.  C4GPU_CK16_TMAX keeps some jobs of the launch on the 32-bit
    kernel: both kernels then serve one call."""
    rng = random.Random(4100 + dpm)
    model = ex.Model("est2genome")
    pairs = []
    sizes = [(900, 30000), (400, 52000), (1000, 9000), (640, 30000), (130, 20000), (777, 41000), (1300, 7000), (190, 2500),
             (64, 3000), (1000, 100000)]
    if dpm == 0:
        sizes = [(300, 5000), (77, 900), (640, 2000), (1000, 1300), (1000, 12000), (150, 600), (450, 3000)]
    for ql, tl in sizes:
        pairs += _seeded_pairs(rng, "est2genome", ql, tl, 1)
    q = _rand(rng, 800)
    pairs.append((q, _rand(rng, 3000) + _mutate(rng, q[:400], 0.03) + "GT" + _rand(rng, 45000) + "AG" + _mutate(rng, q[400:], 0.03) + _rand(rng, 2000)))
    pairs.append((_rand(rng, 500), _rand(rng, 25000)))                          # unrelated: below the threshold
    monkeypatch.setenv("C4GPU_TRACE", "1")
    res, fin = {}, {}
    # C4GPU_CK16 = 1..4: the shapes of the rooted form (one strand's states: the jobs whose region pass said where END was
    # entered from), 5..7: the strips of a pair of jobs on 4 / 2 / 3 cooperating waves (1: chosen by the jobs' strips, 8: one wave, 9: four waves at three per SIMD); C4GPU_CK16_ROOT=0: every job on the
    # form that computes all inner states
    rooted_seen = 0
    for ck, tmax, root in (("1", None, "1"), ("2", None, "1"), ("3", None, "1"), ("4", None, "1"), ("5", None, "1"),
                           ("6", None, "1"), ("7", None, "1"), ("8", None, "1"), ("9", None, "1"), ("1", None, "0"),
                           ("1", "20000", "1"), ("0", None, "1")):
        monkeypatch.setenv("C4GPU_CK16", ck)
        monkeypatch.setenv("C4GPU_CK16_ROOT", root)
        if tmax: monkeypatch.setenv("C4GPU_CK16_TMAX", tmax)
        else: monkeypatch.delenv("C4GPU_CK16_TMAX", raising=False)
        key = ck + (tmax or "") + ("" if root == "1" else "a")
        res[key] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=dpm, threshold=30)]
        err = capfd.readouterr().err
        fused = [ln for ln in err.splitlines() if "fused: packed checkpoint kernels" in ln]
        assert fused, err[-2000:]
        words = fused[0].split("packed checkpoint kernels")[1].replace(",", " ").split()   # "<rooted> for <n> <all> for <n> of <n> jobs"
        n16r, n16a, n_all = int(words[2]), int(words[5]), int(words[7])
        n32 = n_all - n16r - n16a
        if ck == "0":
            assert words[0] == "-" and words[3] == "-" and n16r + n16a == 0
        else:
            assert n16r + n16a >= 2, fused[0]
            assert (words[0].startswith("kck16r_est2genome")) == (n16r > 0) and (words[3].startswith("kck16_est2genome")) == (n16a > 0), fused[0]
            assert (n32 > 0) == (tmax is not None), fused[0]
            if root == "0": assert n16r == 0, fused[0]
            rooted_seen += n16r
        finished = [ln for ln in err.splitlines() if "pairs finished on the device route" in ln][0].split("fused:")[1].split()
        fin[key] = (finished[0], finished[2])                 # "N of M pairs finished ..."
    # the 100 kb targets take the windowed region pass, which names the root (the short batch of -D 0 only where the context's
    # earlier batches left the windowed form switched on: c4gpu_ctx::window_rate)
    if dpm != 0: assert rooted_seen > 0
    for k, v in res.items():
        assert v == res["0"], k
        assert fin[k] == fin["0"], (k, fin)            # the packed pass sends no pair to the host route that the 32-bit pass keeps
    long_pair = len(pairs) - 2
    ops = [model.c.transitions[t].label for t, n in res["1"][long_pair]["ops"] if n >= 45000]
    assert ops == [6], "the long intron is not in the alignment"              # C4_Label_INTRON
    for k in (2, 6, 7):
        if k < len(pairs) and res["1"][k] is not None and len(pairs[k][1]) <= 12000:
            q, t = pairs[k]
            assert res["1"][k] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dpm, threshold=30)


@pytest.mark.parametrize("model_type", ["est2genome", "affine:local", "protein2genome"])
def test_two_launch_lanes_give_the_one_lane_results(eng, monkeypatch, model_type):
    """A large batch is cut into two halves of equal work that walk through the passes of Optimal_find_path on two streams
    from two host threads over the same resident sequences (c4_engine_find_path.inc, find_path_lanes; default for 2 048 pairs and
    more).  C4GPU_LANES=2 forces the cut on a small ragged batch: same alignments as one lane, through the one-shot entry
    point and through a resident batch (whose kernel statistics then count the launches of both lanes)."""
    rng = random.Random(991)
    model = ex.Model(model_type)
    if model_type == "protein2genome":
        proteins, contig, _ = workloads.protein_vs_contig(7, 150, 80000, seed=77, introns=True)
        pairs = [(p, contig) for p in proteins]
    else:
        pairs = []
        for ql, tl in [(900, 30000), (400, 52000), (1000, 9000), (640, 30000), (130, 20000), (777, 41000), (1000, 1200)]:
            pairs += _seeded_pairs(rng, model_type, ql, tl, 1)
        pairs.append((_rand(rng, 500), _rand(rng, 25000)))
    res = {}
    for lanes in ("1", "2"):
        monkeypatch.setenv("C4GPU_LANES", lanes)
        res[lanes] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=30)]
        batch = ex.ResidentBatch(eng, model, pairs)
        batch.kernel_stats(0, reset=True)
        batch.kernel_stats(2, reset=True)
        batch.run(2, 32, 30)
        batch.run(2, 32, 30)
        res["b" + lanes] = [(lambda a: a.as_dict() if a else None)(batch.alignment(i)) for i in range(len(pairs))]
        res["n" + lanes] = batch.kernel_stats(0)["launches"] + batch.kernel_stats(2)["launches"]
        batch.close()
    assert res["1"] == res["2"] and res["b1"] == res["1"] and res["b2"] == res["1"]
    assert any(r is not None for r in res["1"])
    assert res["n2"] > res["n1"] > 0                  # each lane launches its own passes


def test_memory_rule_on_the_device_is_the_host_rule(eng):
    """Viterbi_use_reduced_space / Viterbi_checkpoint_rows (viterbi.c:128-150,207-218) decide which passes a region gets;
    the device lists the sub-alignments of a checkpoint pass and evaluates the rule itself (csrc/c4_memrule.h compiled for
    host and device).  Same decisions as the host functions (which test_abi.py pins on the oracle) for region sizes from
    the smallest to the largest a sequence can have, at every --dpmemory the tests use."""
    import ctypes as C
    lib = _abi.load()
    rng = random.Random(4242)
    sizes = [(q, t) for q in (0, 1, 5, 6, 7, 12, 13, 100, 1000, 1001, 65535, 1 << 20, (1 << 30) - 1)
             for t in (0, 1, 6, 7, 12, 13, 14, 250, 1000, 4096, 100000, 10 ** 7, (1 << 30) - 1)]
    sizes += [(rng.randint(0, 5000), rng.randint(0, 200000)) for _ in range(3000)]
    sizes += [(rng.randint(0, 1 << 30), rng.randint(0, 1 << 30)) for _ in range(500)]
    n = len(sizes)
    ql = (C.c_int32 * n)(*[s[0] for s in sizes])
    tl = (C.c_int32 * n)(*[s[1] for s in sizes])
    for mt in ("affine:local", "est2genome", "protein2genome"):
        model = ex.Model(mt)
        for dpm in (0, 1, 32, 512, 2047):
            red, rows = (C.c_int32 * n)(), (C.c_int32 * n)()
            assert lib.c4gpu_memrule_device(eng.ctx, model.c, dpm, ql, tl, n, red, rows) == 0
            for k, (q, t) in enumerate(sizes):
                r = _abi.Region(0, 0, q, t)
                assert red[k] == lib.c4gpu_use_reduced_space(model.c, r, dpm), (mt, dpm, q, t)
                if red[k] and rows[k] >= 0 and dpm > 0:
                    assert rows[k] == lib.c4gpu_checkpoint_rows(model.c, r, dpm), (mt, dpm, q, t)



def test_sub_alignments_on_the_kernel_of_their_root(eng, monkeypatch):
    """The sub-alignment pass of the device route runs a job that names its alignment's root (the state END is entered from:
    known where the packed score pass found the end cell) on the path kernel of that root's COMPONENT -- one strand of est2genome,
    half the transitions (c4_viterbi_kernel.h: COMP, BYROOT); C4GPU_BYROOT=0 hands the jobs over without a root (whole model).  The
    continuation's seeding keeps its place in the transition order (viterbi.c:705-714: at the first transition out of START by
    id, i.e. behind 6 -> 5 / 7 -> 5 and in front of 3 -> 2 / 4 -> 2 whichever strand the job is on), which matters where a
    checkpoint cell sits in a gap state.  Forward and reverse strands, indels at 8 % (checkpoint cells in gap states), tiny dump
    intervals so that short targets take the route, -D 1 (many sections): both forms give the same alignments, and every pair the
    oracle's."""
    rng = random.Random(2718)
    model = ex.Model("est2genome")
    pairs = []
    for k, (ql, tl) in enumerate([(700, 5000), (900, 7000), (560, 4000), (1000, 6000), (640, 9000), (800, 5200), (1023, 4800), (530, 3000)]):
        q = _rand(rng, ql)
        c1, c2 = ql // 3, 2 * ql // 3
        d5, d3 = ("CT", "AC") if k % 2 else ("GT", "AG")        # odd pairs: introns of the other strand (the reverse-strand states)
        gene = (_mutate(rng, q[:c1], 0.08) + d5 + _rand(rng, rng.randint(80, 900)) + d3 + _mutate(rng, q[c1:c2], 0.08) +
                d5 + _rand(rng, rng.randint(80, 600)) + d3 + _mutate(rng, q[c2:], 0.08))
        t = _rand(rng, rng.randint(50, 400)) + gene
        pairs.append((q, t + _rand(rng, max(10, tl - len(t)))))
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "8")
    got = {}
    for dp in (32, 1):
        for byroot in ("1", "0"):
            monkeypatch.setenv("C4GPU_BYROOT", byroot)
            got[dp, byroot] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=dp, threshold=20)]
        assert got[dp, "1"] == got[dp, "0"], dp
        for k, (q, t) in enumerate(pairs):
            assert got[dp, "1"][k] == oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dp, threshold=20), (dp, k)
    assert sum(1 for a in got[32, "1"] if a) == len(pairs)
    # both strands occur: the vulgar line labels the splice sites of a forward intron 5 .. 3 and of a reverse one 3 .. 5
    fwd = sum(1 for a in got[32, "1"] if " 5 0 2 I " in a["vulgar"]), sum(1 for a in got[32, "1"] if " 3 0 2 I " in a["vulgar"])
    assert fwd[0] >= 2 and fwd[1] >= 2, fwd
