"""BASELINE.json's other configurations as parity cases at (or near) their full sizes on one MI355X:
C2 affine:local 1 kb x 1 kb, C3 protein2dna 500 aa x one shared 1 Mb contig, C5-shaped protein2genome
against a shared contig.  Where the oracle cannot finish a full-size rectangle in seconds, parity goes
through a size-independent property: a local alignment whose optimal path lies inside a window of the
contig is the window's alignment shifted by the window offset (same ops, same score)."""
import os
import pytest

import exonerate_amd as ex
from exonerate_amd import workloads
import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def test_c2_affine_local_1kb_batch(eng):
    """C2 at BASELINE's size: 4 096 pairs of 1 kb x 1 kb in one batch; every 64th checked against the oracle (reduced
    space at -D 32: ~190 sub-alignments per pair)."""
    model = ex.Model("affine:local")
    pairs = workloads.affine_dna_pairs(4096, 1000)
    alns = eng.find_path(model, pairs, dpmemory=32)
    assert all(a is not None for a in alns)
    for i in range(0, 4096, 64):
        q, t = pairs[i]
        assert alns[i].as_dict() == oracle_lib.find_path(model.c, model.params, q, t, dpmemory=32), i


def _check_against_windows(eng, model, proteins, contig, places, margin, check_every=1):
    pairs = [(p, contig) for p in proteins]                  # one shared buffer: uploaded once
    alns = eng.find_path(model, pairs, dpmemory=32)
    assert all(a is not None for a in alns)
    for k, (p, place, a) in enumerate(zip(proteins, places, alns)):
        if k % check_every or place is None:
            continue
        g0, g1 = place
        w0, w1 = max(0, g0 - margin), min(len(contig), g1 + margin)
        exp = oracle_lib.find_path(model.c, model.params, p, contig[w0:w1], dpmemory=32)
        assert a is not None and exp is not None
        assert a.score == exp["score"]
        assert [list(o) for o in a.ops] == exp["ops"]
        r = exp["region"]
        assert list(a.region) == [r[0], r[1] + w0, r[2], r[3]]
        assert g0 - 30 <= a.region[1] and a.region[1] + a.region[3] <= g1 + 30


def test_c3_protein2dna_shared_megabase_contig(eng):
    """C3 at BASELINE's size: 1 024 proteins of 500 aa against ONE 1 Mb contig (T = 10^6 columns per job; 5 x 10^11 cells
    per pass); every fourth protein has its gene in the contig, every 64th is checked against the oracle on a window."""
    proteins, contig, places = workloads.protein_vs_contig(1024, 500, 1000000, plant_every=4)
    _check_against_windows(eng, ex.Model("protein2dna"), proteins, contig, places, 1500, check_every=64)


def test_c5_shape_protein2genome_shared_contig(eng):
    """C5 shape at reduced contig length: proteins with intron-split genes against one shared 300 kb contig."""
    proteins, contig, places = workloads.protein_vs_contig(8, 300, 300000, seed=20260935, introns=True)
    _check_against_windows(eng, ex.Model("protein2genome"), proteins, contig, places, 1000)


@pytest.mark.parametrize("name,model_name,n,tlen", [("c5", "protein2genome", 256, 10000000), ("c3", "protein2dna", 1024, 1000000)])
def test_protein_configs_at_full_size_against_the_reference_records(eng, name, model_name, n, tlen):
    """BASELINE config 5's exhaustive shape at its size -- 256 proteins of 300 aa against ONE 10 Mb chromosome, protein2genome,
    7.7 x 10^11 first-pass cells -- and config 3 -- 1 024 proteins of 500 aa against ONE 1 Mb contig, protein2dna --: every 64th
    alignment (score, region, operations) against the records the REFERENCE made for them (tests/golden/bench_configs.json:
    refdump on each sampled protein and its window, tools/make_bench_golden.py).  Until round 5 this comparison lived in
    bench.py only."""
    import json
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_configs.json")))[name]
    got_model, pairs, _ = workloads.bench_config(name)
    assert got_model == model_name and len(pairs) == n and len(pairs[0][1]) == tlen
    b = ex.ResidentBatch(eng, ex.Model(model_name), pairs)
    try:
        b.run(2)
        assert len(want["sample"]) >= 4
        for rec in want["sample"]:
            a = b.alignment(rec["pair"])
            assert a is not None and a.score == rec["score"], rec["pair"]
            assert list(a.region) == rec["region"] and [list(o) for o in a.ops] == rec["ops"], rec["pair"]
    finally:
        b.close()


def test_shared_buffers_give_the_same_results_as_private_copies(eng):
    model = ex.Model("est2genome")
    base = workloads.est2genome_pairs(4, 300, 40000)
    shared = [(q, base[0][1]) for q, _ in base]               # same target object for all
    private = [(q, bytes(bytearray(base[0][1]))) for q, _ in base]
    a = eng.find_path(model, shared)
    b = eng.find_path(model, private)
    assert [x.as_dict() if x else None for x in a] == [x.as_dict() if x else None for x in b]
    assert a[0] is not None


def test_c4_north_star_batch_against_the_reference_binary(eng):
    """C4 at BASELINE's size (the configuration bench.py times): 4 096 cDNAs of 1 kb against their 100 kb windows in one
    batch, est2genome, -D 32.  Every pair must align; the vulgar lines of a sample — one pair per two host cores, up to 32,
    spread over the batch — are compared with the reference's own compiled exonerate run here (oracle/_ref, travels as a
    binary).  Round 6: four of the sampled cDNAs also on their REVERSE strand -- what the reference aligns by
    default for a DNA query (fastapipe.c:42-44): chance alignments that span most of their window, i.e. the wide-region route
    (two dozen window hops, a checkpoint pass over the whole rectangle) -- against the reference run on the reverse-complemented
    query."""
    import os, subprocess, tempfile
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "exonerate-compiled")
    if not os.path.exists(exe):
        pytest.skip("the reference binary is built in the build container")
    model = ex.Model("est2genome")
    pairs = workloads.est2genome_pairs(4096, 1000, 100000)
    alns = eng.find_path(model, pairs, dpmemory=32)
    assert all(a is not None for a in alns)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = max(4, min(cores // 2, 32))          # (one reference process per TWO cores: as many as cores made each take twice as long)
    sample = [(k * 4096) // n for k in range(n)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rsample = sample[1::max(1, n // 4)][:4]
    rpairs = [(pairs[k][0].translate(comp)[::-1], pairs[k][1]) for k in rsample]
    ralns = eng.find_path(model, rpairs, dpmemory=32)
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for tag, k, (q, t) in [("f", k, pairs[k]) for k in sample] + [("r", k, rp) for k, rp in zip(rsample, rpairs)]:
            open(os.path.join(d, "q%s%d.fa" % (tag, k)), "w").write(">qy\n%s\n" % q.decode())
            open(os.path.join(d, "t%s%d.fa" % (tag, k)), "w").write(">tg\n%s\n" % t.decode())
            procs.append(subprocess.Popen([exe, "-m", "est2genome", "-E", "yes", "-S", "no", "--revcomp", "no", "--showalignment", "no",
                                           "--showvulgar", "yes", "-V", "0", os.path.join(d, "q%s%d.fa" % (tag, k)),
                                           os.path.join(d, "t%s%d.fa" % (tag, k))],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
        outs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    for k, o in zip(sample, outs[:n]):
        ref = [l.strip() for l in o.splitlines() if l.startswith("vulgar:")]
        assert ref and alns[k].vulgar("qy", "tg") == ref[0], k
    wide = 0
    for k, a, o in zip(rsample, ralns, outs[n:]):
        ref = [l.strip() for l in o.splitlines() if l.startswith("vulgar:")]
        assert ref and a is not None and a.vulgar("qy", "tg") == ref[0], ("reverse strand", k)
        wide += a.region[3] > 50000
    assert wide >= 2                       # (chance alignments across most of the 100 kb window)
