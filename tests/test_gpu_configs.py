"""BASELINE.json's other configurations as parity cases at (or near) their full sizes on one MI355X:
C2 affine:local 1 kb x 1 kb, C3 protein2dna 500 aa x one shared 1 Mb contig, C5-shaped protein2genome
against a shared contig.  Where the oracle cannot finish a full-size rectangle in seconds, parity goes
through a size-independent property: a local alignment whose optimal path lies inside a window of the
contig is the window's alignment shifted by the window offset (same ops, same score)."""
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
    """C2: 1 024 pairs in one batch; every 64th checked against the oracle (reduced space at -D 32)."""
    model = ex.Model("affine:local")
    pairs = workloads.affine_dna_pairs(1024, 1000)
    alns = eng.find_path(model, pairs, dpmemory=32)
    assert all(a is not None for a in alns)
    for i in range(0, 1024, 64):
        q, t = pairs[i]
        assert alns[i].as_dict() == oracle_lib.find_path(model.c, model.params, q, t, dpmemory=32), i


def _check_against_windows(eng, model, proteins, contig, places, margin):
    pairs = [(p, contig) for p in proteins]                  # one shared buffer: uploaded once
    alns = eng.find_path(model, pairs, dpmemory=32)
    for p, (g0, g1), a in zip(proteins, places, alns):
        w0, w1 = max(0, g0 - margin), min(len(contig), g1 + margin)
        exp = oracle_lib.find_path(model.c, model.params, p, contig[w0:w1], dpmemory=32)
        assert a is not None and exp is not None
        assert a.score == exp["score"]
        assert [list(o) for o in a.ops] == exp["ops"]
        r = exp["region"]
        assert list(a.region) == [r[0], r[1] + w0, r[2], r[3]]
        assert g0 - 30 <= a.region[1] and a.region[1] + a.region[3] <= g1 + 30


def test_c3_protein2dna_shared_megabase_contig(eng):
    """C3 shape: 500 aa proteins against ONE 1 Mb contig (T = 10^6 columns per job, unpacked region slots
    not needed: 9 + 20 bits)."""
    proteins, contig, places = workloads.protein_vs_contig(16, 500, 1000000)
    _check_against_windows(eng, ex.Model("protein2dna"), proteins, contig, places, 1500)


def test_c5_shape_protein2genome_shared_contig(eng):
    """C5 shape at reduced contig length: proteins with intron-split genes against one shared 300 kb contig."""
    proteins, contig, places = workloads.protein_vs_contig(8, 300, 300000, seed=20260935, introns=True)
    _check_against_windows(eng, ex.Model("protein2genome"), proteins, contig, places, 1000)


def test_shared_buffers_give_the_same_results_as_private_copies(eng):
    model = ex.Model("est2genome")
    base = workloads.est2genome_pairs(4, 300, 40000)
    shared = [(q, base[0][1]) for q, _ in base]               # same target object for all
    private = [(q, bytes(bytearray(base[0][1]))) for q, _ in base]
    a = eng.find_path(model, shared)
    b = eng.find_path(model, private)
    assert [x.as_dict() if x else None for x in a] == [x.as_dict() if x else None for x in b]
    assert a[0] is not None
