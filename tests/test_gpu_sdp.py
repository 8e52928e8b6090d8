"""SDP on the device (c4gpu_sdp_batch: both Scheduler passes of the seeded flavour, sdp.c:322-341, for a whole batch)
against the reference's own SDP alignments (tests/golden/sdp_*.jsonl, refdump --cmd sdp) and, on larger pairs with many
HSPs, against the oracle (pinned on the same vectors in test_oracle_sdp.py)."""
import random
import pytest

import exonerate_amd as ex
import oracle_lib
from golden_util import load_set
from test_oracle_sdp import SDP_SETS, sdp_case, expected

pytestmark = pytest.mark.gpu

SEEDED = ["sdp_affine_local", "sdp_affine_local_protein", "sdp_protein2dna"]
# the boundary flavour (spans, shadows): every single-pass set (the multipass one has no device form: sdp.c:760-767)
BOUNDARY = ["sdp_est2genome", "sdp_est2genome_drop", "sdp_protein2genome", "sdp_protein2genome_altparams"]
AA = "ARNDCQEGHILKMFPSTWYV"


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name", SEEDED + BOUNDARY)
def test_device_sdp_matches_reference_vectors(eng, name):
    model, par, recs, adv = sdp_case(name)
    assert par["use_boundary"] == (1 if name in BOUNDARY else 0) and par["singlepass"] == 1
    pairs = [(r["query"], r["target"]) for r in recs]
    got = eng.sdp(model, pairs, [r["hsps"] for r in recs], adv[0], adv[1], par["dropoff"], par["threshold"], 4)
    total = 0
    for r, alns in zip(recs, got):
        g = [{"score": a.score, "region": list(a.region), "ops": [list(o) for o in a.ops], "vulgar": a.vulgar(r["id"])}
             for a in alns]
        assert g == expected(r), r["id"]
        total += len(g)
    assert total >= 10


def _mut(rng, s, rate, alphabet):
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


def _hsps(params, match, q, t, w):
    """HSPs of a pair as HSPset_seed_hsp grows them from every shared word (oracle_hsp_set: pinned on reference HSPs)."""
    words = {}
    for i in range(len(q) - w + 1):
        words.setdefault(q[i:i + w], []).append(i)
    seeds = [(i, j) for j in range(len(t) - w + 1) for i in words.get(t[j:j + w], ())]
    seedlen = w
    return oracle_lib.hsp_set(params, match, q.encode(), t.encode(), seedlen, 30 if match == "dna2dna" else 20, 30, seeds)


def test_c1_shape_batch_against_oracle(eng):
    """BASELINE config 1's shape: proteins of ~300 aa against a 10 kaa target holding diverged copies, affine:local,
    24 pairs per launch."""
    rng = random.Random(11)
    params = ex.default_params()
    model = ex.Model("affine:local", query_alphabet=ex.ALPHABET_PROTEIN, target_alphabet=ex.ALPHABET_PROTEIN, params=params)
    pairs, hsps = [], []
    for k in range(24):
        q = "".join(rng.choice(AA) for _ in range(rng.randint(200, 400)))
        t = "".join(rng.choice(AA) for _ in range(rng.randint(0, 3000)))
        for _ in range(rng.randint(1, 3)):
            t += _mut(rng, q[rng.randint(0, 60):], rng.choice([0.08, 0.2]), AA) + "".join(rng.choice(AA) for _ in range(rng.randint(50, 2500)))
        t = t[:10000]
        h = _hsps(params, "protein2protein", q, t, 4)
        if h:
            pairs.append((q, t)); hsps.append(h)
    assert len(pairs) >= 20
    got = eng.sdp(model, pairs, hsps, 1, 1, 50, 60, 5)
    found = 0
    for (q, t), h, alns in zip(pairs, hsps, got):
        ub, exp = oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, 1, 1, 50, True, 60, 5)
        assert ub == 0
        assert [a.as_dict() for a in alns] == exp
        found += len(exp)
    assert found >= 24


def test_dna_and_protein2dna_pairs_against_oracle(eng):
    rng = random.Random(12)
    params = ex.default_params()
    for mt, match, adv, w in (("affine:local", "dna2dna", (1, 1), 11), ("protein2dna", "protein2dna", (1, 3), 4)):
        model = ex.Model(mt, params=params)
        pairs, hsps = [], []
        for k in range(12):
            if mt == "affine:local":
                q = "".join(rng.choice("ACGT") for _ in range(rng.randint(300, 900)))
                t = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 800))) + _mut(rng, q, 0.06, "ACGT") + \
                    "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 800)))
                h = _hsps(params, match, q, t, w)
            else:
                from test_library_fuzz_gpu import CODON
                q = "".join(rng.choice(AA) for _ in range(rng.randint(80, 250)))
                coding = "".join(rng.choice(CODON[a]) for a in _mut(rng, q, 0.05, AA))
                p = rng.randint(10, len(coding) - 10)
                coding = coding[:p] + rng.choice("ACGT") + coding[p:]                     # a frameshift
                t = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 500))) + coding + \
                    "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 500)))
                # protein2dna HSPs: words of 4 aa against the translation in each frame
                rev = {c: a for a, cs in CODON.items() for c in cs}
                words = {}
                for i in range(len(q) - w + 1):
                    words.setdefault(q[i:i + w], []).append(i)
                seeds = []
                for j in range(len(t) - 3 * w + 1):
                    word = "".join(rev.get(t[j + 3 * x:j + 3 * x + 3], "X") for x in range(w))
                    # not the hits more than a query length above the main diagonal: the reference's horizon index is
                    # out of bounds for them (hspset.c:943), so they have no parity target
                    seeds += [(i, j) for i in words.get(word, ()) if j - 3 * i + len(q) >= 0]
                h = oracle_lib.hsp_set(params, match, q.encode(), t.encode(), w, 20, 30, seeds)
            if h:
                pairs.append((q, t)); hsps.append(h)
        got = eng.sdp(model, pairs, hsps, adv[0], adv[1], 50, 50, 4)
        n = 0
        for (q, t), h, alns in zip(pairs, hsps, got):
            ub, exp = oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, adv[0], adv[1], 50, True, 50, 4)
            assert [a.as_dict() for a in alns] == exp, (mt, len(q), len(t))
            n += len(exp)
        assert n >= 6, mt


@pytest.mark.parametrize("seed", range(6))
def test_device_sdp_fuzz_against_oracle(eng, seed):
    """Seeded random batches of the three boundary-free families (tests/sdp_cases.py) against the oracle."""
    from sdp_cases import seeded_fuzz_cases
    for cs in seeded_fuzz_cases(seed):
        model, adv = cs["model"], cs["adv"]
        got = eng.sdp(model, cs["pairs"], cs["hsps"], adv[0], adv[1], cs["dropoff"], cs["threshold"], 4)
        for (q, t), h, alns in zip(cs["pairs"], cs["hsps"], got):
            ub, exp = oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, adv[0], adv[1], cs["dropoff"], True, cs["threshold"], 4)
            assert [a.as_dict() for a in alns] == exp, (seed, cs["kind"], cs["variant"], cs["dropoff"], cs["threshold"], len(q), len(t), len(h))


@pytest.mark.parametrize("seed", range(6))
def test_device_sdp_boundary_fuzz_against_oracle(eng, seed):
    """The boundary flavour (est2genome, protein2genome; tests/sdp_cases.py) against the oracle (pinned on the reference's SDP)."""
    from sdp_cases import boundary_fuzz_cases
    for cs in boundary_fuzz_cases(seed):
        model, adv = cs["model"], cs["adv"]
        got = eng.sdp(model, cs["pairs"], cs["hsps"], adv[0], adv[1], cs["dropoff"], cs["threshold"], 4)
        n = 0
        for (q, t), h, alns in zip(cs["pairs"], cs["hsps"], got):
            ub, exp = oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, adv[0], adv[1], cs["dropoff"], True, cs["threshold"], 4)
            assert ub == 1
            assert [a.as_dict() for a in alns] == exp, (seed, cs["kind"], cs["variant"], cs["dropoff"], cs["threshold"], len(q), len(t), len(h))
            n += len(exp)
        assert n >= 3


def test_sdp_batches_on_a_side_context_beside_alignments_on_the_main_one(eng):
    """What the drop-in's SDP seam does with a flush cut in the middle of a run (integration/c4gpu_sdp.c): the batch on a second
    context with a stream of its own and a record arena taken ahead and kept (c4gpu_ctx_own_stream, c4gpu_ctx_sdp_reserve), on a
    second host thread, while the first goes on aligning on the main context.  Both give what they give alone."""
    import threading
    from exonerate_amd import workloads
    name = BOUNDARY[0]
    model, par, recs, adv = sdp_case(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    hsps = [r["hsps"] for r in recs]
    side = ex.Engine(0)
    try:
        side.own_stream()
        side.sdp_reserve(1 << 30)
        e2g = ex.Model("est2genome")
        vp = workloads.est2genome_pairs(16, 600, 40000, seed=5)
        alone = [a.vulgar("q", "t") for a in eng.find_path(e2g, vp, dpmemory=32)]
        got, errs = {}, []

        def flights():
            try:
                for k in range(6):
                    got[k] = side.sdp(model, pairs, hsps, adv[0], adv[1], par["dropoff"], par["threshold"], 4)
            except Exception as e:                      # (reported by the assertion below)
                errs.append(e)
        th = threading.Thread(target=flights)
        th.start()
        beside = [[a.vulgar("q", "t") for a in eng.find_path(e2g, vp, dpmemory=32)] for _ in range(4)]
        th.join()
        assert not errs, errs
        assert all(b == alone for b in beside)
        for k in range(6):
            for r, alns in zip(recs, got[k]):
                g = [{"score": a.score, "region": list(a.region), "ops": [list(o) for o in a.ops], "vulgar": a.vulgar(r["id"])}
                     for a in alns]
                assert g == expected(r), (k, r["id"])
        side.sdp_reserve(0)
    finally:
        side.close()
