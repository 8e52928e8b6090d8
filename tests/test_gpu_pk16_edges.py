"""The packed 16-bit route at the edges of its guard (Engine::pk16_fits, c4_engine_launch.inc): the host may only take the packed
score pass / region windows / checkpoint pass where every score a path can reach fits 16 000 and the intron length test
cannot fail on the upper side.  Each case puts a WINDOWED launch (dumps every 256 columns, so that targets of a few thousand
columns take the two-pass route) just inside and just outside one term of the guard, asserts from C4GPU_TRACE which score
kernel ran, and compares EVERY pair with the oracle (Optimal_find_path, optimal.c:368-413):
  * --intronpenalty -1 / -5: an intron's two sites can outweigh its opening, so a path gains per intron the target has room for;
  * the longest query the guard lets through (3 199 nt under the default parameters), and one row more;
  * --maxintron against T + 4: the packed length counter saturates and cannot see "too long";
  * a substitution score of 16: (Q + 1) x 16 (+ the introns' term) <= 16 000 flips just under 1 000 rows;
and a forced disagreement between a window and the score pass (C4GPU_FORCE_CORNER_MISMATCH) hands exactly those pairs to the
one-pass 32-bit kernel instead of failing the batch."""
import random

import pytest

import exonerate_amd as ex
from exonerate_amd import _abi
import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def _rand(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def _mutate(rng, s, rate):
    out = []
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            out.append(rng.choice("ACGT"))
        elif r < 2 * rate / 3:
            out.append(ch + rng.choice("ACGT"))
        elif r >= rate:
            out.append(ch)
    return "".join(out)


def _gene(rng, q, tlen, introns=2):
    """q cut into introns + 1 exons with GT..AG introns, inside a target of about tlen columns."""
    cuts = sorted(rng.sample(range(20, len(q) - 20), introns))
    room = max(200, (tlen - len(q)) // (introns + 2))
    parts, last = [], 0
    for c in cuts + [len(q)]:
        parts.append(_mutate(rng, q[last:c], 0.03))
        if c != len(q):
            parts.append("GT" + _rand(rng, rng.randrange(60, room)) + "AG")
        last = c
    gene = "".join(parts)
    flank = max(0, tlen - len(gene))
    left = rng.randrange(0, flank + 1)
    return _rand(rng, left) + gene + _rand(rng, flank - left)


def _batch(rng, sizes, introns=2):
    pairs = []
    for ql, tl in sizes:
        q = _rand(rng, ql)
        pairs.append((q, _gene(rng, q, tl, introns)))
    return pairs


def _run(eng, model, pairs, monkeypatch, capfd, want_packed, dpmemory=32):
    monkeypatch.setenv("C4GPU_TRACE", "1")
    monkeypatch.setenv("C4GPU_SEED_KSHIFT", "8")
    capfd.readouterr()
    got = eng.find_path(model, pairs, dpmemory=dpmemory, threshold=20)
    err = capfd.readouterr().err
    assert "seeded pass 1" in err, "the call did not take the windowed route: " + err[-800:]
    first = [l for l in err.splitlines() if "seeded pass 1 with kernel" in l][0]
    assert ("kpk16" in first) == want_packed, first
    for k, ((q, t), a) in enumerate(zip(pairs, got)):
        exp = oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dpmemory, threshold=20)
        assert (a.as_dict() if a else None) == exp, k
    return err


@pytest.mark.parametrize("penalty", [-1, -5])
def test_small_intron_penalties_at_the_edge_of_the_gain_bound(eng, monkeypatch, capfd, penalty):
    """(Q + 1) x 5 + (T / min_intron + 1) x (best 5' + best 3' + penalty) <= 16 000: targets just short enough and just too
    long for it, every pair of both launches against the oracle."""
    params = ex.default_params()
    params.intron_open_penalty = penalty
    model = ex.Model("est2genome", params=params)
    rng = random.Random(100 - penalty)
    lib = _abi.load()
    q_len = 600
    fits = lambda t: lib.c4gpu_packed_route_fits(model.c, model.params, q_len, t)
    assert fits(1000) == 1 and fits(90000) == 0
    lo, hi = 1000, 90000                                      # the last target length the guard lets through for this query
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if fits(mid) == 1 else (lo, mid)
    t_fit = lo
    assert 2500 < t_fit < 20000, t_fit
    inside = _batch(rng, [(q_len, t_fit), (q_len - 37, t_fit - 500), (300, 2500), (q_len, t_fit - 3)], introns=3)
    inside = [(q, t[:t_fit]) for q, t in inside]
    _run(eng, model, inside, monkeypatch, capfd, want_packed=True)
    outside = [(inside[0][0], inside[0][1] + _rand(rng, 31))] + inside[1:3]      # one intron's room more
    assert fits(len(outside[0][1])) == 0
    _run(eng, model, outside, monkeypatch, capfd, want_packed=False)


def test_longest_query_the_guard_lets_through(eng, monkeypatch, capfd):
    """(Q + 1) x 5 + what introns can gain <= 16 000: 3 199 rows under the default parameters (an intron's two sites never outweigh its opening there: 13 + 16 - 30).  The longest query that fits runs thirteen strips of 256 rows on the packed
    kernels (HBM carry rows between super-strips), one row more the 32-bit kernels."""
    model = ex.Model("est2genome")
    lib = _abi.load()
    T = 2600
    fits = lambda q: lib.c4gpu_packed_route_fits(model.c, model.params, q, T)
    lo, hi = 1000, 4000
    assert fits(lo) == 1 and fits(hi) == 0
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if fits(mid) == 1 else (lo, mid)
    assert 3000 < lo < 3200, lo
    rng = random.Random(3199)

    def pair(qlen, tlen):          # a target shorter than its query: it holds two exons of the query's first 1 800 nt
        q = _rand(rng, qlen)
        t = _rand(rng, 150) + _mutate(rng, q[:900], 0.03) + "GT" + _rand(rng, 300) + "AG" + _mutate(rng, q[900:1800], 0.03)
        return q, (t + _rand(rng, tlen))[:tlen]

    packed = [pair(lo, T), pair(2900, T - 300)]
    _run(eng, model, packed, monkeypatch, capfd, want_packed=True)
    _run(eng, model, [pair(lo + 1, T), pair(2900, T - 300)], monkeypatch, capfd, want_packed=False)


def test_max_intron_against_the_target_length(eng, monkeypatch, capfd):
    """T + 4 <= --maxintron: the packed length counter saturates, so a launch with a target in which an intron COULD exceed the
    limit keeps the 32-bit kernels; the targets hold an intron longer than the limit, which the reference rejects."""
    rng = random.Random(77)
    sizes = [(500, 4000), (420, 3600), (510, 3996)]
    pairs = _batch(rng, sizes, introns=1)
    tmax = max(len(t) for _, t in pairs)
    for limit, packed in ((tmax + 4, True), (tmax + 3, False), (900, False)):
        params = ex.default_params()
        params.max_intron = limit
        model = ex.Model("est2genome", params=params)
        _run(eng, model, pairs, monkeypatch, capfd, want_packed=packed)


def test_substitution_score_of_16(eng, monkeypatch, capfd):
    """A match score of 16 (instead of 5): (Q + 1) x 16 + what introns can gain <= 16 000 flips near 985 rows."""
    params = ex.default_params()
    for i in range(24):
        for j in range(24):
            if params.dna_submat[i][j] == 5:
                params.dna_submat[i][j] = 16
    model = ex.Model("est2genome", params=params)
    lib = _abi.load()
    T = 3300
    fits = lambda q: lib.c4gpu_packed_route_fits(model.c, model.params, q, T)
    lo, hi = 500, 1100
    assert fits(lo) == 1 and fits(hi) == 0
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if fits(mid) == 1 else (lo, mid)
    assert 900 < lo <= 999, lo
    rng = random.Random(16)
    _run(eng, model, _batch(rng, [(lo, T), (lo - 40, 3000), (640, 2800)]), monkeypatch, capfd, want_packed=True)
    _run(eng, model, _batch(rng, [(lo + 1, T), (lo - 40, 3000), (640, 2800)]), monkeypatch, capfd, want_packed=False)


def test_a_disagreeing_window_demotes_its_pair_not_the_batch(eng, monkeypatch, capfd):
    """Should a packed window's corner ever differ from the score pass, that pair goes to the one-pass 32-bit kernel and the
    others keep their results (C4GPU_FORCE_CORNER_MISMATCH=2: every second pair is treated as such a case); C4GPU_STRICT=1
    keeps the failure for debugging."""
    model = ex.Model("est2genome")
    rng = random.Random(5)
    pairs = _batch(rng, [(700, 4000), (512, 3000), (300, 2500), (900, 5000), (650, 3500)])
    monkeypatch.setenv("C4GPU_FORCE_CORNER_MISMATCH", "2")
    err = _run(eng, model, pairs, monkeypatch, capfd, want_packed=True)
    # (four of the five pairs take the windowed route -- the 300-row query is below its domain --, every second of them is "wrong")
    assert err.count(": one-pass kernel") == 2, err[-1500:]
    assert "2 of 4 paths left to the one-pass kernel" in err
    monkeypatch.setenv("C4GPU_STRICT", "1")
    with pytest.raises(ex.C4GpuError):
        eng.find_path(model, pairs, dpmemory=32, threshold=20)


def test_six_rows_per_lane_for_queries_of_1024_to_1535_rows(eng, monkeypatch, capfd):
    """cDNAs longer than 1 023 nt do not fit the four strips of 256 rows the staged packed score pass holds per workgroup; up to
    1 535 nt they fit four strips of 384 rows (kpk16h: six rows per lane, two waves per SIMD) instead of taking a second pass over
    the target on the form that loads per step.  Ragged batch around both edges (1 024 and 1 535 rows), N in a target, every pair
    against the oracle; a query of 1 536 nt and more runs super-strips of that form one after the other (kpk16j); C4GPU_PK16_R6=0 /
    C4GPU_PK16_LONG=0 give the same alignments on the per-step form."""
    model = ex.Model("est2genome")
    rng = random.Random(1535)
    pairs = _batch(rng, [(1024, 3000), (1100, 2600), (1535, 2800), (700, 2500), (1300, 6000)])
    q, t = pairs[4]
    pairs[4] = (q, t[:900] + "NNNNN" + t[905:])
    err = _run(eng, model, pairs, monkeypatch, capfd, want_packed=True)
    assert "kpk16h_est2genome" in err and "kpk16h_est2genome: 2 workgroups per CU" in err, err[:1500]
    got = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16_R6", "0")
    capfd.readouterr()
    assert got == [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=20)]
    assert "kpk16h" not in capfd.readouterr().err
    monkeypatch.delenv("C4GPU_PK16_R6")
    # one row more: two super-strips of 1 536 rows, the row between them through memory (kpk16j); 2 000 rows likewise;
    # C4GPU_PK16_LONG=0: the form that loads per step
    longer = _batch(rng, [(1536, 2600), (1100, 2600), (2000, 2400)])
    err = _run(eng, model, longer, monkeypatch, capfd, want_packed=True)
    assert "kpk16h" not in err and "kpk16j_est2genome" in err
    got_long = [a.as_dict() if a else None for a in eng.find_path(model, longer, dpmemory=32, threshold=20)]
    monkeypatch.setenv("C4GPU_PK16_LONG", "0")
    capfd.readouterr()
    assert got_long == [a.as_dict() if a else None for a in eng.find_path(model, longer, dpmemory=32, threshold=20)]
    err = capfd.readouterr().err
    assert "kpk16j" not in err and "kpk16d_est2genome" in err
    monkeypatch.delenv("C4GPU_PK16_LONG")
    # at the default dump interval: 1.2 kb against 40 kb, both forms
    monkeypatch.delenv("C4GPU_SEED_KSHIFT")
    big = _batch(rng, [(1200, 40000), (1400, 36000), (1024, 33000)], introns=4)
    capfd.readouterr()
    a = [x.as_dict() if x else None for x in eng.find_path(model, big, dpmemory=32, threshold=20)]
    assert "kpk16h_est2genome" in capfd.readouterr().err
    monkeypatch.setenv("C4GPU_PK16", "0")
    assert a == [x.as_dict() if x else None for x in eng.find_path(model, big, dpmemory=32, threshold=20)]
