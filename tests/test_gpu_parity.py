"""Parity of the HIP engine (through the C ABI of libc4gpu.so) with the reference.

Golden vectors = outputs of the reference itself (tests/golden/*.jsonl, tools/make_golden.py); the oracle
(oracle/c4_oracle.c, pinned to the same vectors in test_oracle_golden.py) is the checker for seeded inputs
that have no golden record.  Integer work: every comparison is bit-exact.
"""
import random
import pytest

import exonerate_amd as ex
from exonerate_amd import _abi
import oracle_lib
from golden_util import SETS, SUBOPT_SETS, DERIVED_SETS, SPAN_SETS, ANNOT_SETS, load_set, expected, set_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def _model(name):
    if name in DERIVED_SETS:
        mt, qa, ta, (src, dst, ss, es) = DERIVED_SETS[name]
        return ex.Model.derived(mt, src, dst, ss, es, qa, ta)
    mt, qa, ta = SETS[name] if name in SETS else ANNOT_SETS[name] if name in ANNOT_SETS else SUBOPT_SETS[name]
    return ex.Model(mt, qa, ta, params=set_params(_abi.load(), name))


@pytest.mark.parametrize("name", sorted(SETS) + sorted(DERIVED_SETS))
def test_find_score_and_path_match_reference_vectors(eng, name):
    model = _model(name)
    recs = load_set(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    scores = eng.find_score(model, pairs)
    assert scores == [r["score"] for r in recs]
    alns = eng.find_path(model, pairs, dpmemory=recs[0]["dpmemory"])
    for rec, aln in zip(recs, alns):
        if "path_score" not in rec:
            assert aln is None, rec["id"]
            continue
        assert aln is not None, rec["id"]
        assert aln.as_dict(rec["id"]) == expected(rec), rec["id"]


@pytest.mark.parametrize("name", sorted(ANNOT_SETS))
def test_annotated_queries_match_reference_vectors(eng, name, monkeypatch, capfd):
    """exonerate's --annotation (match.c:276-281): a DNA query's positions inside its annotated CDS cannot take part in a 1:1 DNA
    match.  c4gpu_batch_set_annotation gives those positions a matrix row of their own (MATCH_IMPOSSIBLY_LOW_SCORE against
    everything) and sends the batch's passes to the kernels that keep every validity mask; scores, regions, operations and printed
    lines are the reference's (vectors made with the annotation attached to the query), on both memory routes; the same batch
    without the annotation gives other alignments, and taking the annotation away again restores them."""
    model = _model(name)
    recs = load_set(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    dpm = recs[0]["dpmemory"]
    monkeypatch.setenv("C4GPU_TRACE", "1")
    b = ex.ResidentBatch(eng, model, pairs)

    def results():
        b.run(2, dpmemory=dpm)
        return [b.alignment(i) for i in range(len(recs))]
    plain = [a.as_dict(r["id"]) if a else None for a, r in zip(results(), recs)]
    capfd.readouterr()
    b.set_annotation([tuple(r["cds"]) for r in recs])
    got = results()
    err = capfd.readouterr().err
    launched = [l for l in err.splitlines() if "c4gpu trace:   kernel " in l]
    assert launched and not any("_local" in l or "pk16" in l or "16_" in l for l in launched), launched     # every mask kept, 32-bit passes
    differ = 0
    for rec, aln, pl in zip(recs, got, plain):
        if "path_score" not in rec:
            assert aln is None, rec["id"]
        else:
            assert aln is not None and aln.as_dict(rec["id"]) == expected(rec), rec["id"]
        differ += (aln.as_dict(rec["id"]) if aln else None) != pl
    assert differ >= len(recs) // 3
    b.set_annotation(None)
    assert [a.as_dict(r["id"]) if a else None for a, r in zip(results(), recs)] == plain
    b.close()


def test_reference_known_answer_tests_on_gpu(eng):
    """src/model/affine.test.c:107-110, est2genome.test.c:63, with the printed local vulgar."""
    for name, score in (("affine_global_protein", -151), ("affine_bestfit_protein", 18),
                        ("affine_local_protein", 32), ("affine_overlap_protein", 18)):
        rec = [r for r in load_set(name) if r["id"] == "kat_affine"][0]
        assert eng.find_score(_model(name), [(rec["query"], rec["target"])]) == [score]
    rec = [r for r in load_set("est2genome") if r["id"] == "kat_est2genome"][0]
    assert eng.find_score(_model("est2genome"), [(rec["query"], rec["target"])]) == [157]
    # src/model/protein2dna.test.c:34 and protein2genome.test.c:34
    for name, rid, score in (("protein2dna", "kat_protein2dna", 134), ("protein2genome", "kat_protein2genome", 125)):
        rec = [r for r in load_set(name) if r["id"] == rid][0]
        assert eng.find_score(_model(name), [(rec["query"], rec["target"])]) == [score]
        assert eng.find_path(_model(name), [(rec["query"], rec["target"])])[0].as_dict(rid) == expected(rec)
    rec = [r for r in load_set("affine_local_protein") if r["id"] == "kat_affine"][0]
    aln = eng.find_path(_model("affine_local_protein"), [(rec["query"], rec["target"])])[0]
    assert aln.vulgar("kat_affine").endswith("32 M 8 8 G 1 0 M 4 4") and aln.region == (11, 33, 13, 12)


def test_splice_arrays_match_reference(eng):
    model = _model("est2genome")
    keys = ["ss5_forward", "ss3_forward", "ss3_reverse", "ss5_reverse"]       # C4GPU_SS* order
    for rec in load_set("est2genome")[:10]:
        got = eng.splice_predict(model.params, rec["target"])
        for k, key in enumerate(keys):
            assert got[k] == rec[key], (rec["id"], key)


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


@pytest.mark.parametrize("model_type,qlen,tlen,dpm", [
    ("affine:local", 300, 330, 32), ("affine:local", 700, 900, 32), ("affine:global", 520, 500, 32),
    ("affine:bestfit", 130, 900, 32), ("affine:overlap", 260, 300, 1),
    ("est2genome", 257, 3000, 32), ("est2genome", 600, 9000, 32), ("est2genome", 300, 2500, 0),
])
def test_seeded_pairs_match_oracle(eng, model_type, qlen, tlen, dpm):
    """Sizes that cross the 64*R strip boundary and take the region -> checkpoint -> sub-alignment route."""
    rng = random.Random(hash((model_type, qlen, tlen)) & 0xffff)
    model = ex.Model(model_type)
    pairs = []
    for k in range(6):
        q = _rand(rng, qlen + k)
        if model_type == "est2genome":
            cut = sorted(rng.sample(range(20, qlen - 20), 2))
            t = (_rand(rng, 100) + _mutate(rng, q[:cut[0]], 0.03) + "GT" + _rand(rng, tlen // 4) + "AG" +
                 _mutate(rng, q[cut[0]:cut[1]], 0.03) + "GT" + _rand(rng, tlen // 3) + "AG" +
                 _mutate(rng, q[cut[1]:], 0.03) + _rand(rng, 150))
        else:
            t = _rand(rng, 40) + _mutate(rng, q, 0.1) + _rand(rng, max(0, tlen - qlen))
        pairs.append((q, t))
    scores = eng.find_score(model, pairs)
    alns = eng.find_path(model, pairs, dpmemory=dpm)
    for (q, t), s, a in zip(pairs, scores, alns):
        assert s == oracle_lib.find_score(model.c, model.params, q.encode(), t.encode())
        exp = oracle_lib.find_path(model.c, model.params, q.encode(), t.encode(), dpmemory=dpm)
        assert a.as_dict() == exp


@pytest.mark.parametrize("model_type,qlen,tlen", [("est2genome", 1300, 5000), ("affine:local", 2100, 2300),
                                                   ("affine:global", 1500, 1400)])
def test_long_queries_cross_several_super_strips(eng, model_type, qlen, tlen):
    """Queries longer than one workgroup's 4 x 64 x R rows: the cooperating-wave kernels hand rows from one
    super-strip to the next through the HBM carry rows (and the one-wave kernels through theirs)."""
    rng = random.Random(qlen)
    model = ex.Model(model_type)
    q = _rand(rng, qlen)
    if model_type == "est2genome":
        c = qlen // 3
        t = _rand(rng, 200) + _mutate(rng, q[:c], 0.03) + "GT" + _rand(rng, tlen // 3) + "AG" + _mutate(rng, q[c:], 0.03) + _rand(rng, 300)
    else:
        t = _rand(rng, 60) + _mutate(rng, q, 0.08) + _rand(rng, max(0, tlen - qlen))
    pairs = [(q, t), (q[:qlen - 37], t)]
    scores = eng.find_score(model, pairs)
    alns = eng.find_path(model, pairs, dpmemory=32)
    for (qq, tt), s_, a in zip(pairs, scores, alns):
        assert s_ == oracle_lib.find_score(model.c, model.params, qq.encode(), tt.encode())
        assert a.as_dict() == oracle_lib.find_path(model.c, model.params, qq.encode(), tt.encode(), dpmemory=32)


def test_raw_viterbi_modes_match_oracle(eng):
    """Viterbi_DP_Func level: region + checkpoint (continuation) passes against oracle_viterbi."""
    import ctypes as C
    olib = oracle_lib.load()
    model = ex.Model("est2genome")
    rng = random.Random(5)
    q = _rand(rng, 150)
    t = _rand(rng, 60) + q[:70] + "GT" + _rand(rng, 300) + "AG" + q[70:] + _rand(rng, 80)
    region = (0, 0, len(q), len(t))
    got = eng.viterbi(model, ex.MODE_FIND_REGION, [(q, t)], [{"pair": 0, "region": region}])[0]
    vo = oracle_lib.ViterbiOut()
    olib.oracle_viterbi(model.c, model.params, ex.MODE_FIND_REGION, q.encode(), len(q), t.encode(), len(t),
                        _abi.Region(*region), None, 0, vo)
    assert (got["score"], got["query_start"], got["target_start"], got["query_end"], got["target_end"]) == \
           (vo.score, vo.query_start, vo.target_start, vo.query_end, vo.target_end)
    olib.oracle_viterbi_out_clear(vo)
    # checkpoint pass over the aligned region, from START to END
    ar = (got["query_start"], got["target_start"], got["query_end"] - got["query_start"],
          got["target_end"] - got["target_start"])
    cont = _abi.Continuation()
    cont.first_state, cont.final_state = model.c.start_state, model.c.end_state
    vo = oracle_lib.ViterbiOut()
    olib.oracle_viterbi(model.c, model.params, ex.MODE_FIND_CHECKPOINTS, q.encode(), len(q), t.encode(), len(t),
                        _abi.Region(*ar), cont, 5, vo)
    got = eng.viterbi(model, ex.MODE_FIND_CHECKPOINTS, [(q, t)],
                      [{"pair": 0, "region": ar, "checkpoints": 5,
                        "continuation": {"first_state": model.c.start_state, "final_state": model.c.end_state}}])[0]
    assert got["score"] == vo.score and got["last_srp"] == vo.last_srp
    # slot 1 (intron shadow) of a non-intron state is never read again: the engine reports 0 there
    assert got["final_cell"][0] == vo.final_cell[0] and got["final_cell"][vo.cell_size - 1] == vo.final_cell[vo.cell_size - 1]
    olib.oracle_viterbi_out_clear(vo)


@pytest.mark.parametrize("name", sorted(SUBOPT_SETS))
def test_suboptimal_loop_matches_reference_vectors(eng, name):
    """GAM_Result_exhaustive_create's loop with SubOpt blocking on the device (viterbi.c:701-704) against
    the successive alignments the reference itself produced — all pairs of the set in one batch per round."""
    model = _model(name)
    recs = load_set(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    found = eng.find_all_paths(model, pairs, dpmemory=recs[0]["dpmemory"], threshold=recs[0]["threshold"],
                               max_paths=3 if "global" in name else 6)
    for rec, alns in zip(recs, found):
        assert len(alns) == len(rec["subopt"]), rec["id"]
        for a, exp in zip(alns, rec["subopt"]):
            assert (a.score, list(a.region), [list(o) for o in a.ops], a.vulgar(rec["id"])) == \
                   (exp["path_score"], exp["region"], exp["ops"], exp["vulgar"]), rec["id"]


def test_resident_batch_suboptimal_loop(eng):
    """c4gpu_batch_next_paths == the per-call loop == the reference vectors."""
    recs = load_set("est2genome_subopt")
    model = ex.Model("est2genome")
    batch = ex.ResidentBatch(eng, model, [(r["query"], r["target"]) for r in recs])
    thr = recs[0]["threshold"]
    batch.run(2, 32, thr)
    rounds = [[batch.alignment(i) for i in range(len(recs))]]
    while len(rounds) < 6 and batch.next_paths(32, thr) > 0:
        rounds.append([batch.alignment(i) for i in range(len(recs))])
    batch.close()
    for i, rec in enumerate(recs):
        got = [r[i] for r in rounds if r[i] is not None]
        assert [(a.score, a.vulgar(rec["id"])) for a in got] == \
               [(e["path_score"], e["vulgar"]) for e in rec["subopt"]][:len(got)], rec["id"]
        assert len(got) == min(len(rec["subopt"]), 6)


def test_seeded_suboptimal_pairs_match_oracle(eng):
    """Larger seeded repeats (several 64*R strips, reduced-space route at -D 1) against the oracle's loop."""
    rng = random.Random(77)
    for model_type, qlen, dpm in (("affine:local", 400, 1), ("est2genome", 300, 1), ("est2genome", 520, 32)):
        model = ex.Model(model_type)
        pairs = []
        for k in range(3):
            q = _rand(rng, qlen + 7 * k)
            if model_type == "est2genome":
                c = qlen // 2
                gene = _mutate(rng, q[:c], 0.03) + "GT" + _rand(rng, 400) + "AG" + _mutate(rng, q[c:], 0.03)
                t = _rand(rng, 80) + gene + _rand(rng, 200) + _mutate(rng, gene, 0.05) + _rand(rng, 60)
            else:
                t = _rand(rng, 30) + _mutate(rng, q, 0.05) + _rand(rng, 90) + _mutate(rng, q, 0.15) + _rand(rng, 20)
            pairs.append((q, t))
        found = eng.find_all_paths(model, pairs, dpmemory=dpm, threshold=100, max_paths=4)
        for (q, t), alns in zip(pairs, found):
            exp = oracle_lib.find_paths_subopt(model.c, model.params, q.encode(), t.encode(), dpm, 100, 4)
            assert [a.as_dict() for a in alns] == [d for d, _ in exp]
            assert len(alns) >= 2


def test_raw_viterbi_with_blocked_cells_matches_oracle(eng):
    """Viterbi_DP_Func level with a soi: the region pass of the second alignment."""
    olib = oracle_lib.load()
    model = ex.Model("affine:local")
    rng = random.Random(11)
    q = _rand(rng, 300)
    t = _rand(rng, 20) + _mutate(rng, q, 0.05) + _rand(rng, 50) + _mutate(rng, q, 0.1)
    first = eng.find_path(model, [(q, t)])[0]
    so = ex.SubOpt(len(q), len(t))
    so.add_alignment(first)
    oso = olib.oracle_subopt_create(len(q), len(t))
    olib.oracle_subopt_add_alignment(oso, model.c, first._c())
    assert so.points() == oracle_lib.subopt_points(oso)
    region = (0, 0, len(q), len(t))
    got = eng.viterbi(model, ex.MODE_FIND_REGION, [(q, t)], [{"pair": 0, "region": region, "subopt": so}])[0]
    vo = oracle_lib.ViterbiOut()
    olib.oracle_viterbi_subopt(model.c, model.params, ex.MODE_FIND_REGION, q.encode(), len(q), t.encode(), len(t),
                               _abi.Region(*region), None, 0, oso, vo)
    assert (got["score"], got["query_start"], got["target_start"], got["query_end"], got["target_end"]) == \
           (vo.score, vo.query_start, vo.target_start, vo.query_end, vo.target_end)
    assert got["score"] < first.score
    olib.oracle_viterbi_out_clear(vo)
    olib.oracle_subopt_destroy(oso)
    so.close()


def test_residue_outside_alphabet_is_rejected(eng):
    model = ex.Model("affine:local")
    with pytest.raises(ex.C4GpuError):
        eng.find_score(model, [("ACGT", "AC#T")])


def test_full_size_properties(eng):
    """BASELINE-size pair (1 kb x 100 kb est2genome): size-independent properties.
    (a) the alignment of a cDNA to genomic with planted introns recovers every exon block;
    (b) score of (q, t) is unchanged by appending unrelated flank to the target (local model);
    (c) region pass and score pass agree on the score."""
    rng = random.Random(11)
    model = ex.Model("est2genome")
    q = _rand(rng, 1000)
    cuts = [0, 230, 520, 790, 1000]
    t = _rand(rng, 30000)
    for a, b in zip(cuts[:-1], cuts[1:]):
        t += q[a:b]
        if b != 1000:
            t += "GT" + _rand(rng, 4000) + "AG"
    core = t
    t = core + _rand(rng, 100000 - len(core))
    s_full, s_core = eng.find_score(model, [(q, t), (q, core)])
    assert s_full == s_core
    aln = eng.find_path(model, [(q, t)])[0]
    assert aln.score == s_full
    v = aln.vulgar().split()[10:]
    m_blocks = [int(v[i + 1]) for i in range(0, len(v), 3) if v[i] == "M"]
    assert m_blocks == [230, 290, 270, 210]
    introns = [int(v[i + 2]) for i in range(0, len(v), 3) if v[i] == "I"]
    assert introns == [4000, 4000, 4000]


def test_sequential_fallback_path_is_exact(monkeypatch):
    """The strictly sequential reduced-space route (used when the batched prediction of continuation cells
    fails its check) gives the reference's alignments too.  Forced through C4GPU_FORCE_SEQUENTIAL in a
    fresh process (the switch is read once)."""
    import subprocess, sys, os, json
    code = r'''
import sys, json
sys.path.insert(0, "tests")
import exonerate_amd as ex
from golden_util import load_set, expected
eng = ex.Engine(0)
for name, mt in (("est2genome_D0", "est2genome"), ("affine_local_dna_D0", "affine:local")):
    recs = load_set(name)
    alns = eng.find_path(ex.Model(mt), [(r["query"], r["target"]) for r in recs], dpmemory=0)
    for r, a in zip(recs, alns):
        assert a.as_dict(r["id"]) == expected(r), r["id"]
print("OK")
'''
    env = dict(os.environ, C4GPU_FORCE_SEQUENTIAL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0 and b"OK" in out.stdout, out.stderr.decode()[-2000:]


def test_thousands_of_bsdp_sized_jobs_on_derived_models(eng):
    """K5: BSDP's unit of work — rectangles of at most 49 x 49 cells of ONE pair under a derived model
    (join: CORNER to CORNER between two HSPs; terminal: open scope on one side) — as one launch of thousands
    of jobs (Viterbi_DP_Func level), every score and every path checked against the oracle."""
    import ctypes as C
    import time
    olib = oracle_lib.load()
    rng = random.Random(2024)
    q = _rand(rng, 1500)
    t = _mutate(rng, q, 0.12) + _rand(rng, 300)
    for mt, spec in (("affine:local", (2, 2, 4, 4)), ("affine:local", (0, 2, 0, 4)), ("est2genome", (2, 2, 4, 4)),
                     ("est2genome", (5, 1, 4, 0))):
        model = ex.Model.derived(mt, *spec)
        jobs = []
        for k in range(3000):
            ql, tl = rng.randint(1, 49), rng.randint(2, 49)
            qs, ts = rng.randint(0, len(q) - ql), rng.randint(0, len(t) - tl)
            jobs.append({"pair": 0, "region": (qs, ts, ql, tl)})
        t0 = time.perf_counter()
        got = eng.viterbi(model, ex.MODE_FIND_PATH, [(q, t)], jobs)
        dt = time.perf_counter() - t0
        scores = eng.viterbi(model, ex.MODE_FIND_SCORE, [(q, t)], jobs)
        print("%s %s: %d path jobs in %.1f ms" % (mt, spec, len(jobs), dt * 1e3))
        for k in range(0, len(jobs), 7):
            vo = oracle_lib.ViterbiOut()
            olib.oracle_viterbi(model.c, model.params, ex.MODE_FIND_PATH, q.encode(), len(q), t.encode(), len(t),
                                _abi.Region(*jobs[k]["region"]), None, 0, vo)
            g = got[k]
            assert (g["score"], g["query_start"], g["target_start"], g["query_end"], g["target_end"]) == \
                   (vo.score, vo.query_start, vo.target_start, vo.query_end, vo.target_end), (mt, spec, jobs[k])
            assert g["ops"] == [vo.ops[x] for x in range(vo.n_ops)], (mt, spec, jobs[k])
            assert scores[k]["score"] == vo.score
            olib.oracle_viterbi_out_clear(vo)


def test_score_pass_first_gives_the_same_alignments(eng, monkeypatch):
    """find_path_batch may put a FIND_SCORE pass in front of the region pass when few pairs reach the
    threshold (all-vs-all runs): forced on, forced off and adaptive give identical results."""
    from exonerate_amd import workloads
    base = workloads.est2genome_pairs(24, 300, 40000)
    pairs = []
    for k in range(600):                      # 1 in 25 pairs is a true cDNA/gene pair, the rest are unrelated
        q, t = base[k % 24]
        pairs.append((q, t) if k % 25 == 0 else (q, base[(k + 7) % 24][1]))
    model = ex.Model("est2genome")
    runs = {}
    for mode in ("0", "1", None, None):
        if mode is None:
            monkeypatch.delenv("C4GPU_SCORE_FIRST", raising=False)
        else:
            monkeypatch.setenv("C4GPU_SCORE_FIRST", mode)
        alns = eng.find_path(model, pairs, dpmemory=1, threshold=400)
        runs.setdefault(str(mode), []).append([a.as_dict() if a else None for a in alns])
    assert runs["0"][0] == runs["1"][0] == runs["None"][0] == runs["None"][1]
    hits = [a for a in runs["0"][0] if a]
    assert 20 <= len(hits) <= 40
    q, t = pairs[0]
    assert runs["0"][0][0] == oracle_lib.find_path(model.c, model.params, q, t, dpmemory=1, threshold=400)


def test_stale_sub_alignment_tails_are_recomputed(eng, monkeypatch, capfd):
    """With intron length limits a sub-DP seeded at its corner can end above the cell the checkpoint pass
    predicted (no optimal substructure); the sub-alignments after it are then recomputed in the reference's
    order.  Pair 214 of the north-star batch does this in its second sub-optimal round (1 kb x 100 kb, checked
    against the oracle's loop at full size, tests/golden/stale_tails_pair214.json)."""
    from exonerate_amd import workloads
    model = ex.Model("est2genome")
    q, t = workloads.est2genome_pairs(1, 1000, 100000, first=214)[0]
    monkeypatch.setenv("C4GPU_TRACE", "1")
    # C4GPU_CELL_STRICT=1: a final cell whose SCORE differs from the prediction counts as a miss (the form of rounds 1-4) and the
    # tail is recomputed; the default compares the shadow slots only (a score offset moves no decision of a continuation sub-DP:
    # c4_engine_staging.inc, final_cell_equiv) and keeps the batch's results -- both must give the oracle's alignments
    monkeypatch.setenv("C4GPU_CELL_STRICT", "1")
    found = eng.find_all_paths(model, [(q, t)], dpmemory=32, threshold=300, max_paths=2)[0]
    err = capfd.readouterr().err
    assert "predicted" in err, "this input no longer exercises the repair route"
    monkeypatch.delenv("C4GPU_CELL_STRICT")
    relaxed = eng.find_all_paths(model, [(q, t)], dpmemory=32, threshold=300, max_paths=2)[0]
    assert [a.as_dict() for a in relaxed] == [a.as_dict() for a in found]
    # the oracle's loop over this pair (45 s of CPU), written down by tools/make_stale_tails_golden.py
    import json, os
    exp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stale_tails_pair214.json")))
    assert (exp["dpmemory"], exp["threshold"], exp["max_paths"]) == (32, 300, 2)
    assert [a.as_dict() for a in found] == exp["alignments"]
    assert len(found) == 2


def test_score_offsets_and_repaired_tails_on_wide_chance_alignments(eng, monkeypatch, capfd):
    """Chance alignments across whole 100 kb windows (8 cDNAs against 8 windows, all against all: 56 of the 64 pairs) have
    ~54 sections each, and a few per hundred end a section a few points above the cell the checkpoint pass predicted (intron
    length limits: no optimal substructure) -- same shadow slots, another score.  Three ways through, one result:
      * default: the shadow slots decide (a score offset moves no decision of a continuation sub-DP, final_cell_equiv): the
        batch's results stand, no repair launch;
      * C4GPU_CELL_STRICT=1: the score counts as a miss, the tail is recomputed, and a recomputed sub-alignment that ends in
        the predicted cell re-joins the batch's own results;
      * C4GPU_CELL_STRICT=1 C4GPU_REPAIR_REJOIN=0 C4GPU_NESTED_REDO=1: every sub-alignment behind a miss recomputed one launch
        at a time / the pair follows the reference call by call (the form of rounds 1-4, pinned on the oracle at full size by
        test_stale_sub_alignment_tails_are_recomputed)."""
    from exonerate_amd import workloads
    model = ex.Model("est2genome")
    base = workloads.est2genome_pairs(8, 1000, 100000, first=56)        # (cDNA 58 against window 63 misses a cell in section 4)
    pairs = [(base[i][0], base[j][1]) for i in range(8) for j in range(8)]
    monkeypatch.setenv("C4GPU_TRACE", "1")
    got, misses, score_only = {}, {}, {}
    for name, env in (("relaxed", {}), ("strict", {"C4GPU_CELL_STRICT": "1"}),
                      ("old", {"C4GPU_CELL_STRICT": "1", "C4GPU_REPAIR_REJOIN": "0", "C4GPU_NESTED_REDO": "1"})):
        for k in ("C4GPU_CELL_STRICT", "C4GPU_REPAIR_REJOIN", "C4GPU_NESTED_REDO"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        capfd.readouterr()
        got[name] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=100)]
        err = capfd.readouterr().err
        misses[name] = err.count(" predicted")
        import re
        score_only[name] = sum(int(x) for x in re.findall(r"(\d+) final cells that differ from their prediction in the score only", err))
    # (the device route counts what the relaxation let through: some on this input by default, none when the score counts)
    assert score_only["relaxed"] >= 1 and score_only["strict"] == 0, score_only
    assert got["relaxed"] == got["strict"] == got["old"]
    assert sum(1 for a in got["relaxed"] if a) == 64
    assert misses["old"] >= 1, "this input no longer has a section that misses its predicted cell"
    assert misses["relaxed"] < misses["old"]


def test_one_row_sections_inside_an_intron_answered_without_a_dp(eng, monkeypatch, capfd):
    """A sub-alignment between two checkpoints that has no query row and starts and ends in one intron state can only be that
    state's loop, T times, when leaving the intron and coming back cannot pay (best 3' site + best 5' site + opening constant
    < 0: 13 + 15 - 30 under the default parameters) -- the path kernel answers it without a DP (KParams::loop_tr).  Chance
    alignments across whole 100 kb windows are full of such sections: the same alignments with and without the shortcut, one
    shorter one against the oracle; with an opening constant of -20 an intron's two sites can pay for it and the shortcut must be off."""
    from exonerate_amd import workloads
    base = workloads.est2genome_pairs(4, 1000, 100000, first=56)
    pairs = [(base[i][0], base[j][1]) for i in range(4) for j in range(4)]
    small = workloads.est2genome_pairs(2, 400, 40000, first=11)
    pairs += [(small[0][0], small[1][1])]          # a chance alignment the oracle runs in 15 s: introns of 2-4 kb across its sections
    monkeypatch.setenv("C4GPU_TRACE", "1")
    for penalty, want_on in ((None, True), (-20, False)):
        params = ex.default_params()
        if penalty is not None:
            params.intron_open_penalty = penalty
        model = ex.Model("est2genome", params=params)
        got = {}
        for off in (False, True):
            if off:
                monkeypatch.setenv("C4GPU_LOOP_SHORTCUT", "0")
            else:
                monkeypatch.delenv("C4GPU_LOOP_SHORTCUT", raising=False)
            capfd.readouterr()
            got[off] = [a.as_dict() if a else None for a in eng.find_path(model, pairs, dpmemory=32, threshold=60)]
            err = capfd.readouterr().err
            lines = [l for l in err.splitlines() if "answered without a DP" in l]
            assert lines, err[-600:]
            assert all((" none" not in l) == (want_on and not off) for l in lines), lines[:3]
        monkeypatch.delenv("C4GPU_LOOP_SHORTCUT", raising=False)
        assert got[False] == got[True]
        assert sum(1 for a in got[False] if a) == len(pairs)
        if penalty is None:
            assert got[False][16] == oracle_lib.find_path(model.c, model.params, pairs[16][0], pairs[16][1], dpmemory=32, threshold=60)
            assert max(op[1] for op in got[False][16]["ops"]) > 3000          # (an intron longer than a section)


@pytest.mark.parametrize("name,mtype,qa,match_state,span_state", SPAN_SETS)
def test_span_seam_matches_reference_vectors(eng, name, mtype, qa, match_state, span_state):
    """BSDP's span models on the device: the src DP copies out every END cell (cell_end_func), the dst DP reads
    its START cells from a matrix (cell_start_func) — viterbi.c:728-741,793-799 — against what the reference
    itself produced (refdump --cmd span), all pairs of the set in one launch per DP."""
    import ctypes as C
    src = ex.Model.derived(mtype, match_state, span_state, 4, 0, qa, 0)
    dst = ex.Model.derived(mtype, span_state, match_state, 0, 4, qa, 0)
    recs = load_set(name)
    pairs = [(r["query"], r["target"]) for r in recs]
    cs = 1 + src.c.total_shadow_designations
    mats, jobs = [], []
    for k, r in enumerate(recs):
        Q, T = len(r["query"]), len(r["target"])
        m = (C.c_int32 * ((Q + 1) * (T + 1) * cs))()
        for x in range((Q + 1) * (T + 1)):
            m[x * cs] = _abi.IMPOSSIBLY_LOW_SCORE
        mats.append(m)
        jobs.append({"pair": k, "region": (0, 0, Q, T), "end_cells": m})
    got = eng.viterbi(src, ex.MODE_FIND_SCORE, pairs, jobs)
    for r, g, m in zip(recs, got, mats):
        Q, T = len(r["query"]), len(r["target"])
        assert g["score"] == r["src_score"], r["id"]
        cells = {}
        for i in range(Q + 1):
            for j in range(T + 1):
                x = (i * (T + 1) + j) * cs
                if m[x] != _abi.IMPOSSIBLY_LOW_SCORE:
                    cells[(i, j)] = [m[x + l] for l in range(cs)]
        assert cells == {(c[0], c[1]): c[2:] for c in r["end_cells"]}, r["id"]
    jobs = [{"pair": k, "region": (0, 0, len(r["query"]), len(r["target"])), "start_cells": mats[k]} for k, r in enumerate(recs)]
    scores = eng.viterbi(dst, ex.MODE_FIND_SCORE, pairs, jobs)
    paths = eng.viterbi(dst, ex.MODE_FIND_PATH, pairs, jobs)
    for r, s_, p in zip(recs, scores, paths):
        assert s_["score"] == r["dst_score"], r["id"]
        if "path_score" in r:
            assert p["score"] == r["path_score"]
            assert [p["query_start"], p["target_start"], p["query_end"] - p["query_start"],
                    p["target_end"] - p["target_start"]] == r["region"], r["id"]
            rle = []
            for o in p["ops"]:
                if rle and rle[-1][0] == o:
                    rle[-1][1] += 1
                else:
                    rle.append([o, 1])
            assert rle == r["ops"], r["id"]


@pytest.mark.parametrize("fill", ["90", "255"])
def test_nothing_depends_on_what_a_new_buffer_held(eng, monkeypatch, fill):
    """C4GPU_FILL_ALLOC=<byte> starts every device buffer of the library as that byte (DevBuf::alloc): the reference's vector
    sets of the four families, the derived models and a reduced-space set come out as with fresh (zeroed) memory -- no kernel
    reads a buffer before something wrote it.  (The companion for registers is the value-initialised per-wave DP object,
    c4_viterbi_kernel.h: round 4 found one derived protein2genome set depending on what the registers held.)"""
    monkeypatch.setenv("C4GPU_FILL_ALLOC", fill)
    for name in ("affine_local_dna", "est2genome", "protein2dna", "protein2genome", "derived_protein2genome_end",
                 "derived_est2genome_fwd_join", "est2genome_D0", "est2genome_big"):
        model = _model(name)
        recs = load_set(name)
        pairs = [(r["query"], r["target"]) for r in recs]
        assert eng.find_score(model, pairs) == [r["score"] for r in recs], name
        for rec, aln in zip(recs, eng.find_path(model, pairs, dpmemory=recs[0]["dpmemory"])):
            if "path_score" in rec:
                assert aln is not None and aln.as_dict(rec["id"]) == expected(rec), (name, rec["id"])
