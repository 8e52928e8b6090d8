"""End-to-end drop-in check: the reference's own `exonerate` binary with integration/c4gpu_shim.c linked in
(integration/_build/exonerate-gpu; every Optimal Viterbi call of accelerated models goes to libc4gpu.so)
prints byte-identical output to the unmodified reference with its compiled CPU Viterbi
(oracle/_ref/exonerate-compiled).  Both binaries are built in the build container and travel with the repo."""
import re
import os, subprocess, random
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_EXE = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
CPU_EXE = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")


def _fasta(path, recs):
    with open(path, "w") as f:
        for name, seq in recs:
            f.write(">%s\n%s\n" % (name, seq))


def _run(exe, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode(), r.stderr.decode()


class _Ref:
    """The reference binary on `args`, started at once on a host core and collected when its output is first needed: the
    drop-in's run of the same command goes beside it instead of after it (the reference is the slower of the two)."""

    def __init__(self, args):
        self.p = subprocess.Popen([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)

    def out(self):
        o, e = self.p.communicate(timeout=900)
        assert self.p.returncode == 0, e.decode()[-2000:]
        return o.decode()


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,extra,batch", [
    ("est2genome", [], "4096"), ("est2genome", ["-D", "1"], "4096"), ("affine:local", [], "4096"),
    ("affine:global", [], "4096"), ("protein2dna", [], "4096"), ("protein2genome", [], "4096"),
    # C4GPU_BATCH=0: every Viterbi call goes to the device on its own (the plain Bootstrapper_lookup shim);
    # 2: several flushes of the batching seam; --bestn: thresholds that move while results are submitted
    ("est2genome", [], "0"), ("affine:local", ["-D", "1"], "0"), ("protein2genome", [], "0"),
    ("est2genome", ["--forcegtag", "yes"], "4096"), ("protein2genome", ["--forcegtag", "yes"], "0"),
    ("est2genome", ["--percent", "60"], "4096"), ("affine:local", ["--percent", "30", "-S", "yes"], "5"),
    ("est2genome", [], "2"), ("affine:local", ["--bestn", "1"], "4"), ("est2genome", ["--bestn", "2", "-S", "yes"], "4096"),
])
def test_exonerate_gpu_output_is_byte_identical(tmp_path, model, extra, batch):
    rng = random.Random(len(model) * 7 + len([e for e in extra if e in ("-D", "1")]))
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    aa = lambda n: "".join(rng.choice("ARNDCQEGHILKMFPSTWYV") for _ in range(n))
    table = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    codon = {}
    for i, a in enumerate("TCAG"):
        for j, b in enumerate("TCAG"):
            for k, c in enumerate("TCAG"):
                codon.setdefault(table[i * 16 + j * 4 + k], []).append(a + b + c)
    qs, ts = [], []
    # (the sub-optimal loop over est2genome pairs is the reference's slowest case here: two sequences a side instead of three)
    for n in range(2 if (model == "est2genome" and "-S" in extra) else 3):
        if model.startswith("protein"):
            q = aa(120 + 30 * n)
            coding = "".join(rng.choice(codon[x]) for x in q)
            if model == "protein2genome":
                c1, c2 = 100 + n, 250 + 2 * n
                coding = coding[:c1] + "GT" + dna(300) + "AG" + coding[c1:c2] + "GT" + dna(500) + "AG" + coding[c2:]
            t = dna(400) + coding + dna(600)
        else:
            q = dna(500 + 100 * n)
            if model == "est2genome" and "-S" in extra:     # (the reference's sub-optimal loop is the slow side of this test)
                t = dna(1000) + q[:200] + "GT" + dna(800) + "AG" + q[200:420] + "GT" + dna(1500) + "AG" + q[420:] + dna(1200)
            elif model == "est2genome":
                t = dna(2000) + q[:200] + "GT" + dna(1500) + "AG" + q[200:420] + "GT" + dna(3000) + "AG" + q[420:] + dna(2500)
            else:
                t = dna(100) + q[:250] + dna(3) + q[260:] + dna(150)
        qs.append(("qy%d" % n, q))
        ts.append(("tg%d" % n, t))
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", model, "-E", "yes", "--showalignment", "yes", "--showvulgar", "yes",
            "--showcigar", "yes", "-V", "0"] + ([] if "-S" in extra else ["-S", "no"]) + extra + [qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1", "C4GPU_BATCH": batch})
    ref_out = ref.out()
    assert "c4gpu:" in gpu_err, "the GPU engine was not used:\n" + gpu_err[-1500:]
    assert ("c4gpu: batch of" in gpu_err) == (batch != "0"), gpu_err[-1500:]
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= (1 if "--bestn" in extra else 3)
    if "--percent" in extra:
        assert ref_out.count("vulgar:") < 9       # the unrelated query/target combinations are filtered


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,batch", [("est2genome", "4096"), ("est2genome", "0"), ("affine:local", "4096")])
def test_annotation_runs_are_served_by_the_device(tmp_path, model, batch):
    """exonerate's --annotation (sequence.c:51-88): a cDNA with a CDS annotation may not take part in a 1:1 DNA match inside its
    CDS (match.c:276-281).  Until round 5 the drop-in refused every model once a query carried an annotation; now the Optimal
    seams hand it over (c4gpu_batch_set_annotation; pinned on the reference's own Optimal_find_path with the annotation attached
    to the query: tests/golden/*_annot*.jsonl).  Through THIS reference's command line the option is accepted and changes
    nothing: Sequence_Annotation_compare (sequence.c:45-50) compares the sequence id with the BYTES of the annotation record
    (its first member is a pointer to the id, not the id), so Sequence_create_internal's tfind (sequence.c:176-178) never finds
    an entry and no Sequence ever carries one.  What this test holds: such a run is served by the device, byte-identical --
    and identical to the run without the option, as the reference's is (should the reference's lookup ever be repaired, this
    assertion fails and the seam's annotation path gets its end-to-end case)."""
    rng = random.Random(77 + len(model))
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    qs, ts, ann = [], [], []
    for n in range(3):
        q = dna(420 + 60 * n)
        if model == "est2genome":
            t = dna(900) + q[:180] + "GT" + dna(700) + "AG" + q[180:] + dna(800)
        else:
            t = dna(100) + q[:200] + dna(4) + q[210:] + dna(150)
        qs.append(("qy%d" % n, q))
        ts.append(("tg%d" % n, t))
        if n != 1:
            ann.append("qy%d + %d %d" % (n, 121 + 30 * n, 150))        # id strand cds_start (1-based) cds_length
    qf, tf, af = str(tmp_path / "q.fa"), str(tmp_path / "t.fa"), str(tmp_path / "q.annotation")
    _fasta(qf, qs)
    _fasta(tf, ts)
    with open(af, "w") as f:
        f.write("\n".join(ann) + "\n")
    base = ["-m", model, "-E", "yes", "--showalignment", "yes", "--showvulgar", "yes", "-V", "0", "-S", "no"]
    args = base + ["--annotation", af, qf, tf]
    ref = _Ref(args)
    plain = _Ref(base + [qf, tf])
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1", "C4GPU_BATCH": batch})
    ref_out = ref.out()
    assert "c4gpu:" in gpu_err and "using the CPU" not in gpu_err, "the GPU engine was not used:\n" + gpu_err[-1500:]
    assert ("c4gpu: batch of" in gpu_err) == (batch != "0"), gpu_err[-1500:]
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= 3 and ref_out == plain.out()


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,batch", [("est2genome", "4096"), ("affine:local", "4096"), ("est2genome", "0"),
                                         ("affine:local", "3")])
def test_exonerate_gpu_suboptimal_alignments(tmp_path, model, batch):
    """Targets with two copies of the gene: the default exhaustive run reports both (GAM's sub-optimal loop,
    gam.c:1158-1172); the calls that carry a SubOpt_Index go to the device as well."""
    rng = random.Random(31)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    mut = lambda s, r: "".join((rng.choice("ACGT") if rng.random() < r else c) for c in s)
    qs, ts = [], []
    for n in range(2):
        q = dna(400 + 50 * n)
        if model == "est2genome":
            gene = q[:180] + "GT" + dna(700) + "AG" + q[180:]
        else:
            gene = q
        t = dna(300) + mut(gene, 0.02) + dna(500) + mut(gene, 0.06) + dna(400)
        qs.append(("qy%d" % n, q))
        ts.append(("tg%d" % n, t))
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", model, "-E", "yes", "-S", "yes", "--showalignment", "no", "--showvulgar", "yes", "-V", "0",
            "--score", "300", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1", "C4GPU_BATCH": batch})
    ref_out = ref.out()
    if batch == "0":
        assert "with blocked cells" in gpu_err        # per-call: the SubOpt_Index itself crossed the boundary
    else:
        assert "round(s) on the device" in gpu_err and "with blocked cells" not in gpu_err
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= 4          # both copies, for both queries


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("refine", ["region", "full"])
def test_exonerate_gpu_heuristic_with_refinement(tmp_path, refine):
    """The default (heuristic, BSDP) mode with --refine: seeding, BSDP and its small derived-model DPs stay on
    the reference's CPU code; the refinement's Optimal_find_path (gam.c:605-655) runs on the device."""
    rng = random.Random(5)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    qs, ts = [], []
    for n in range(3):
        q = dna(600 + 100 * n)
        t = dna(2000) + q[:200] + "GT" + dna(1500) + "AG" + q[200:420] + "GT" + dna(3000) + "AG" + q[420:] + dna(2500)
        qs.append(("qy%d" % n, q))
        ts.append(("tg%d" % n, t))
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", "est2genome", "--refine", refine, "--showalignment", "yes", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert "c4gpu: est2genome mode" in gpu_err, gpu_err[-1500:]
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= 3


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_exonerate_gpu_command_line_switch(tmp_path):
    """--gpu no (beside -C/--compiled, codegen.c:25-37) keeps every call on the reference's CPU code;
    --gpubatch 0 selects the per-call shim."""
    rng = random.Random(9)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    q = dna(300)
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, [("qy", q)])
    _fasta(tf, [("tg", dna(200) + q[:150] + "GT" + dna(400) + "AG" + q[150:] + dna(100))])
    args = ["-m", "est2genome", "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    off_out, off_err = _run(GPU_EXE, ["--gpu", "no"] + args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert "c4gpu" not in off_err and off_out.replace("--gpu no ", "") == ref_out
    one_out, one_err = _run(GPU_EXE, ["--gpubatch", "0"] + args, {"C4GPU_VERBOSE": "1"})
    assert "c4gpu: est2genome mode" in one_err and "batch of" not in one_err
    assert one_out.replace("--gpubatch 0 ", "") == ref_out
    help_out = subprocess.run([GPU_EXE, "--help"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "--gpu" in help_out and "--gpubatch" in help_out


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_multi_process_query_shards_restore_submission_order(tmp_path):
    """integration/exonerate_multigpu.py: one process per GPU, sharded by query (the reference's own
    --querychunkid scheme); here 3 processes share device 0.  Output = the single-process output."""
    import sys
    rng = random.Random(12)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    qs = [("qy%d" % n, dna(300 + 40 * n)) for n in range(7)]
    ts = [("tg%d" % n, dna(150) + qs[n][1][:140] + "GT" + dna(300) + "AG" + qs[n][1][140:] + dna(90)) for n in range(7)]
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", "est2genome", "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "integration", "exonerate_multigpu.py"), "--gpus", "3",
                        "--devices", "0,0,0", "--"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    ref_out = ref.out()
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert r.stdout.decode() == ref_out
    assert ref_out.count("vulgar:") >= 7


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model", ["est2genome", "affine:local"])
def test_low_complexity_inputs_tie_everywhere(tmp_path, model):
    """Tandem repeats, homopolymers and a target made of many copies of the query at (near) north-star size:
    thousands of equal-score cells and paths, so the end cell (first in row-major order), the winning
    transition (lowest id) and the region start all hinge on tie-breaking.  Byte-identical to the reference."""
    rng = random.Random(77)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    unit = dna(97)
    q3 = dna(600)
    qs = [("tandem", "ACGT" * 200), ("polya", "A" * 700), ("copies", q3), ("unit", (unit * 9)[:800])]
    ts = [("tandem_t", "ACGT" * 6000), ("polya_t", "A" * 18000), ("copies_t", (q3[:300] + "GT" + "C" * 80 + "AG" + q3[300:]) * 20),
          ("unit_t", unit * 200)]
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", model, "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "--showcigar", "yes",
            "-V", "0", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert "c4gpu: batch of" in gpu_err
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= 4


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_selenocysteine_stays_in_the_device_batch(tmp_path):
    """ADVICE r01 worried that one residue outside the 24-letter matrix alphabet sends a whole flush to the CPU.  Through
    the binary that cannot happen: the reference's FASTA parser rejects every symbol outside its alphabets
    (alphabet.c:190: ARNDCQEGHILKMFPSTWYUV* + BZX), and the one member without a matrix row of its own, selenocysteine,
    is indexed as cysteine (submat.c:35,42) - the shim copies that index table, so the pair stays in the batch."""
    rng = random.Random(77)
    aa = lambda n: "".join(rng.choice("ARNDCQEGHILKMFPSTWYV") for _ in range(n))
    qs = [("qy%d" % k, aa(150 + 10 * k)) for k in range(4)]
    qs[2] = ("qy2", qs[2][1][:70] + "U" + qs[2][1][71:])
    ts = [("tg", aa(60) + "".join(q[:100] + aa(5) + q[100:] for _, q in qs) + aa(60))]
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qs)
    _fasta(tf, ts)
    args = ["-m", "affine:local", "-E", "yes", "-S", "no", "--showalignment", "yes", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert gpu_out == ref_out and ref_out.count("vulgar:") == 4
    assert "batch of 4 pairs" in gpu_err and "using the CPU Viterbi" not in gpu_err, gpu_err[-1500:]


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_c1_protein_heuristic_run_takes_the_sdp_batches(tmp_path):
    """BASELINE config 1: `--model affine:local`, 100 protein queries of ~300 aa against one ~10 kaa target, the DEFAULT
    heuristic mode (seeding + SDP).  The word hits are extended on the device (c4gpu_hsp.c) and every candidate pair's
    SDP comes from one c4gpu_sdp_batch (c4gpu_sdp.c); the drop-in must print what the reference prints.  With the device
    switched off (--gpu no) the same run passes through untouched."""
    rng = random.Random(20260929)
    aa = "ARNDCQEGHILKMFPSTWYV"
    queries = ["".join(rng.choice(aa) for _ in range(rng.randint(250, 350))) for _ in range(100)]
    target = []
    while sum(len(x) for x in target) < 10000:
        q = queries[rng.randrange(100)]
        a = rng.randrange(0, len(q) - 80)
        piece = q[a:a + rng.randint(60, 150)]
        target.append("".join((rng.choice(aa) if rng.random() < 0.10 else c) for c in piece))
        target.append("".join(rng.choice(aa) for _ in range(rng.randint(20, 120))))
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, [("prot%03d" % k, q) for k, q in enumerate(queries)])
    _fasta(tf, [("tg", "".join(target)[:10000])])
    args = ["-m", "affine:local", "--querytype", "protein", "--targettype", "protein", "--showalignment", "yes",
            "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert gpu_out == ref_out
    assert ref_out.count("vulgar:") >= 20
    m = re.search(r"c4gpu sdp: (\d+) pairs in (\d+) flush\(es\): (\d+) served from device batches \((\d+) alignments\)", gpu_err)
    assert m, gpu_err[-1500:]
    pairs, flushes, served_pairs, alignments = [int(x) for x in m.groups()]
    assert served_pairs == pairs >= 20 and alignments >= 20 and flushes == 1
    assert "c4gpu hsp:" in gpu_err
    off_out, off_err = _run(GPU_EXE, ["--gpu", "no"] + args, {"C4GPU_VERBOSE": "1"})
    assert off_out == ref_out and "c4gpu sdp:" not in off_err


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,extra,batch", [
    ("est2genome", [], "4096"), ("est2genome", ["-S", "no"], "4096"), ("est2genome", ["--refine", "region"], "4096"),
    ("est2genome", ["--refine", "region", "-S", "no"], "3"), ("est2genome", ["--refine", "full", "--bestn", "1"], "4096"),
    ("affine:local", [], "4096"), ("protein2dna", ["-S", "no"], "4096"), ("protein2genome", ["--refine", "region"], "4096"),
    ("est2genome", ["--percent", "50"], "2"),
])
def test_heuristic_bsdp_mode_runs_its_sub_dps_in_device_batches(tmp_path, model, extra, batch):
    """The default heuristic mode with --gappedextension no (BSDP): every collected pair's candidate sub-alignment
    regions (terminals, joins, span sources and destinations, their paths) go to the MI355X in a few launches per flush
    (integration/c4gpu_bsdp.c) and, with --refine, so does each pair's first refinement; the reference's own BSDP then
    confirms against those results.  Byte-identical output, and the share of its DP calls that the batches answered."""
    import re
    import test_integration_bsdp_host as hb
    ref, gpu, err = hb.run_pair(tmp_path, model, extra, {"C4GPU_BATCH": batch}, n=8, seed=11)
    assert gpu == ref
    assert ref.count(b"vulgar:") >= 4
    assert "stay on the CPU" not in err and "stay one call at a time" not in err, err[-1500:]
    assert "c4gpu hsp:" in err and "HSP extensions of this scan on the CPU" not in err, err[-800:]   # the seeding seam too
    s_ok, s_all, p_ok, p_all = hb.served(err)
    assert s_all > 20 and p_all > 10
    if "-S" in extra:                                     # nothing is ever blocked: every call comes from a batch
        assert (s_ok, p_ok) == (s_all, p_all), err[-600:]
    else:
        assert s_ok + p_ok >= 0.7 * (s_all + p_all), err[-600:]
    if "--refine" in extra:
        m = re.search(r"(\d+) of (\d+) refinements from refinement batches", err)
        assert m and int(m.group(2)) >= 4 and int(m.group(1)) >= int(m.group(2)) - 3, err[-800:]


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,extra,batch", [
    ("affine:local", [], "4096"), ("affine:local", ["--bestn", "1"], "4096"), ("affine:local", ["--percent", "40"], "3"),
    ("protein2dna", [], "4096"), ("protein2dna", ["-S", "no", "--extensionthreshold", "20"], "4096"),
    ("est2genome", [], "4096"), ("est2genome", ["--bestn", "1"], "3"), ("protein2genome", [], "4096"),
    ("protein2genome", ["--percent", "30", "--extensionthreshold", "25"], "4096"),
])
def test_heuristic_sdp_mode_takes_its_alignments_from_device_batches(tmp_path, model, extra, batch):
    """The DEFAULT heuristic mode (--gappedextension yes: SDP), both flavours (bidirectional from the seeds: affine,
    protein2dna; boundary + spans: est2genome, protein2genome): both Scheduler passes of every collected pair in two launches per flush (c4gpu_sdp_batch behind
    integration/c4gpu_sdp.c), the reference's own GAM_Result_SDP_create loop replayed on top.  Byte-identical output."""
    import test_integration_bsdp_host as hb
    ref, gpu, err = hb.run_pair(tmp_path, model, ["--gappedextension", "yes"] + extra, {"C4GPU_BATCH": batch}, n=8, seed=21)
    assert gpu == ref
    assert ref.count(b"vulgar:") >= 6
    assert "SDP stays on the CPU" not in err, err[-1500:]
    pairs, flushes, served_pairs, alignments = hb.sdp_served(err)
    assert served_pairs == pairs >= 6 and alignments >= 6
    assert "c4gpu hsp:" in err
    if batch == "3":
        # flushes cut in the middle of the run go to the device on a thread and a context of their own while the main thread
        # carries on with the comparisons behind them (c4gpu_sdp.c); C4GPU_SDP_ASYNC=0: every flush on the main thread
        m = re.search(r"(\d+) flush\(es\) beside the main thread", err)
        assert m and int(m.group(1)) >= 1 and flushes >= 2, err[-800:]
        ref, gpu, err = hb.run_pair(tmp_path, model, ["--gappedextension", "yes"] + extra,
                                    {"C4GPU_BATCH": batch, "C4GPU_SDP_ASYNC": "0"}, n=8, seed=21)
        assert gpu == ref and " 0 flush(es) beside the main thread" in err, err[-800:]
    if model == "est2genome" and not extra:
        # an arena too small for some pairs (C4GPU_SDP_ARENA_MB: a test hook; the second round's arena is eight times the
        # size): those pairs are run again or handed back to the reference's scheduler one by one, the output stays identical
        ref, gpu, err = hb.run_pair(tmp_path, model, ["--gappedextension", "yes"], {"C4GPU_SDP_ARENA_MB": "1.5"}, n=8, seed=21)
        assert gpu == ref
        assert "SDP stays on the CPU" not in err, err[-1500:]


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_heuristic_bsdp_on_north_star_shaped_input(tmp_path):
    """C4-shaped input (1 kb cDNAs against 100 kb genomic windows, all against all, both strands) through the heuristic
    BSDP mode with --refine region: byte-identical, >= 95 % of the sub-DP calls of a run without sub-optimal rounds
    answered by device batches."""
    import re
    import test_integration_bsdp_host as hb
    from exonerate_amd import workloads
    pairs = workloads.est2genome_pairs(6, 1000, 100000)
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, [("q%d" % k, q.decode()) for k, (q, t) in enumerate(pairs)])
    _fasta(tf, [("t%d" % k, t.decode()) for k, (q, t) in enumerate(pairs)])
    args = ["-m", "est2genome", "--gappedextension", "no", "--refine", "region", "-S", "no", "--showalignment", "yes",
            "--showvulgar", "yes", "-V", "0", qf, tf]
    ref = _Ref(args)
    gpu_out, gpu_err = _run(GPU_EXE, args, {"C4GPU_VERBOSE": "1"})
    ref_out = ref.out()
    assert gpu_out == ref_out and ref_out.count("vulgar:") >= 6
    s_ok, s_all, p_ok, p_all = hb.served(gpu_err)
    assert s_ok + p_ok >= 0.95 * (s_all + p_all), gpu_err[-800:]
    m = re.search(r"(\d+) of (\d+) refinements from refinement batches", gpu_err)
    assert m and int(m.group(1)) == int(m.group(2)) >= 6, gpu_err[-800:]


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_c5_heuristic_protein2genome_against_a_10mb_chromosome(tmp_path):
    """BASELINE config 5's HEURISTIC leg near its size: 32 proteins of 300 aa against ONE 10 Mb chromosome holding their
    intron-split genes, -m protein2genome in the reference's default mode (seeding -> HSPs -> SDP with spans,
    GAM_Result_SDP_create gam.c:852) and with --gappedextension no (BSDP).  The drop-in must print what the reference
    prints, byte for byte, with the word hits extended on the device and — default settings, no size limit — EVERY
    candidate pair's SDP served from the device batches."""
    import sys
    sys.path.insert(0, ROOT)
    from exonerate_amd import workloads
    import test_integration_bsdp_host as hb
    proteins, contig, places = workloads.protein_vs_contig(32, 300, 10000000, seed=20260935, introns=True)
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, [("p%d" % i, p.decode()) for i, p in enumerate(proteins)])
    _fasta(tf, [("chr", contig.decode())])
    base = ["-m", "protein2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0"]
    modes = {"sdp": [], "bsdp": ["--gappedextension", "no"]}
    refs = {k: subprocess.Popen([CPU_EXE] + base + m + [qf, tf], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for k, m in modes.items()}
    got = {}
    for k, m in modes.items():
        got[k] = _run(GPU_EXE, base + m + [qf, tf], {"C4GPU_VERBOSE": "1"})
    for k in modes:
        out, err = refs[k].communicate(timeout=1500)
        assert refs[k].returncode == 0, err.decode()[-800:]
        assert got[k][0] == out.decode(), k
        assert out.count(b"vulgar:") >= 32
        assert "c4gpu hsp:" in got[k][1], got[k][1][-1500:]
    err = got["sdp"][1]
    assert "SDP stays on the CPU" not in err, err[-1500:]
    pairs, flushes, served_pairs, alignments = hb.sdp_served(err)
    assert served_pairs == pairs >= 32 and alignments >= 32
    assert "c4gpu bsdp:" in got["bsdp"][1] and "stay on the CPU" not in got["bsdp"][1], got["bsdp"][1][-1500:]


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("model,extra", [
    ("est2genome", ["--gappedextension", "yes"]), ("affine:local", ["--gappedextension", "yes"]), ("protein2dna", ["--gappedextension", "yes"]),
    ("protein2genome", ["--gappedextension", "yes"]), ("affine:local", ["--forcefsm", "compact", "--gappedextension", "yes"]),
    ("est2genome", ["--gappedextension", "no"]),
])
def test_word_scan_on_the_device_equals_the_reference_walk(tmp_path, model, extra):
    """The automaton walk of Seeder_add_target on the device (integration/c4gpu_seed.c -> c4gpu_seed_scan), with
    C4GPU_SEED_CHECK=1: the reference's own FSM / VFSM traversal runs first and every word hit of the device scan must be
    the same seed in the same place of the list (the drop-in aborts otherwise); the rest of the heuristic pipeline (HSP
    extension, SDP or BSDP) runs on the device as usual and the output must be the reference's, byte for byte."""
    import test_integration_bsdp_host as hb
    ref, gpu, err = hb.run_pair(tmp_path, model, extra, {"C4GPU_SEED_CHECK": "1", "C4GPU_SEED_FACTOR": "0"}, n=8, seed=41)
    assert gpu == ref and ref.count(b"vulgar:") >= 4
    m = re.search(r"c4gpu seed: (\d+) targets walked in (\d+) device scans \((\d+) symbols\): (\d+) word hits", err)
    assert m and int(m.group(4)) > 100, err[-800:]
    assert "every seed equal to the reference's own walk" in err and "scanned on the CPU" not in err
    assert "c4gpu hsp:" in err


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
@pytest.mark.parametrize("extra", [["--gappedextension", "no", "-S", "no"], ["--gappedextension", "no"], [],
                                   ["-m", "protein2genome"]])
def test_small_work_does_not_wait_for_the_device(tmp_path, extra):
    """Without C4GPU_WAIT the seams in front of small work (word scan, HSP extension, BSDP sub-DPs, SDP) take the device only
    once it is open (shim_ctx_nowait): whatever comes earlier keeps the reference's own function, mid-run switches
    included (an HSP set that already holds an HSP stays with the reference's function).  The output is the reference's
    byte for byte whichever way each piece went; libc4gpu.so itself is bound lazily (no load before main)."""
    from exonerate_amd import workloads
    if "protein2genome" in extra:
        proteins, contig, _ = workloads.protein_vs_contig(12, 200, 400000, seed=5, introns=True)
        qrecs = [("p%d" % k, p.decode()) for k, p in enumerate(proteins)]
        trecs = [("chr", contig.decode())]
        args = list(extra)
    else:
        pairs = workloads.est2genome_pairs(8, 1000, 100000, seed=77)
        qrecs = [("q%d" % k, q.decode()) for k, (q, t) in enumerate(pairs)]
        trecs = [("t%d" % k, t.decode()) for k, (q, t) in enumerate(pairs)]
        args = ["-m", "est2genome"] + extra
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, qrecs)
    _fasta(tf, trecs)
    args += ["--showalignment", "yes", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref_out, _ = _run(CPU_EXE, args)
    env = {k: v for k, v in os.environ.items() if k != "C4GPU_WAIT"}
    for rep in range(3):                                   # the switch-over point moves from run to run
        r = subprocess.run([GPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout.decode() == ref_out and ref_out.count("vulgar:") >= 4
    ldd = subprocess.run(["ldd", GPU_EXE], stdout=subprocess.PIPE).stdout.decode()
    assert "libc4gpu" not in ldd and "amdhip" not in ldd


@pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                    reason="reference binaries are built in the build container (make -C integration)")
def test_short_runs_leave_with_the_device_thread_joined(tmp_path):
    """The way out of the drop-in (integration/c4gpu_shim.c, shim_quiesce): a run that is over while the device thread still
    loads code objects stops that warm-up (c4gpu_ctx_warm_cancel) and joins the thread before it leaves -- round 3 left with
    _exit under a thread still inside the runtime, round 4 with _exit after the join.  Round 5: the ordinary exit(), handlers
    and all (the crash that _exit papered over was getenv racing with the HIP start-up's setenv: shim_env; 1 500 of 1 500 short
    runs then left cleanly through exit()).  The 0.2 s heuristic est2genome run thirty times on the default way out, and the
    error path (exit(1) from general/argument.c's handler, pointed at shim_exit by the Makefile) with the device thread
    started: exit status and output as the reference's every time.  Six more times through _exit (C4GPU_FAST_EXIT=1)."""
    from exonerate_amd import workloads
    pairs = workloads.est2genome_pairs(8, 400, 40000, seed=123)
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    _fasta(qf, [("q%d" % k, q.decode()) for k, (q, t) in enumerate(pairs)])
    _fasta(tf, [("t%d" % k, t.decode()) for k, (q, t) in enumerate(pairs)])
    args = ["-m", "est2genome", "--showalignment", "no", "--showvulgar", "yes", "-V", "0", qf, tf]
    ref_out, _ = _run(CPU_EXE, args)
    assert ref_out.count("vulgar:") >= 4
    env = {k: v for k, v in os.environ.items() if k not in ("C4GPU_WAIT", "C4GPU_FAST_EXIT")}
    for rep in range(36):
        e = dict(env)
        if rep >= 30:
            e["C4GPU_FAST_EXIT"] = "1"
        r = subprocess.run([GPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=300)
        assert r.returncode == 0, (rep, r.returncode, r.stderr.decode()[-2000:])
        assert r.stdout.decode() == ref_out, rep
    # the error path: a protein query for a DNA model is refused after the options were parsed and the device thread started
    pf = str(tmp_path / "p.fa")
    _fasta(pf, [("p", "MKVLAAGIVGLLLAQWERTYHSAAPPKKLMNDE")])
    bad = ["-m", "est2genome", "-V", "0", pf, tf]
    rr = subprocess.run([CPU_EXE] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    for rep in range(6):
        r = subprocess.run([GPU_EXE] + bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        assert r.returncode == rr.returncode == 1, (rep, r.returncode, r.stderr.decode()[-1000:])
        assert r.stdout == rr.stdout
