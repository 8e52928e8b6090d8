"""SURVEY 8f-4, first part: the GFF2 dump of --showtargetgff / --showquerygff (Alignment_display_gff, alignment.c:2710-3236)
from the library (c4gpu_alignment_format_gff) against the reference's own compiled exonerate (oracle/_ref, built from
/root/reference by oracle/Makefile, travelling as a binary): gene / utr / cds / exon / intron / splice / similarity lines with
their identity and similarity attributes, both report sides, forward and reverse-complemented targets, five models.  The
alignments themselves come from the oracle (the formatter needs no device)."""
import os, random, re, subprocess
import pytest

import exonerate_amd as ex
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_EXE = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
pytestmark = pytest.mark.skipif(not os.path.exists(CPU_EXE), reason="the reference binary is built in the build container")



def _align(model, qq, tt, threshold):
    """The alignment the printers are given: here the oracle's (the formatter needs no device); tests/test_gpu_printers.py runs
    the same cases with the alignments the MI355X makes."""
    exp = oracle_lib.find_path(model.c, model.params, qq.encode(), tt.encode(), dpmemory=32, threshold=threshold)
    assert exp is not None
    return ex.Alignment.from_parts(model, exp["score"], exp["region"], exp["ops"], len(qq), len(tt))


AA = "ARNDCQEGHILKMFPSTWYV"
TABLE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
CODON = {}
for _i, _a in enumerate("TCAG"):
    for _j, _b in enumerate("TCAG"):
        for _k, _c in enumerate("TCAG"):
            CODON.setdefault(TABLE[_i * 16 + _j * 4 + _k], []).append(_a + _b + _c)
COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def _revcomp(s):
    return s.translate(COMP)[::-1]


def _dna(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _mut(rng, s, rate, alpha):
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


def _case(rng, model_type, flip):
    if model_type in ("est2genome", "affine:local", "affine:global", "ungapped"):
        q = _dna(rng, 260)
        if model_type == "est2genome":
            m = _mut(rng, q, 0.04, "ACGT")
            t = _dna(rng, 90) + m[:80] + "GT" + _dna(rng, 150) + "AG" + m[80:170] + "GT" + _dna(rng, 230) + "AG" + m[170:] + _dna(rng, 70)
        else:
            t = _dna(rng, 40) + _mut(rng, q, 0.08, "ACGT") + _dna(rng, 60)
    else:
        q = "".join(rng.choice(AA) for _ in range(110))
        coding = "".join(rng.choice(CODON[x]) for x in _mut(rng, q, 0.06, AA))
        if model_type == "protein2genome":
            coding = coding[:100] + "GT" + _dna(rng, 180) + "AG" + coding[100:211] + "GT" + _dna(rng, 140) + "AG" + coding[211:]
        t = _dna(rng, 120) + coding + _dna(rng, 90)
        if model_type == "affine:local:protein":
            t = "".join(rng.choice(AA) for _ in range(20)) + _mut(rng, q, 0.1, AA) + "".join(rng.choice(AA) for _ in range(15))
    if flip and not model_type.endswith("protein"):
        t = _revcomp(t)                                   # the alignment is then on the target's reverse strand
    return q, t


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "affine:local", "protein2dna", "affine:local:protein"])
@pytest.mark.parametrize("flip", [False, True])
def test_gff_dump_is_the_reference_s(tmp_path, model_type, flip):
    rng = random.Random(len(model_type) * 31 + int(flip))
    q, t = _case(rng, model_type, flip)
    mt = "affine:local" if model_type.endswith(":protein") else model_type
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    qf.write_text(">qy some text\n%s\n" % q)
    tf.write_text(">tg\n%s\n" % t)
    args = ["-m", mt, "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "--showquerygff", "yes",
            "--showtargetgff", "yes", "-V", "0", str(qf), str(tf)]
    r = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    ref = r.stdout.decode()
    blocks = re.split(r"(?m)^(?=vulgar: )", ref)
    blocks = [b for b in blocks if b.startswith("vulgar: ")]
    assert blocks, ref[:500]
    protein_query = model_type.startswith("protein") or model_type.endswith(":protein")
    protein_target = model_type.endswith(":protein")
    model = ex.Model(mt, query_alphabet=1, target_alphabet=1) if model_type.endswith(":protein") else ex.Model(mt)
    seen_reverse = False
    for b in blocks:
        vulgar = b.splitlines()[0]
        f = vulgar.split()
        qstrand, tstrand = f[4], f[8]
        date = re.search(r"##date (\S+)", b).group(1)
        qq = _revcomp(q) if qstrand == "-" else q
        tt = _revcomp(t) if tstrand == "-" else t
        seen_reverse |= tstrand == "-" or qstrand == "-"
        a = _align(model, qq, tt, int(f[9]))
        assert a.vulgar("qy", "tg", qstrand, tstrand) == vulgar
        got = vulgar + "\n" + a.gff(qq, tt, "qy", "tg", qstrand, tstrand, on_query=True, date=date) + \
            a.gff(qq, tt, "qy", "tg", qstrand, tstrand, on_query=False, date=date)
        assert got == b, "\n".join(l for l in _diff(got, b))
    if flip and not protein_target:
        assert seen_reverse


def _diff(a, b):
    import difflib
    return list(difflib.unified_diff(b.splitlines(), a.splitlines(), "reference", "library", lineterm=""))[:60]


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "affine:local", "protein2dna", "affine:local:protein"])
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("width", [80, 57])
def test_alignment_display_is_the_reference_s(tmp_path, model_type, flip, width):
    """Alignment_display (alignment.c:234-1380), the default output of exonerate: header and rows of query / translation /
    match line / target with coordinates, gaps, codon gaps, splice sites, collapsed introns, split codons and the
    reverse-translation marks, against the reference binary (both strands, two display widths)."""
    rng = random.Random(len(model_type) * 131 + int(flip) + width)
    q, t = _case(rng, model_type, flip)
    mt = "affine:local" if model_type.endswith(":protein") else model_type
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    qf.write_text(">qy some text\n%s\n" % q)
    tf.write_text(">tg\n%s\n" % t)
    args = ["-m", mt, "-E", "yes", "-S", "no", "--showalignment", "yes", "--showvulgar", "yes", "--alignmentwidth", str(width),
            "-V", "0", str(qf), str(tf)]
    r = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    ref = r.stdout.decode()
    blocks = [b for b in re.split(r"(?m)(?=^\nC4 Alignment:\n)", "\n" + ref) if "C4 Alignment:" in b]
    blocks = re.findall(r"\nC4 Alignment:\n.*?\nvulgar: [^\n]*\n", ref, flags=re.S)
    assert blocks, ref[:400]
    model = ex.Model(mt, query_alphabet=1, target_alphabet=1) if model_type.endswith(":protein") else ex.Model(mt)
    for b in blocks:
        vulgar = b.rstrip("\n").splitlines()[-1]
        f = vulgar.split()
        qstrand, tstrand = f[4], f[8]
        qq = _revcomp(q) if qstrand == "-" else q
        tt = _revcomp(t) if tstrand == "-" else t
        a = _align(model, qq, tt, int(f[9]))
        assert a.vulgar("qy", "tg", qstrand, tstrand) == vulgar
        # Sequence_revcomp (sequence.c:405-410) marks the description of the strand it makes
        qdef = "some text:[revcomp]" if qstrand == "-" else "some text"
        tdef = "[revcomp]" if tstrand == "-" else None
        got = a.display(qq, tt, "qy", "tg", qstrand, tstrand, qdef=qdef, tdef=tdef, width=width) + vulgar + "\n"
        assert got == b, "\n".join(_diff(got, b))


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "protein2dna", "affine:local", "affine:global"])
def test_printers_on_random_cases(tmp_path, model_type):
    """Several pairs per model in one reference run -- frameshifts (single bases inserted into the coding sequence), codon
    gaps, split codons in both phases, introns on either gene orientation, ambiguity symbols, `--useaatla no
    --forwardcoordinates no` -- display and both GFF dumps byte for byte."""
    rng = random.Random(len(model_type) * 977)
    cases = []
    for k in range(6):
        q, t = _case(rng, model_type, flip=bool(k & 1))
        if model_type.startswith("protein") and k >= 2:            # frameshifts and an ambiguous base
            pos = sorted(rng.sample(range(130, len(t) - 100), 3))
            t = t[:pos[0]] + rng.choice("ACGT") + t[pos[0]:pos[1]] + t[pos[1] + 1:pos[2]] + "N" + t[pos[2] + 1:]
        if model_type in ("est2genome", "affine:local", "affine:global") and k >= 3:
            i = rng.randrange(20, len(q) - 20)
            q = q[:i] + "N" + q[i + 1:]
        cases.append((q, t))
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    extra_sets = [[], ["--useaatla", "no", "--forwardcoordinates", "no", "--alignmentwidth", "100"]]
    model = ex.Model(model_type)
    for extra in extra_sets:
        for k, (q, t) in enumerate(cases):
            qf.write_text(">q%d\n%s\n" % (k, q))
            tf.write_text(">t%d a target\n%s\n" % (k, t))
            args = ["-m", model_type, "-E", "yes", "-S", "no", "--showalignment", "yes", "--showvulgar", "yes", "--showtargetgff", "yes",
                    "--showquerygff", "yes", "-V", "0"] + extra + [str(qf), str(tf)]
            r = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            ref = r.stdout.decode()
            blocks = re.findall(r"\nC4 Alignment:\n.*?# --- END OF GFF DUMP ---\n#\n.*?# --- END OF GFF DUMP ---\n#\n", ref, flags=re.S)
            if model_type != "affine:global":
                assert blocks or "vulgar" not in ref
            fwd = "--forwardcoordinates" not in extra
            for b in blocks:
                vulgar = [l for l in b.splitlines() if l.startswith("vulgar: ")][0]
                f = vulgar.split()
                qstrand, tstrand = f[4], f[8]
                date = re.search(r"##date (\S+)", b).group(1)
                qq = _revcomp(q) if qstrand == "-" else q
                tt = _revcomp(t) if tstrand == "-" else t
                a = _align(model, qq, tt, int(f[9]))
                assert a.vulgar("q%d" % k, "t%d" % k, qstrand, tstrand, forward_coords=fwd) == vulgar
                qdef = "[revcomp]" if qstrand == "-" else None
                tdef = "a target:[revcomp]" if tstrand == "-" else "a target"
                got = a.display(qq, tt, "q%d" % k, "t%d" % k, qstrand, tstrand, qdef=qdef, tdef=tdef,
                                width=100 if extra else 80, forward_coords=fwd, use_aa_tla=not extra) + vulgar + "\n" + \
                    a.gff(qq, tt, "q%d" % k, "t%d" % k, qstrand, tstrand, on_query=True, date=date) + \
                    a.gff(qq, tt, "q%d" % k, "t%d" % k, qstrand, tstrand, on_query=False, date=date)
                assert got == b, "\n".join(_diff(got, b))


RYO_FORMATS = [
    "%S | %C | %V | %s %m %r %g %pi %ps %pI %pc %et %ei %es %em\\n",
    ">%qi %qd [%qS %qt %ql] %qab-%qae (%qal)\\n%qas>%ti %td [%tS %tt %tl] %tab-%tae (%tal)\\n%tas",
    "{%Pn|%Pl|%Ps|%Pqs|%Pts|%Pqa,%Pta|%Pqb-%Pqe|%Ptb-%Pte\\n}100%%\\t\\{x\\}\\\\\\n",
    "%qi %qs%ti %ts",
]


@pytest.mark.parametrize("model_type", ["est2genome", "protein2genome", "protein2dna", "affine:local", "affine:local:protein"])
def test_ryo_is_the_reference_s(tmp_path, model_type):
    """Alignment_display_ryo (alignment.c:1781-2669): every token -- sequence and alignment fields, percentages and
    equivalenced counts, sugar / cigar / vulgar blocks, the per-transition section with names, labels, scores (match, gap,
    splice-site, intron-length and split-codon calcs), residues and coordinates, coding sequences, escapes -- on both
    strands against the reference binary."""
    mt = "affine:local" if model_type.endswith(":protein") else model_type
    model = ex.Model(mt, query_alphabet=1, target_alphabet=1) if model_type.endswith(":protein") else ex.Model(mt)
    formats = list(RYO_FORMATS)
    if model_type.startswith("protein"):
        formats.append(">%ti cds %tcb %tce %tcl\\n%tcs")
    else:
        formats.append("self %pS\\n")
    rng = random.Random(len(model_type) * 313)
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    n_blocks = 0
    for k in range(4):
        q, t = _case(rng, model_type, flip=bool(k & 1))
        if model_type.startswith("protein") and k >= 2:
            i = rng.randrange(140, len(t) - 110)
            t = t[:i] + rng.choice("ACGT") + t[i:]
        qf.write_text(">qy a query\n%s\n" % q)
        tf.write_text(">tg\n%s\n" % t)
        for fmt in formats:
            sep = "@@%S %V@@ "
            args = ["-m", mt, "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "no", "--ryo", sep + "\\n" + fmt + "\\n##\\n",
                    "-V", "0", str(qf), str(tf)]
            r = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            ref = r.stdout.decode()
            for b in re.findall(r"@@[^\n]*@@ \n.*?\n##\n", ref, flags=re.S):
                f = ("vulgar: " + b[2:b.index("@@ \n")]).split()
                qstrand, tstrand = f[4], f[8]
                qq = _revcomp(q) if qstrand == "-" else q
                tt = _revcomp(t) if tstrand == "-" else t
                a = _align(model, qq, tt, int(f[9]))
                qdef = "a query:[revcomp]" if qstrand == "-" else "a query"
                tdef = "[revcomp]" if tstrand == "-" else None
                full = sep + "\\n" + fmt + "\\n##\\n"
                got = a.ryo(full, qq, tt, "qy", "tg", qstrand, tstrand, qdef=qdef, tdef=tdef, rank=0)
                assert got == b, (fmt, "\n".join(_diff(got, b)))
                n_blocks += 1
    assert n_blocks >= 4 * len(formats)


@pytest.mark.parametrize("model_type", ["protein2dna", "protein2genome"])
def test_percent_self_on_a_codon_match_is_an_error_where_the_reference_crashes(tmp_path, model_type):
    """`%pS` (Alignment_get_percent_self, alignment.c:1586-1618) scores the query against itself through self_data =
    Model_Type_create_data(type, query, query) (gam.c:599): for a protein-vs-DNA model that reads the PROTEIN query as the DNA
    side of a codon match (Match_3_translate_self_func, match.c:186-197, three residues past each position), and the
    reference binary dies of SIGSEGV there -- there is no output to reproduce.  The library says so instead of printing a
    number: c4gpu_alignment_format_ryo returns INT32_MIN (include/c4gpu.h), the Python mirror raises; every other token of the
    same string still prints for these models (test_ryo_is_the_reference_s)."""
    rng = random.Random(7 + len(model_type))
    q, t = _case(rng, model_type, flip=False)
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    qf.write_text(">qy\n%s\n" % q)
    tf.write_text(">tg\n%s\n" % t)
    r = subprocess.run([CPU_EXE, "-m", model_type, "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes",
                        "--ryo", "self %pS\\n", "-V", "0", str(qf), str(tf)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == -11 and b"self" not in r.stdout, (r.returncode, r.stdout[-300:])       # SIGSEGV before a byte of it
    r = subprocess.run([CPU_EXE, "-m", model_type, "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes",
                        "-V", "0", str(qf), str(tf)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    f = [l for l in r.stdout.decode().splitlines() if l.startswith("vulgar: ")][0].split()
    model = ex.Model(model_type)
    a = _align(model, q, t, int(f[9]))
    assert a.ryo("id %pi\\n", q, t) .startswith("id ")
    with pytest.raises(ex.C4GpuError, match="pS"):
        a.ryo("self %pS\\n", q, t)


def test_gff_lines_with_identifiers_longer_than_any_fixed_buffer(tmp_path):
    """The GFF printers build their lines with sequence identifiers in them ("Target %s %d %d", "sequence %s"): identifiers of
    700 bytes print whole, as the reference's g_strdup_printf does (round 3 cut them at 512 bytes)."""
    rng = random.Random(99)
    q, t = _case(rng, "est2genome", False)
    qid, tid = "q" * 700, "t" * 650
    qf, tf = tmp_path / "q.fa", tmp_path / "t.fa"
    qf.write_text(">%s\n%s\n" % (qid, q))
    tf.write_text(">%s\n%s\n" % (tid, t))
    args = ["-m", "est2genome", "-E", "yes", "-S", "no", "--showalignment", "no", "--showvulgar", "yes", "--showquerygff", "yes",
            "--showtargetgff", "yes", "-V", "0", str(qf), str(tf)]
    r = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    b = [x for x in re.split(r"(?m)^(?=vulgar: )", r.stdout.decode()) if x.startswith("vulgar: ")][0]
    vulgar = b.splitlines()[0]
    f = vulgar.split()
    date = re.search(r"##date (\S+)", b).group(1)
    model = ex.Model("est2genome")
    a = _align(model, q, t, int(f[9]))
    got = vulgar + "\n" + a.gff(q, t, qid, tid, f[4], f[8], on_query=True, date=date) + a.gff(q, t, qid, tid, f[4], f[8], on_query=False, date=date)
    assert got == b, "\n".join(_diff(got, b))
