"""The heuristic (BSDP) seam of the drop-in binary, checked WITHOUT a device: C4GPU_BSDP_HOST=1 makes step 3 of
integration/c4gpu_bsdp.c (all candidate sub-DPs of all collected pairs) run on the reference's own
Optimal_find_score / Optimal_find_path instead of the device, so steps 1, 2 and 4 — collecting the comparisons, the dry
runs that write the candidates down, the replay that answers SAR_*_find_score and the path calls from the batch — are
exercised here, byte for byte against the unmodified reference.  The same run also checks that every derived model the
reference's Heuristic builds flattens and has a compiled device family.  The device route itself: test_integration_gpu.py."""
import os, random, re, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_EXE = os.path.join(ROOT, "integration", "_build", "exonerate-gpu")
CPU_EXE = os.path.join(ROOT, "oracle", "_ref", "exonerate-compiled")
pytestmark = pytest.mark.skipif(not (os.path.exists(GPU_EXE) and os.path.exists(CPU_EXE)),
                                reason="reference binaries are built in the build container (make -C integration)")

TABLE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
CODON = {}
for _i, _a in enumerate("TCAG"):
    for _j, _b in enumerate("TCAG"):
        for _k, _c in enumerate("TCAG"):
            CODON.setdefault(TABLE[_i * 16 + _j * 4 + _k], []).append(_a + _b + _c)


def heuristic_inputs(model, n, seed):
    """Related query / target sets: genes with introns (est2genome, protein2genome), mutated copies otherwise; every
    query has a partner target, and some targets hold a second copy (sub-optimal rounds)."""
    rng = random.Random(seed)
    dna = lambda k: "".join(rng.choice("ACGT") for _ in range(k))
    aa = lambda k: "".join(rng.choice("ARNDCQEGHILKMFPSTWYV") for _ in range(k))
    mut = lambda s, r, alpha="ACGT": "".join((rng.choice(alpha) if rng.random() < r else c) for c in s)
    qs, ts = [], []
    for k in range(n):
        if model.startswith("protein"):
            q = aa(rng.randint(120, 260))
            coding = "".join(rng.choice(CODON[x]) for x in mut(q, 0.05, "ARNDCQEGHILKMFPSTWYV"))
            if model == "protein2genome":
                cuts = sorted(rng.sample(range(60, len(coding) - 60), 2))
                coding = coding[:cuts[0]] + "GT" + dna(rng.randint(80, 600)) + "AG" + coding[cuts[0]:cuts[1]] + \
                    "GT" + dna(rng.randint(80, 900)) + "AG" + coding[cuts[1]:]
            t = dna(rng.randint(100, 800)) + coding + dna(rng.randint(100, 800))
        else:
            q = dna(rng.randint(400, 900))
            if model == "est2genome":
                cuts = sorted(rng.sample(range(80, len(q) - 80), 3))
                parts = [q[:cuts[0]], q[cuts[0]:cuts[1]], q[cuts[1]:cuts[2]], q[cuts[2]:]]
                gene = mut(parts[0], 0.03)
                for p in parts[1:]:
                    gene += "GT" + dna(rng.randint(60, 2500)) + "AG" + mut(p, 0.03)
            else:
                gene = mut(q[:300], 0.05) + dna(rng.randint(0, 30)) + mut(q[300:], 0.05)
            t = dna(rng.randint(200, 3000)) + gene + dna(rng.randint(200, 3000))
            if k % 3 == 2:
                t += dna(300) + mut(gene, 0.08)
        qs.append(("q%d" % k, q))
        ts.append(("t%d" % k, t))
    return qs, ts


def run_pair(tmp_path, model, extra, env_extra, n=6, seed=5):
    qs, ts = heuristic_inputs(model, n, seed)
    qf, tf = str(tmp_path / "q.fa"), str(tmp_path / "t.fa")
    for path, recs in ((qf, qs), (tf, ts)):
        with open(path, "w") as f:
            for name, seq in recs:
                f.write(">%s\n%s\n" % (name, seq))
    mode = [] if "--gappedextension" in extra else ["--gappedextension", "no"]       # BSDP unless the case says otherwise
    args = ["-m", model] + mode + ["--showalignment", "yes", "--showvulgar", "yes", "-V", "0"] + list(extra) + [qf, tf]
    ref = subprocess.run([CPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    gpu = subprocess.run([GPU_EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                         env=dict(os.environ, C4GPU_VERBOSE="1", **env_extra))
    assert ref.returncode == 0, ref.stderr.decode()[-800:]
    assert gpu.returncode == 0, gpu.stderr.decode()[-1500:]
    return ref.stdout, gpu.stdout, gpu.stderr.decode()


def served(err):
    m = re.search(r"(\d+) of (\d+) score calls and (\d+) of (\d+) path calls served", err)
    assert m, err[-1500:]
    return [int(x) for x in m.groups()]


@pytest.mark.parametrize("model,extra", [
    ("est2genome", []), ("est2genome", ["-S", "no"]), ("est2genome", ["--bestn", "1"]),
    ("affine:local", []), ("protein2dna", ["-S", "no"]), ("protein2genome", []),
    ("est2genome", ["--score", "300", "--percent", "40"]),
])
def test_bsdp_seam_is_byte_identical_with_host_scores(tmp_path, model, extra):
    ref, gpu, err = run_pair(tmp_path, model, extra, {"C4GPU_BSDP_HOST": "1"})
    assert gpu == ref
    assert ref.count(b"vulgar:") >= 3
    assert "has no device family" not in err, err[-1500:]
    s_ok, s_all, p_ok, p_all = served(err)
    assert s_all > 0 and p_all > 0
    if "-S" in extra:                         # no sub-optimal rounds: nothing is ever blocked, everything is served
        assert (s_ok, p_ok) == (s_all, p_all), err[-600:]
    else:
        assert s_ok + p_ok >= 0.7 * (s_all + p_all), err[-600:]


@pytest.mark.parametrize("model,extra", [("est2genome", ["--refine", "region"]), ("est2genome", ["--refine", "full", "-S", "no"]),
                                         ("affine:local", ["--refine", "region", "--bestn", "1"]),
                                         ("protein2genome", ["--refine", "region"])])
def test_first_refinements_are_batched_too(tmp_path, model, extra):
    """--refine: every pair's first GAM_Result_refine_alignment call (gam.c:605-655) is found by a second dry run and
    answered from one batch (here: the reference's Optimal_find_path standing in for the device)."""
    ref, gpu, err = run_pair(tmp_path, model, extra, {"C4GPU_BSDP_HOST": "1"}, n=4)
    assert gpu == ref
    m = re.search(r"(\d+) of (\d+) refinements from refinement batches", err)
    assert m and int(m.group(2)) >= 3 and int(m.group(1)) >= int(m.group(2)) - 2, err[-800:]


def test_several_flushes_keep_the_submission_order(tmp_path):
    ref, gpu, err = run_pair(tmp_path, "est2genome", [], {"C4GPU_BSDP_HOST": "1", "C4GPU_BATCH": "4"}, n=7)
    assert gpu == ref
    assert re.search(r"in (\d+) flush", err) and int(re.search(r"in (\d+) flush", err).group(1)) >= 3


@pytest.mark.parametrize("model,extra", [("est2genome", []), ("est2genome", ["--gappedextension", "yes"]), ("affine:local", []),
                                         ("protein2dna", []), ("protein2genome", ["--hspfilter", "3"]), ("ungapped", []),
                                         ("est2genome", ["--dnahspthreshold", "40", "--dnahspdropoff", "10"])])
def test_seeding_seam_is_byte_identical_with_host_extensions(tmp_path, model, extra):
    """The seeding seam (integration/c4gpu_hsp.c): word hits are written down during the scan, extended in one batch
    per target scan and replayed through the horizon test and HSPset_add_known_hsp.  C4GPU_HSP_HOST=1: the extensions
    come from the reference's own HSPset_seed_hsp on scratch sets; the device route is checked in test_integration_gpu.py."""
    ref, gpu, err = run_pair(tmp_path, model, extra, {"C4GPU_BSDP_HOST": "1", "C4GPU_HSP_HOST": "1"})
    assert gpu == ref and ref.count(b"vulgar:") >= 3
    m = re.search(r"c4gpu hsp: (\d+) word hits of (\d+) HSP sets extended in (\d+) device batch", err)
    assert m and int(m.group(1)) > 100 and int(m.group(3)) <= int(m.group(2)), err[-600:]


def test_switching_the_seam_off(tmp_path):
    ref, gpu, err = run_pair(tmp_path, "est2genome", [], {"C4GPU_BSDP_OFF": "1", "C4GPU_HSP_OFF": "1"})
    assert gpu == ref and "c4gpu bsdp" not in err and "c4gpu hsp" not in err


# ---- the SDP seam (integration/c4gpu_sdp.c): --gappedextension yes, the models the reference runs without a boundary ----

def sdp_served(err):
    m = re.search(r"c4gpu sdp: (\d+) pairs in (\d+) flush\(es\): (\d+) served from device batches \((\d+) alignments\)", err)
    assert m, err[-1500:]
    return [int(x) for x in m.groups()]


@pytest.mark.parametrize("model,extra", [
    ("affine:local", []), ("affine:local", ["--bestn", "1"]), ("affine:local", ["--percent", "40", "--extensionthreshold", "20"]),
    ("protein2dna", []), ("protein2dna", ["-S", "no", "--score", "60"]),
    ("est2genome", []), ("est2genome", ["--bestn", "1", "--extensionthreshold", "20"]), ("protein2genome", []),
])
def test_sdp_seam_is_byte_identical_with_host_alignments(tmp_path, model, extra):
    """C4GPU_SDP_HOST=1: the collection, the hand-out of a batch's alignments by the front of SDP_Pair_next_path
    (thresholds that rise between calls under --bestn / --percent included) and the replay order, with the reference's
    own SDP computing the batch on the host."""
    ref, gpu, err = run_pair(tmp_path, model, ["--gappedextension", "yes"] + extra,
                             {"C4GPU_SDP_HOST": "1", "C4GPU_HSP_HOST": "1"}, n=8, seed=21)
    assert gpu == ref and ref.count(b"vulgar:") >= 6
    pairs, flushes, served_pairs, alignments = sdp_served(err)
    assert flushes == 1 and served_pairs == pairs >= 6 and alignments >= 6


def test_sdp_seam_leaves_refinement_alone(tmp_path):
    for model, extra in (("est2genome", ["--refine", "region"]), ("affine:local", ["--refine", "region"])):
        ref, gpu, err = run_pair(tmp_path, model, ["--gappedextension", "yes"] + extra,
                                 {"C4GPU_SDP_HOST": "1", "C4GPU_HSP_HOST": "1", "C4GPU_DISABLE": "1"}, n=4, seed=22)
        assert gpu == ref and ref.count(b"vulgar:") >= 3
        assert "c4gpu sdp:" not in err


def test_sdp_seam_in_small_flushes_and_switched_off(tmp_path):
    ref, gpu, err = run_pair(tmp_path, "affine:local", ["--gappedextension", "yes"],
                             {"C4GPU_SDP_HOST": "1", "C4GPU_HSP_HOST": "1", "C4GPU_BATCH": "3"}, n=8, seed=23)
    assert gpu == ref
    assert sdp_served(err)[1] >= 3
    ref, gpu, err = run_pair(tmp_path, "affine:local", ["--gappedextension", "yes"],
                             {"C4GPU_SDP_HOST": "1", "C4GPU_HSP_HOST": "1", "C4GPU_SDP_OFF": "1"}, n=4, seed=23)
    assert gpu == ref and "c4gpu sdp:" not in err


# ---- the word-scan seam (integration/c4gpu_seed.c): the automaton walk of Seeder_add_target ----

@pytest.mark.parametrize("model,extra", [
    ("est2genome", []), ("affine:local", ["--gappedextension", "yes"]), ("protein2dna", []), ("protein2genome", ["--gappedextension", "yes"]),
    ("ungapped", []), ("est2genome", ["--fsmmemory", "1"]), ("affine:local", ["--forcefsm", "compact", "--gappedextension", "yes"]),
    ("est2genome", ["--dnawordlen", "9"]), ("protein2dna", ["--proteinwordlen", "4", "--proteinwordlimit", "3"]),
])
def test_word_scan_seam_equals_the_reference_walk(tmp_path, model, extra):
    """C4GPU_SEED_HOST=1: the words and emission lists read off the reference's automaton (trie before compilation, or the
    VFSM leaf table with --forcefsm compact), a dictionary scan in place of the device scan, the hits delivered like
    Seeder_WordInfo_seed; C4GPU_SEED_CHECK=1: the reference's own walk runs first and every seed must be the same, in the
    same order (the drop-in aborts otherwise).  Byte-identical output on top."""
    # C4GPU_SEED_FACTOR=0: the word table is read at the first target (by default only once a seeder's targets add up to 16
    # symbols per trie node: small runs keep the reference's walk)
    env = {"C4GPU_SEED_HOST": "1", "C4GPU_SEED_CHECK": "1", "C4GPU_HSP_HOST": "1", "C4GPU_SDP_HOST": "1", "C4GPU_BSDP_HOST": "1",
           "C4GPU_SEED_FACTOR": "0"}
    ref, gpu, err = run_pair(tmp_path, model, extra, env, n=6, seed=31)
    assert gpu == ref and ref.count(b"vulgar:") >= 3
    m = re.search(r"c4gpu seed: (\d+) targets walked in (\d+) device scans \((\d+) symbols\): (\d+) word hits", err)
    assert m and int(m.group(4)) > 100, err[-800:]
    assert "every seed equal to the reference's own walk" in err
    if model.startswith("protein2"):
        assert int(m.group(2)) == 3 * int(m.group(1))                  # three translated frames per target
    if not extra or model == "protein2genome":
        # the words of the last trie level read by several threads (slices joined in order; by default only from 4 096 nodes on)
        ref, gpu, err = run_pair(tmp_path, model, extra, dict(env, C4GPU_SEED_THREADS="3"), n=6, seed=31)
        assert gpu == ref and "every seed equal to the reference's own walk" in err


@pytest.mark.parametrize("model,extra", [("est2genome", []), ("protein2dna", []), ("affine:local", ["--gappedextension", "yes"])])
def test_word_table_read_after_the_automaton_was_compiled(tmp_path, model, extra):
    """The first targets of a seeder go through the reference's own walk (which compiles the automaton: failure links
    everywhere); once the targets add up, the words are read off the COMPILED automaton level by level and the later
    targets are scanned from the table — seed for seed what the reference's walk finds (C4GPU_SEED_CHECK)."""
    env = {"C4GPU_SEED_HOST": "1", "C4GPU_SEED_CHECK": "1", "C4GPU_HSP_HOST": "1", "C4GPU_SDP_HOST": "1", "C4GPU_BSDP_HOST": "1"}
    for factor in ("0.05", "0.2", "0.5", "1", "2"):
        ref, gpu, err = run_pair(tmp_path, model, extra, dict(env, C4GPU_SEED_FACTOR=factor), n=8, seed=33)
        assert gpu == ref
        if "left to the reference's walk" in err and "device scans" in err:
            assert "every seed equal to the reference's own walk" in err
            return
    assert False, "no factor made the seeder switch in the middle of its targets: " + err[-600:]


def test_word_scan_seam_switched_off_and_with_a_saturation_threshold(tmp_path):
    env = {"C4GPU_SEED_HOST": "1", "C4GPU_HSP_HOST": "1", "C4GPU_BSDP_HOST": "1", "C4GPU_SEED_FACTOR": "0"}
    ref, gpu, err = run_pair(tmp_path, "est2genome", [], dict(env, C4GPU_SEED_OFF="1"))
    assert gpu == ref and "c4gpu seed:" not in err
    # --saturatethreshold: a running count per word in walk order, the reference's own walk keeps it
    ref, gpu, err = run_pair(tmp_path, "est2genome", ["--saturatethreshold", "5"], env)
    assert gpu == ref and "c4gpu seed:" not in err
