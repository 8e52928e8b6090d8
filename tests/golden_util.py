"""Helpers shared by the golden-vector tests (data only; no reference code)."""
import json, os
from exonerate_amd import _abi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# golden file -> (model type, query alphabet, target alphabet)
SETS = {
    "affine_local_dna": ("affine:local", 0, 0), "affine_global_dna": ("affine:global", 0, 0),
    "affine_bestfit_dna": ("affine:bestfit", 0, 0), "affine_overlap_dna": ("affine:overlap", 0, 0),
    "affine_local_protein": ("affine:local", 1, 1), "affine_global_protein": ("affine:global", 1, 1),
    "affine_bestfit_protein": ("affine:bestfit", 1, 1), "affine_overlap_protein": ("affine:overlap", 1, 1),
    "est2genome": ("est2genome", 0, 0), "protein2dna": ("protein2dna", 1, 0),
    "affine_local_dna_D0": ("affine:local", 0, 0), "affine_global_dna_D0": ("affine:global", 0, 0),
    "est2genome_D0": ("est2genome", 0, 0), "protein2dna_D0": ("protein2dna", 1, 0),
    "est2genome_big": ("est2genome", 0, 0),
    "protein2genome": ("protein2genome", 1, 0), "protein2genome_D0": ("protein2genome", 1, 0),
    "ungapped_dna": ("ungapped", 0, 0), "ungapped_protein": ("ungapped", 1, 1), "ungapped_dna_D0": ("ungapped", 0, 0),
    "protein2dna_bestfit": ("protein2dna:bestfit", 1, 0), "protein2dna_bestfit_D0": ("protein2dna:bestfit", 1, 0),
    "protein2genome_bestfit": ("protein2genome:bestfit", 1, 0),
    "protein2genome_bestfit_D0": ("protein2genome:bestfit", 1, 0),
    "est2genome_forcegtag": ("est2genome", 0, 0), "est2genome_forcegtag_D0": ("est2genome", 0, 0),
    "protein2genome_forcegtag": ("protein2genome", 1, 0),
    # non-default penalties / intron window / matrices (tools/make_golden.py:ALT_FLAGS, parsed by the reference)
    "affine_local_dna_altparams": ("affine:local", 0, 0), "affine_local_dna_altparams_D0": ("affine:local", 0, 0),
    "affine_global_protein_altparams": ("affine:global", 1, 1),
    "est2genome_altparams": ("est2genome", 0, 0), "est2genome_altparams_D0": ("est2genome", 0, 0),
    "protein2dna_altparams": ("protein2dna", 1, 0), "protein2dna_altparams_D0": ("protein2dna", 1, 0),
    "protein2genome_altparams": ("protein2genome", 1, 0), "protein2genome_altparams_D0": ("protein2genome", 1, 0),
}


for _tag, _bases in (("hugegap", ("affine_local_dna", "est2genome", "protein2dna", "protein2genome")),
                     ("hugeintron", ("est2genome", "protein2dna", "protein2genome")),
                     ("tightintron", ("est2genome", "protein2genome")),
                     ("invertedintron", ("est2genome", "protein2genome")),
                     ("posgap", ("affine_local_dna", "est2genome", "protein2dna", "protein2genome"))):
    for _b in _bases:
        SETS["%s_%s" % (_b, _tag)] = SETS[_b]
        SETS["%s_%s_D0" % (_b, _tag)] = SETS[_b]


# --annotation (match.c:276-281): DNA queries with a CDS annotation, rec["cds"] = [cds_start, cds_length]; made by the reference
# with the annotation attached to the query Sequence (oracle/refdump.c, tools/make_golden.py: annotated)
ANNOT_SETS = {"est2genome_annot": ("est2genome", 0, 0), "est2genome_annot_D0": ("est2genome", 0, 0),
              "affine_local_dna_annot": ("affine:local", 0, 0), "affine_local_dna_annot_D0": ("affine:local", 0, 0)}


# sets with the GAM sub-optimal loop (rec["subopt"] = successive alignments, rec["threshold"])
SUBOPT_SETS = {
    "affine_local_dna_subopt": ("affine:local", 0, 0), "affine_local_dna_subopt_D0": ("affine:local", 0, 0),
    "affine_global_dna_subopt": ("affine:global", 0, 0),
    "est2genome_subopt": ("est2genome", 0, 0), "est2genome_subopt_D0": ("est2genome", 0, 0),
    "protein2dna_subopt": ("protein2dna", 1, 0), "protein2dna_subopt_D0": ("protein2dna", 1, 0),
    "protein2genome_subopt": ("protein2genome", 1, 0), "protein2genome_subopt_D0": ("protein2genome", 1, 0),
    "est2genome_altparams_subopt": ("est2genome", 0, 0),
}


# BSDP's derived models: golden file -> (model type, query alphabet, target alphabet, (src, dst, start scope, end scope))
DERIVED_SETS = {}
for _tag, _mt, _qa, _ta, _ms in (("affine_local", "affine:local", 0, 0, (2,)), ("est2genome", "est2genome", 0, 0, (2, 5)),
                                 ("protein2dna", "protein2dna", 1, 0, (2,)), ("protein2genome", "protein2genome", 1, 0, (2,))):
    for _m in _ms:
        _sfx = "" if len(_ms) == 1 else ("_fwd" if _m == 2 else "_rev")
        DERIVED_SETS["derived_%s%s_start" % (_tag, _sfx)] = (_mt, _qa, _ta, (0, _m, 0, 4))
        DERIVED_SETS["derived_%s%s_end" % (_tag, _sfx)] = (_mt, _qa, _ta, (_m, 1, 4, 0))
        DERIVED_SETS["derived_%s%s_join" % (_tag, _sfx)] = (_mt, _qa, _ta, (_m, _m, 4, 4))


def load_set(name):
    with open(os.path.join(GOLDEN_DIR, name + ".jsonl")) as f:
        return [json.loads(l) for l in f if l.strip()]


# Scoring-parameter variants of the vector sets: the reference's own command-line flags (parsed by its
# ArgumentSets in refdump: affine.c:24-49, intron.c:24-32, frameshift.c, match.c) that tools/make_golden.py passes
# when it generates "<set>_<tag>" — and that the tests turn into a c4gpu_params here.  The two non-default
# matrices travel as data in scoring_data_alt.json (dumped by refdump --cmd data).
PARAM_VARIANTS = {
    "altparams": ["--gapopen", "-7", "--gapextend", "-2", "--codongapopen", "-11", "--codongapextend", "-3",
                  "--intronpenalty", "-14", "--minintron", "36", "--maxintron", "180", "--frameshift", "-13",
                  "--dnasubmat", "identity", "--proteinsubmat", "pam250"],
    # magnitudes at which "unset" (-987654321) and real scores are no longer far apart (local models only: a
    # global model's gap chains would overflow int32 in the reference itself)
    "hugegap": ["--gapopen", "-350000000", "--gapextend", "-300000000", "--codongapopen", "-350000000",
                "--codongapextend", "-300000000"],
    "hugeintron": ["--intronpenalty", "-350000000", "--frameshift", "-300000000"],
    "tightintron": ["--minintron", "77", "--maxintron", "78", "--intronpenalty", "-1"],
    "invertedintron": ["--minintron", "500", "--maxintron", "40"],       # every intron is rejected
    "posgap": ["--gapopen", "3", "--gapextend", "1", "--codongapopen", "4", "--codongapextend", "2",
               "--frameshift", "2"],                                      # rewards: nonsense the reference accepts
}
_FLAG_FIELD = {"--gapopen": "gap_open", "--gapextend": "gap_extend", "--codongapopen": "codon_gap_open",
               "--codongapextend": "codon_gap_extend", "--intronpenalty": "intron_open_penalty",
               "--minintron": "min_intron", "--maxintron": "max_intron", "--frameshift": "frameshift_penalty"}


def apply_flags(p, flags):
    for k in range(0, len(flags), 2):
        flag, val = flags[k], flags[k + 1]
        if flag in _FLAG_FIELD:
            setattr(p, _FLAG_FIELD[flag], int(val))
        elif flag in ("--dnasubmat", "--proteinsubmat"):
            with open(os.path.join(GOLDEN_DIR, "scoring_data_alt.json")) as f:
                mat = json.load(f)[flag[2:] + ":" + val]
            dst = p.dna_submat if flag == "--dnasubmat" else p.protein_submat
            for i in range(24):
                for j in range(24):
                    dst[i][j] = mat[i][j]
        else:
            raise ValueError(flag)
    return p


def set_params(lib, name):
    """Parameters a set was generated with: the defaults, --forcegtag for the *_forcegtag sets, and the flags of
    the PARAM_VARIANTS tag in the set's name."""
    p = _abi.Params()
    lib.c4gpu_params_default(p)
    if "forcegtag" in name:
        lib.c4gpu_params_set_forcegtag(p, 1)
    for tag, flags in PARAM_VARIANTS.items():
        if ("_" + tag) in name:
            apply_flags(p, flags)
    return p


def get_model(lib, params, name):
    if name in DERIVED_SETS:
        mt, qa, ta, (src, dst, ss, es) = DERIVED_SETS[name]
        m = _abi.Model()
        assert lib.c4gpu_model_get_derived(mt.encode(), qa, ta, params, src, dst, ss, es, m, None) == 0
        return m
    mt, qa, ta = SETS[name] if name in SETS else ANNOT_SETS[name] if name in ANNOT_SETS else SUBOPT_SETS[name]
    m = _abi.Model()
    assert lib.c4gpu_model_get(mt.encode(), qa, ta, params, m) == 0
    return m


def expected(rec):
    """What the reference printed for this record, in the shape oracle_lib.alignment_to_dict gives."""
    return {"score": rec["path_score"], "region": rec["region"], "ops": rec["ops"],
            "sugar": rec["sugar"], "cigar": rec["cigar"], "vulgar": rec["vulgar"]}


# BSDP span models: (set, model, query alphabet (1 = protein), match state, span state)
SPAN_SETS = [("span_est2genome_fwd", "est2genome", 0, 2, 8), ("span_est2genome_rev", "est2genome", 0, 5, 9),
             ("span_protein2genome_phase0", "protein2genome", 1, 2, 10),
             ("span_protein2genome_phase1", "protein2genome", 1, 2, 11),
             ("span_protein2genome_phase2", "protein2genome", 1, 2, 12)]
