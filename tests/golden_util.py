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
}


# sets with the GAM sub-optimal loop (rec["subopt"] = successive alignments, rec["threshold"])
SUBOPT_SETS = {
    "affine_local_dna_subopt": ("affine:local", 0, 0), "affine_local_dna_subopt_D0": ("affine:local", 0, 0),
    "affine_global_dna_subopt": ("affine:global", 0, 0),
    "est2genome_subopt": ("est2genome", 0, 0), "est2genome_subopt_D0": ("est2genome", 0, 0),
    "protein2dna_subopt": ("protein2dna", 1, 0), "protein2dna_subopt_D0": ("protein2dna", 1, 0),
    "protein2genome_subopt": ("protein2genome", 1, 0), "protein2genome_subopt_D0": ("protein2genome", 1, 0),
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


def set_params(lib, name):
    """Parameters a set was generated with (the defaults, or --forcegtag for the *_forcegtag sets)."""
    p = _abi.Params()
    lib.c4gpu_params_default(p)
    if "forcegtag" in name:
        lib.c4gpu_params_set_forcegtag(p, 1)
    return p


def get_model(lib, params, name):
    if name in DERIVED_SETS:
        mt, qa, ta, (src, dst, ss, es) = DERIVED_SETS[name]
        m = _abi.Model()
        assert lib.c4gpu_model_get_derived(mt.encode(), qa, ta, params, src, dst, ss, es, m, None) == 0
        return m
    mt, qa, ta = SETS[name] if name in SETS else SUBOPT_SETS[name]
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
