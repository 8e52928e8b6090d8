"""The oracle's restatement of SDP (oracle/c4_oracle_sdp.c: SDP_Pair_next_path sdp.c:743 over Scheduler_Pair_calculate
scheduler.c:1445 and the interpreted Scheduler_Cell_process scheduler.c:859) against the reference's own alignments
(tests/golden/sdp_*.jsonl, refdump --cmd sdp: the loop of GAM_Result_SDP_create, gam.c:852, on a Comparison whose HSPset
grew from every shared word of the pair).  Both SDP flavours: bidirectional from the seeds (affine, protein2dna) and
boundary + span freeze/thaw (est2genome, protein2genome); default and lowered --extensionthreshold, non-default penalties
and intron window, --singlepass no on a boundary model (the seeded flavour asserts in the reference itself: sdp.c:611)."""
import pytest

import exonerate_amd as ex
import oracle_lib
from golden_util import PARAM_VARIANTS, apply_flags, load_set

# set -> (model type, query alphabet, target alphabet, (query advance, target advance) of the HSPset, parameter variant)
SDP_SETS = {
    "sdp_affine_local": ("affine:local", None, None, (1, 1), None),
    "sdp_affine_local_protein": ("affine:local", ex.ALPHABET_PROTEIN, ex.ALPHABET_PROTEIN, (1, 1), None),
    "sdp_est2genome": ("est2genome", None, None, (1, 1), None),
    "sdp_est2genome_drop": ("est2genome", None, None, (1, 1), None),
    "sdp_est2genome_altparams": ("est2genome", None, None, (1, 1), "altparams"),
    "sdp_protein2dna": ("protein2dna", None, None, (1, 3), None),
    "sdp_protein2genome": ("protein2genome", None, None, (1, 3), None),
    "sdp_protein2genome_altparams": ("protein2genome", None, None, (1, 3), "altparams"),
}


def sdp_case(name):
    mt, qa, ta, adv, variant = SDP_SETS[name]
    recs = load_set(name)
    par, recs = recs[0]["params"], recs[1:]
    params = ex.default_params() if variant is None else apply_flags(ex.default_params(), PARAM_VARIANTS[variant])
    model = ex.Model(mt, query_alphabet=qa, target_alphabet=ta, params=params)
    return model, par, recs, adv


def expected(r):
    return [{"score": a["path_score"], "region": a["region"], "ops": a["ops"], "vulgar": a["vulgar"]} for a in r["alignments"]]


@pytest.mark.parametrize("name", sorted(SDP_SETS))
def test_oracle_sdp_matches_reference(name):
    model, par, recs, adv = sdp_case(name)
    total = 0
    for r in recs:
        # the reference was asked for at most 4 alignments per pair
        ub, got = oracle_lib.sdp(model.c, model.params, r["query"].encode(), r["target"].encode(), r["hsps"], adv[0], adv[1],
                                 par["dropoff"], bool(par["singlepass"]), par["threshold"], 4, qid=r["id"])
        if r["hsps"]:
            assert ub == par["use_boundary"], r["id"]
        got = [{k: a[k] for k in ("score", "region", "ops", "vulgar")} for a in got]
        assert got == expected(r), r["id"]
        total += len(got)
    assert total >= 10


def test_sets_cover_both_flavours_and_later_alignments():
    flavours, later, spans = set(), 0, 0
    for name in SDP_SETS:
        recs = load_set(name)
        flavours.add(recs[0]["params"]["use_boundary"])
        later += sum(1 for r in recs[1:] if len(r["alignments"]) > 1)
        model = sdp_case(name)[0]
        for r in recs[1:]:
            for a in r["alignments"]:
                spans += sum(1 for t, _ in a["ops"] if model.c.transitions[t].label == 6)       # C4_Label_INTRON
    assert flavours == {0, 1} and later >= 20 and spans >= 20
