"""The oracle's restatement of HSP seeding (HSPset_seed_hsp, src/comparison/hspset.c:933: trim, initial score, ungapped
X-drop extension, horizon filter, threshold, cobs) against what the reference itself produced (tests/golden/hsp_*.jsonl,
refdump --cmd hsp): per seed on a fresh HSPset, and whole HSPsets fed their seeds in scan order; DNA, protein and
protein-vs-DNA matches, default and lowered thresholds / dropoffs, and the inputs of the reference's hspset.test.c."""
import pytest

import oracle_lib
from golden_util import load_set

HSP_SETS = ["hsp_dna2dna", "hsp_dna2dna_low", "hsp_protein2protein", "hsp_protein2dna", "hsp_protein2dna_drop"]


@pytest.mark.parametrize("name", HSP_SETS)
def test_oracle_hsps_match_reference(lib, params, name):
    recs = load_set(name)
    par, recs = recs[0]["params"], recs[1:]
    assert par["seed_repeat"] == 1
    for r in recs:
        q, t = r["query"].encode(), r["target"].encode()
        for (qs, ts), exp in zip(r["seeds"], r["single"]):
            got = oracle_lib.hsp_extend(params, par["match"], q, t, par["seedlen"], par["dropoff"], qs, ts)
            if exp is None:                       # below the threshold: nothing stored (HSP_store, hspset.c:885-888)
                assert got[3] < par["threshold"], (r["id"], qs, ts)
            else:
                assert got == exp, (r["id"], qs, ts)
        assert oracle_lib.hsp_set(params, par["match"], q, t, par["seedlen"], par["dropoff"], par["threshold"],
                                  r["seeds"]) == r["set"], r["id"]


def test_reference_hspset_test_inputs(lib, params):
    """src/comparison/hspset.test.c:49-66 (it prints, it does not assert): the HSPs the reference grows from its seeds."""
    d2d = [r for r in load_set("hsp_dna2dna_low")[1:] if r["id"] == "kat_d2d"][0]
    assert d2d["set"] == [[4, 4, 20, 82, 11], [34, 34, 17, 76, 7]]
    p2d = [r for r in load_set("hsp_protein2dna")[1:] if r["id"] == "kat_p2d"][0]
    assert p2d["set"] == [[5, 15, 12, 49, 6]]
