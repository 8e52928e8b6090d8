"""The sparse SDP wavefront (exonerate_amd/csrc/c4_sdp_wave.h: per-lane cell function, event-driven strips, carried rows,
mirrored boundary records, span stores, order keys, the walk; c4_sdp_host.h) driven by CPU loops (tests/sdp_sim.hip)
against the reference's own SDP alignments (tests/golden/sdp_*.jsonl) and against the pinned oracle on seeded random pairs
with several strips, several seeds per diagonal, introns, frameshifts and lowered thresholds.  No GPU involved: this
checks the algorithm the kernels implement; tests/test_gpu_sdp.py checks the kernels themselves."""
import random
import pytest

import exonerate_amd as ex
import oracle_lib
import sdp_cases
import sdp_sim_lib
from test_oracle_sdp import SDP_SETS, sdp_case, expected

FAMILY_OF = {"affine:local": "affine", "est2genome": "est2genome", "protein2dna": "protein2dna", "protein2genome": "protein2genome"}
SINGLEPASS = [n for n in sorted(SDP_SETS) if n != "sdp_est2genome_multipass"]


@pytest.mark.parametrize("name", SINGLEPASS)
def test_simulated_wavefront_matches_reference_vectors(name):
    model, par, recs, adv = sdp_case(name)
    if not par["singlepass"]:
        pytest.skip("multipass")
    fam = FAMILY_OF[SDP_SETS[name][0]]
    total = 0
    for r in recs:
        got, steps = sdp_sim_lib.sdp(model.c, model.params, fam, r["query"].encode(), r["target"].encode(), r["hsps"], adv[0], adv[1],
                                     par["dropoff"], par["threshold"], 4, qid=r["id"])
        got = [{k: a[k] for k in ("score", "region", "ops", "vulgar")} for a in got]
        assert got == expected(r), r["id"]
        total += len(got)
    assert total >= 10


def _check_cases(cases):
    n = steps_total = 0
    for cs in cases:
        model, adv = cs["model"], cs["adv"]
        fam = sdp_cases.FAMILY[cs["kind"]]
        for (q, t), h in zip(cs["pairs"], cs["hsps"]):
            ub, exp = oracle_lib.sdp(model.c, model.params, q.encode(), t.encode(), h, adv[0], adv[1], cs["dropoff"], True, cs["threshold"], 4)
            got, steps = sdp_sim_lib.sdp(model.c, model.params, fam, q.encode(), t.encode(), h, adv[0], adv[1], cs["dropoff"], cs["threshold"], 4)
            assert got == exp, (cs["kind"], cs["variant"], cs["dropoff"], cs["threshold"], len(q), len(t), len(h))
            n += len(exp)
            steps_total += steps
    return n, steps_total


@pytest.mark.parametrize("seed", range(4))
def test_simulated_wavefront_seeded_fuzz(seed):
    n, _ = _check_cases(sdp_cases.seeded_fuzz_cases(seed, rounds=2, pairs_per_round=6))
    assert n >= 2


@pytest.mark.parametrize("seed", range(4))
def test_simulated_wavefront_boundary_fuzz(seed):
    n, _ = _check_cases(sdp_cases.boundary_fuzz_cases(seed, rounds=2, pairs_per_round=5))
    assert n >= 2
