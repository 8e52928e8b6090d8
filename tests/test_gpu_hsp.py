"""HSP seeding on the device (c4gpu_hsp_extend_batch: the ungapped X-drop extension of HSPset_seed_hsp,
src/comparison/hspset.c:933) against the reference's own HSPs (tests/golden/hsp_*.jsonl) and, at north-star size, against
the oracle (pinned on the same vectors in test_oracle_hsp.py)."""
import random
import pytest

import exonerate_amd as ex
from exonerate_amd import workloads
import oracle_lib
from golden_util import load_set
from test_oracle_hsp import HSP_SETS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name", HSP_SETS)
def test_device_hsps_match_reference_vectors(eng, name):
    recs = load_set(name)
    par, recs = recs[0]["params"], recs[1:]
    params = ex.default_params()
    pairs = [(r["query"], r["target"]) for r in recs]
    seeds = [(k, qs, ts) for k, r in enumerate(recs) for qs, ts in r["seeds"]]
    got = eng.hsp_extend(params, par["match"], pairs, par["seedlen"], par["dropoff"], seeds)
    exp = [e for r in recs for e in r["single"]]
    assert len(got) == len(exp) > 500
    for (k, qs, ts), g, e in zip(seeds, got, exp):
        if e is None:                              # the reference stored nothing: the HSP is below its threshold
            assert g[3] < par["threshold"], (recs[k]["id"], qs, ts)
        else:
            assert g == e, (recs[k]["id"], qs, ts)
    # the whole-set lists follow from the per-seed HSPs by the horizon rule (hspset.c:952-958,993-995)
    at = par["target_advance"]
    pos = 0
    for r in recs:
        horizon, kept = {}, []
        for qs, ts in r["seeds"]:
            g = got[pos]
            pos += 1
            key = ((ts - qs * at + len(r["query"])) % len(r["query"]), ts % at)
            if ts < horizon.get(key, 0):
                continue
            if g[3] >= par["threshold"]:
                kept.append(g)
            horizon[key] = g[1] + g[2] * at
        assert kept == r["set"], r["id"]


@pytest.mark.parametrize("name", HSP_SETS)
def test_horizon_chains_on_the_device_give_the_reference_sets(eng, name):
    """c4gpu_hsp_extend_chains: one lane per horizon entry takes its seeds in order and does not extend the ones below the
    running horizon (hspset.c:952-958,990).  The HSPs it does extend, kept by the threshold, are the reference's whole-set
    lists; every extended seed equals the unconditional extension; a chain that starts from a non-zero horizon skips what
    the reference would skip."""
    recs = load_set(name)
    par, recs = recs[0]["params"], recs[1:]
    params = ex.default_params()
    at = par["target_advance"]
    pairs = [(r["query"], r["target"]) for r in recs]
    seeds, chain, keys = [], [], {}
    for k, r in enumerate(recs):
        for qs, ts in r["seeds"]:
            key = (k, (ts - qs * at + len(r["query"])) % len(r["query"]), ts % at)
            chain.append(keys.setdefault(key, len(keys)))
            seeds.append((k, qs, ts))
    plain = eng.hsp_extend(params, par["match"], pairs, par["seedlen"], par["dropoff"], seeds)
    got = eng.hsp_extend_chains(params, par["match"], pairs, par["seedlen"], par["dropoff"], seeds, chain, [0] * len(keys))
    skipped = 0
    kept = [[] for _ in recs]
    for (k, qs, ts), g, p in zip(seeds, got, plain):
        if g[2] < 0:
            skipped += 1
            continue
        assert g == p
        if g[3] >= par["threshold"]:
            kept[k].append(g)
    assert skipped > 0
    for r, ks in zip(recs, kept):
        assert ks == r["set"], r["id"]
    # a horizon that is already far to the right: nothing of those chains is extended
    far = [10 ** 9 if c % 2 else 0 for c in range(len(keys))]
    got2 = eng.hsp_extend_chains(params, par["match"], pairs, par["seedlen"], par["dropoff"], seeds, chain, far)
    for c, g, g0 in zip(chain, got2, got):
        assert (g[2] < 0) if c % 2 else (g == g0)


def test_all_word_hits_of_north_star_pairs(eng):
    """1 kb cDNAs against 100 kb windows: every shared 12-mer of every pair as a seed (tens of thousands per launch),
    against the oracle."""
    params = ex.default_params()
    pairs = [(q.decode(), t.decode()) for q, t in workloads.est2genome_pairs(6, 1000, 100000)]
    seeds = []
    for k, (q, t) in enumerate(pairs):
        words = {}
        for i in range(len(q) - 11):
            words.setdefault(q[i:i + 12], []).append(i)
        for j in range(len(t) - 11):
            for i in words.get(t[j:j + 12], ()):
                seeds.append((k, i, j))
    assert len(seeds) > 3000
    got = eng.hsp_extend(params, "dna2dna", pairs, 12, 30, seeds)
    rng = random.Random(3)
    for x in rng.sample(range(len(seeds)), 1500):
        k, i, j = seeds[x]
        assert got[x] == oracle_lib.hsp_extend(params, "dna2dna", pairs[k][0].encode(), pairs[k][1].encode(), 12, 30, i, j)
    # HSPs of a cDNA's exons: long, high-scoring
    assert max(g[3] for g in got) > 500


def test_seed_outside_its_pair_is_rejected(eng):
    with pytest.raises(ex.C4GpuError):
        eng.hsp_extend(ex.default_params(), "dna2dna", [("ACGTACGTACGTACGT", "ACGTACGTACGTACGT")], 12, 30, [(0, 8, 0)])
