"""The oracle's restatement of the seeder's automaton walk (oracle/c4_oracle_seed.c: Seeder_add_target seeder.c:852-915 ->
FSM_traverse fsm.c:186-198 / Seeder_VFSM_traverse_single seeder.c:698-720 -> Seeder_FSM_traverse_func :649-695) against what the
reference itself did: tests/golden/seeds_*.jsonl hold, per seeder, the words its queries put into the reference's automaton
(own seeds and neighbour words in list order, read off the trie / the VFSM leaf table), the target as automaton columns after
the reference's masking, and every HSPset_seed_hsp call the reference's own walk made, in order (oracle/refdump.c --cmd seeds,
tools/make_golden.py): DNA words of 12 and 9, the compact automaton (--forcefsm compact), protein words with neighbourhoods."""
import json
import os

import pytest

import oracle_lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED_SETS = sorted(f[:-6] for f in os.listdir(GOLDEN) if f.startswith("seeds_") and f.endswith(".jsonl"))


def load_seed_set(name):
    with open(os.path.join(GOLDEN, name + ".jsonl")) as f:
        return [json.loads(l) for l in f if l.strip()]


def test_the_sets_cover_both_automata_and_neighbourhoods():
    assert len(SEED_SETS) >= 6
    kinds = {(r["automaton"], bool(sum(len(w[2]) for w in r["words"]))) for n in SEED_SETS for r in load_seed_set(n)}
    assert ("fsm", False) in kinds and ("vfsm", False) in kinds and ("fsm", True) in kinds and ("vfsm", True) in kinds
    # words whose seed list holds seeds of several queries, residues outside the alphabet in the targets
    assert any(len({q for q, _ in w[1]}) > 1 for r in load_seed_set("seeds_dna2dna") for w in r["words"])
    assert any(0 in r["symbols"] for r in load_seed_set("seeds_dna2dna"))


@pytest.mark.parametrize("name", SEED_SETS)
def test_seed_walk_matches_the_reference_walk(name):
    total = 0
    for rec in load_seed_set(name):
        assert oracle_lib.seed_walk(rec) == rec["expected"], rec["id"]
        total += len(rec["expected"])
    assert total > 0
