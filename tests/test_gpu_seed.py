"""The seeder's word scan on the device (c4gpu_seed_scan <-> Seeder_add_target's automaton walk, seeder.c:649-720,852-915)
against what the reference's own walk did (tests/golden/seeds_*.jsonl: oracle/refdump.c --cmd seeds; the oracle's restatement is
pinned on the same records in tests/test_oracle_seed.py) and, at sizes no reference vector has, against a plain dictionary scan: every position whose last W symbols spell a word of the table, in position order, each
with the word's emissions in list order; symbols outside the alphabet (column 0) reset the automaton.  The drop-in's use
of it against the reference's own traversal, hit for hit: tests/test_integration_gpu.py (C4GPU_SEED_CHECK)."""
import random
import pytest

import exonerate_amd as ex
import oracle_lib
from test_oracle_seed import SEED_SETS, load_seed_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = ex.Engine(0)
    yield e
    e.close()


def _expected(width, wordlen, words, symbols):
    first, table = 0, {}
    for code, n in words:
        table[code] = (first, n)
        first += n
    out = []
    for i in range(wordlen - 1, len(symbols)):
        w = symbols[i - wordlen + 1:i + 1]
        if 0 in w:
            continue
        code = 0
        for s in w:
            code = code * width + s
        if code in table:
            f, n = table[code]
            out += [(i, f + k) for k in range(n)]
    return out


@pytest.mark.parametrize("name", SEED_SETS)
def test_scan_reproduces_the_reference_walk(eng, name):
    """Every reference-generated seeder record: the word table as the drop-in builds it (a word's emission list = its own seeds,
    then the seeds of each neighbour word: Seeder_FSM_traverse_func, seeder.c:675-692) goes to the device, the hits come back in
    walk order, and (query, query position, position - tpos_modifier) of every hit is the reference's HSPset_seed_hsp call,
    call for call; the oracle's walk gives the same list."""
    for rec in load_seed_set(name):
        words, emits = [], []
        for code, own, nbrs in rec["words"]:
            mine = list(own)
            for v in nbrs:
                mine += rec["words"][v][1]
            words.append((code, len(mine)))
            emits += mine
        hits, _ = eng.seed_scan(rec["width"], rec["wordlen"], words, bytes(rec["symbols"]))
        got = [[emits[e][0], emits[e][1], pos - rec["tpos_modifier"]] for pos, e in hits]
        assert got == rec["expected"], rec["id"]
        assert got == oracle_lib.seed_walk(rec), rec["id"]


@pytest.mark.parametrize("width,wordlen,n,n_words,seed", [
    (5, 12, 200000, 3000, 1),       # DNA words of 12 (the dna2dna default), a target with N runs
    (5, 3, 5000, 60, 2),            # short words: most positions hit, several emissions each
    (25, 5, 120000, 20000, 3),      # protein words with neighbourhoods: many words, many emissions
    (21, 6, 70000, 10, 4),          # almost nothing hits
    (5, 12, 11, 4, 5), (5, 12, 12, 4, 6), (5, 4, 1, 1, 7),      # targets shorter than / as long as a word
])
def test_scan_matches_dictionary(eng, width, wordlen, n, n_words, seed):
    rng = random.Random(seed)
    symbols = bytearray(rng.randint(1, width - 1) for _ in range(n))
    for _ in range(n // 997):                                   # runs of symbols outside the alphabet
        p = rng.randrange(n)
        for k in range(p, min(n, p + rng.randint(1, 30))):
            symbols[k] = 0
    words, seen = [], set()
    while len(words) < n_words:
        if n >= wordlen and rng.random() < 0.7:                 # a word of the target itself
            p = rng.randrange(n - wordlen + 1)
            w = symbols[p:p + wordlen]
            if 0 in w:
                continue
        else:
            w = bytes(rng.randint(1, width - 1) for _ in range(wordlen))
        code = 0
        for s in w:
            code = code * width + s
        if code in seen:
            continue
        seen.add(code)
        words.append((code, rng.choice([1, 1, 1, 2, 3, 17])))
    got, ms = eng.seed_scan(width, wordlen, words, bytes(symbols))
    assert got == _expected(width, wordlen, words, symbols)
    if n >= 100000:
        assert len(got) > 0


def test_scan_of_a_megabase_target_in_order(eng):
    """10 Mb of symbols against 50 000 words: the hit list is sorted by position and complete (count against the dictionary)."""
    rng = random.Random(9)
    n, width, wordlen = 10000000, 5, 12
    import numpy as np
    sym = np.random.default_rng(9).integers(1, width, size=n, dtype=np.uint8)
    symbols = sym.tobytes()
    words, seen = [], set()
    while len(words) < 50000:
        p = rng.randrange(n - wordlen)
        code = 0
        for s in symbols[p:p + wordlen]:
            code = code * width + s
        if code not in seen:
            seen.add(code)
            words.append((code, 1 + (len(words) % 3)))
    got, ms = eng.seed_scan(width, wordlen, words, symbols)
    pos = [p for p, _ in got]
    assert pos == sorted(pos) and len(got) >= 50000
    # spot check: a window of the target against the dictionary
    lo, hi = 4000000, 4200000
    exp = _expected(width, wordlen, words, symbols[lo - wordlen + 1:hi])
    assert [(p, e) for p, e in got if lo <= p < hi] == [(p + lo - wordlen + 1, e) for p, e in exp if p + lo - wordlen + 1 >= lo]
    print("scan of %d symbols: %.3f ms on the device, %d hits" % (n, ms, len(got)))
