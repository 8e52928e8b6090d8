"""Seeded synthetic inputs of BASELINE.json's configs.  numpy only; alphabets are
restricted to ACGT / the 20 amino acids so that every residue is inside exonerate's Submat index."""
import numpy as np

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
AA = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)


def _rand(rng, n, alpha=DNA):
    return alpha[rng.integers(0, len(alpha), size=n)]


def _mutate(rng, seq, rate, alpha=DNA):
    """`rate` of the positions hit, one third each substitution / insertion / deletion."""
    r = rng.random(len(seq))
    out = []
    last = 0
    for p in np.nonzero(r < rate)[0]:
        out.append(seq[last:p])
        kind = r[p] * 3 / rate
        if kind < 1:
            out.append(_rand(rng, 1, alpha))
        elif kind < 2:
            out.append(seq[p:p + 1])
            out.append(_rand(rng, 1, alpha))
        last = p + 1
    out.append(seq[last:])
    return np.concatenate(out) if out else seq


def est2genome_pairs(n_pairs, qlen=1000, tlen=100000, seed=20260932, first=0):
    """C4 (north star): cDNA of `qlen`; genomic window of exactly `tlen` = flank + exons split at 3-6
    points by GT...AG introns of U[100,5000] nt + 3 % mutation of the exons + flank."""
    pairs = []
    for k in range(first, first + n_pairs):
        rng = np.random.default_rng([seed, k])
        q = _rand(rng, qlen)
        ncut = int(rng.integers(3, 7))
        cuts = np.sort(rng.choice(np.arange(30, qlen - 30), size=ncut, replace=False))
        pieces = []
        last = 0
        for c in list(cuts) + [qlen]:
            pieces.append(_mutate(rng, q[last:c], 0.03))
            if c != qlen:
                ilen = int(rng.integers(100, 5001))
                pieces.append(np.concatenate([np.frombuffer(b"GT", np.uint8), _rand(rng, ilen - 4),
                                              np.frombuffer(b"AG", np.uint8)]))
            last = c
        gene = np.concatenate(pieces)
        flank = tlen - len(gene)
        left = int(rng.integers(0, flank + 1))
        t = np.concatenate([_rand(rng, left), gene, _rand(rng, flank - left)])
        pairs.append((q.tobytes(), t.tobytes()))
    return pairs


def affine_dna_pairs(n_pairs, qlen=1000, seed=20260930, first=0):
    """C2: target = query with 10 % substitution/insertion/deletion."""
    pairs = []
    for k in range(first, first + n_pairs):
        rng = np.random.default_rng([seed, k])
        q = _rand(rng, qlen)
        pairs.append((q.tobytes(), _mutate(rng, q, 0.10).tobytes()))
    return pairs
