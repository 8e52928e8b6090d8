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


_CODONS = None


def _codon_table():
    """amino acid -> list of codons (standard genetic code)."""
    global _CODONS
    if _CODONS is None:
        table = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
        _CODONS = {}
        for i, a in enumerate("TCAG"):
            for j, b in enumerate("TCAG"):
                for k, c in enumerate("TCAG"):
                    _CODONS.setdefault(table[i * 16 + j * 4 + k], []).append((a + b + c).encode())
    return _CODONS


def _encode_protein(rng, prot):
    cod = _codon_table()
    return b"".join(cod[chr(a)][int(rng.integers(0, len(cod[chr(a)])))] for a in prot)


def protein_vs_contig(n_proteins, plen=500, contig_len=1000000, seed=20260931, introns=False, mutation=0.05, plant_every=1):
    """C3 (protein2dna) / C5 (protein2genome, introns=True): `n_proteins` proteins against ONE contig that
    holds a codon-encoded, `mutation`-mutated copy of each at a known place.  Returns (proteins, contig,
    [(start, end) of each planted gene]).  plant_every = k: only proteins 0, k, 2k, .. have a gene in the contig (BASELINE
    config 3 names 1 024 proteins of ~500 aa against ONE 1 Mb contig, which 1 024 genes of 1.5 kb do not fit into); the
    places of the others are None."""
    rng = np.random.default_rng([seed, 0])
    proteins = [_rand(rng, plen, AA) for _ in range(n_proteins)]
    planted = [i for i in range(n_proteins) if i % plant_every == 0]
    genes = []
    for p in (proteins[i] for i in planted):
        m = _mutate(rng, p, mutation, AA)
        coding = _encode_protein(rng, m)
        if introns:
            cuts = np.sort(rng.choice(np.arange(30, len(coding) - 30), size=int(rng.integers(1, 4)), replace=False))
            parts, last = [], 0
            for c in list(cuts) + [len(coding)]:
                parts.append(coding[last:c])
                last = c
            g = parts[0]
            for ex in parts[1:]:
                g += b"GT" + _rand(rng, int(rng.integers(100, 3000)) - 4).tobytes() + b"AG" + ex
            coding = g
        genes.append(coding)
    total = sum(len(g) for g in genes)
    gaps = contig_len - total
    assert gaps > 0, "contig too short for the planted genes"
    cutp = np.sort(rng.integers(0, gaps + 1, size=len(genes)))
    out, places, last, pos = [], [], 0, 0
    for g, c in zip(genes, cutp):
        flank = _rand(rng, int(c - last)).tobytes()
        out.append(flank)
        pos += len(flank)
        places.append((pos, pos + len(g)))
        out.append(g)
        pos += len(g)
        last = c
    out.append(_rand(rng, contig_len - pos).tobytes())
    contig = b"".join(out)
    assert len(contig) == contig_len
    if plant_every != 1:
        full = [None] * n_proteins
        for i, pl in zip(planted, places):
            full[i] = pl
        places = full
    return [p.tobytes() for p in proteins], contig, places


def bench_config(name):
    """BASELINE.json's other configurations at their full sizes, as bench.py's `configs` block, tests/test_gpu_configs.py and
    tools/make_bench_golden.py use them: (model name, pairs, place of each pair's planted gene in the shared contig or None).
    c2: affine:local, 4 096 DNA pairs of 1 kb x 1 kb; c3: protein2dna, 1 024 proteins of 500 aa against ONE 1 Mb contig
    (every fourth protein has its gene there); c5: protein2genome, exhaustive, 256 proteins of 300 aa against ONE 10 Mb
    chromosome (every protein's intron-split gene is there)."""
    if name == "c2":
        return "affine:local", affine_dna_pairs(4096, 1000), None
    if name == "c3":
        proteins, contig, places = protein_vs_contig(1024, 500, 1000000, plant_every=4)
        return "protein2dna", [(p, contig) for p in proteins], places
    if name == "c5":
        proteins, contig, places = protein_vs_contig(256, 300, 10000000, seed=20260935, introns=True)
        return "protein2genome", [(p, contig) for p in proteins], places
    raise ValueError("bench_config: c2, c3 or c5")


def write_c5_heuristic_input(directory, n=256, seed=20260935):
    """BASELINE config 5's heuristic leg as FASTA files: n proteins of 300 aa and one 10 Mb chromosome that holds their
    intron-split genes (tools/bench_c5_heuristic.py, tools/make_c5_heuristic_golden.py, bench.py)."""
    import os
    proteins, contig, _ = protein_vs_contig(n, 300, 10000000, seed=seed, introns=True)
    qf, tf = os.path.join(directory, "q.fa"), os.path.join(directory, "t.fa")
    with open(qf, "w") as f:
        for i, p in enumerate(proteins):
            f.write(">p%d\n%s\n" % (i, p.decode()))
    with open(tf, "w") as f:
        f.write(">chr\n%s\n" % contig.decode())
    return qf, tf


def _est2genome_chunk(a):
    first, count, qlen, tlen = a
    return est2genome_pairs(count, qlen, tlen, first=first)


def est2genome_batches(firsts, n_pairs, qlen=1000, tlen=100000, workers=None):
    """Several C4 batches at once -- batch b = pairs [firsts[b], firsts[b] + n_pairs) of the same seeded generator -- made by a
    pool of worker processes (one batch of 4 096 pairs takes one core 6 s; bench.py aligns a fresh batch every step).  The
    workers are forked: call this BEFORE the process opens a HIP device or a process group (bench.py does)."""
    import multiprocessing as mp
    import os
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(32, workers))
    chunk = 64
    tasks = [(f + c, min(chunk, n_pairs - c), qlen, tlen) for f in firsts for c in range(0, n_pairs, chunk)]
    if workers == 1 or len(tasks) <= 2:
        parts = [_est2genome_chunk(t) for t in tasks]
    else:
        with mp.get_context("fork").Pool(min(workers, len(tasks))) as pool:
            parts = pool.map(_est2genome_chunk, tasks, chunksize=1)
    per = (n_pairs + chunk - 1) // chunk
    return [[p for part in parts[b * per:(b + 1) * per] for p in part] for b in range(len(firsts))]


def write_c4_dropin_input(directory, nq=64, nt=64, seed=20260932):
    """BASELINE config 4 through the command line: nq cDNAs of 1 kb and nt genomic windows of 100 kb as FASTA files (window i
    holds cDNA i's gene), aligned all against all by `exonerate -m est2genome -E yes -S no --revcomp no`: nq x nt rectangles
    of 1 001 x 100 001 cells (64 x 64 = 4 096).  tools/make_c4_dropin_golden.py, bench.py's `configs.c4_dropin`."""
    import os
    pairs = est2genome_pairs(max(nq, nt), 1000, 100000, seed=seed)
    qf, tf = os.path.join(directory, "q.fa"), os.path.join(directory, "t.fa")
    with open(qf, "w") as f:
        for i in range(nq):
            f.write(">cdna%d\n%s\n" % (i, pairs[i][0].decode()))
    with open(tf, "w") as f:
        for i in range(nt):
            f.write(">win%d\n%s\n" % (i, pairs[i][1].decode()))
    return qf, tf
