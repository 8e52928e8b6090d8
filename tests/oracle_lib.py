"""ctypes binding of oracle/libc4oracle.so — the CPU restatement used as the CHECKER in tests.

Test infrastructure only (see oracle/c4_oracle.h).  Never imported by exonerate_amd/.
"""
import ctypes as C
import os, subprocess
from exonerate_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libc4oracle.so")


class ViterbiOut(C.Structure):
    _fields_ = [("score", C.c_int32), ("query_start", C.c_int32), ("target_start", C.c_int32),
                ("query_end", C.c_int32), ("target_end", C.c_int32),
                ("final_cell", C.c_int32 * _abi.CELL_MAX), ("last_srp", C.c_int32),
                ("n_ops", C.c_int32), ("ops", C.POINTER(C.c_int32)),
                ("checkpoints", C.POINTER(C.c_int32)), ("cell_size", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("c4_oracle.c", "c4_oracle_sdp.c", "c4_oracle_seed.c")]
        if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(x) for x in srcs):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"],
                                  stdout=subprocess.DEVNULL)
        lib = C.CDLL(SO)
        lib.oracle_splice_predict.argtypes = [C.POINTER(_abi.SpliceModel), C.c_char_p, C.c_int32,
                                              C.POINTER(C.c_int32)]
        lib.oracle_find_score.restype = C.c_int32
        lib.oracle_find_score.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_char_p,
                                          C.c_int32, C.c_char_p, C.c_int32]
        lib.oracle_find_path.restype = C.c_int
        lib.oracle_find_path.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_char_p,
                                         C.c_int32, C.c_char_p, C.c_int32, C.c_int, C.c_int32,
                                         C.POINTER(_abi.Alignment)]
        lib.oracle_alignment_clear.argtypes = [C.POINTER(_abi.Alignment)]
        lib.oracle_viterbi.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_int,
                                       C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                       C.POINTER(_abi.Region), C.POINTER(_abi.Continuation), C.c_int,
                                       C.POINTER(ViterbiOut)]
        lib.oracle_viterbi_out_clear.argtypes = [C.POINTER(ViterbiOut)]
        lib.oracle_use_reduced_space.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Region), C.c_int]
        lib.oracle_checkpoint_rows.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Region), C.c_int]
        lib.oracle_alignment_format.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Alignment), C.c_int,
                                                C.c_char_p, C.c_int32, C.c_char, C.c_char_p, C.c_int32,
                                                C.c_char, C.c_int, C.c_char_p, C.c_size_t]
        lib.oracle_subopt_create.restype = C.c_void_p
        lib.oracle_subopt_create.argtypes = [C.c_int32, C.c_int32]
        lib.oracle_subopt_destroy.argtypes = [C.c_void_p]
        lib.oracle_subopt_add_alignment.argtypes = [C.c_void_p, C.POINTER(_abi.Model), C.POINTER(_abi.Alignment)]
        lib.oracle_subopt_points.restype = C.c_int32
        lib.oracle_subopt_points.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
        lib.oracle_find_path_subopt.restype = C.c_int
        lib.oracle_find_path_subopt.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_char_p,
                                                C.c_int32, C.c_char_p, C.c_int32, C.c_int, C.c_int32,
                                                C.c_void_p, C.POINTER(_abi.Alignment)]
        lib.oracle_find_path_region.restype = C.c_int
        lib.oracle_find_path_region.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_char_p,
                                                C.c_int32, C.c_char_p, C.c_int32, C.POINTER(_abi.Region), C.c_int,
                                                C.c_int32, C.c_void_p, C.POINTER(_abi.Alignment)]
        lib.oracle_hsp_extend.argtypes = [C.POINTER(_abi.Params), C.c_int, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_abi.Hsp)]
        lib.oracle_hsp_set.restype = C.c_int32
        lib.oracle_hsp_set.argtypes = [C.POINTER(_abi.Params), C.c_int, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.c_int32, C.POINTER(_abi.Hsp)]
        lib.oracle_viterbi_subopt.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_int,
                                              C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                              C.POINTER(_abi.Region), C.POINTER(_abi.Continuation), C.c_int,
                                              C.c_void_p, C.POINTER(ViterbiOut)]
        lib.oracle_viterbi_span.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_int,
                                            C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(_abi.Region),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(ViterbiOut)]
        lib.oracle_sdp.restype = C.c_int32
        lib.oracle_sdp.argtypes = [C.POINTER(_abi.Model), C.POINTER(_abi.Params), C.c_char_p, C.c_int32, C.c_char_p,
                                   C.c_int32, C.POINTER(_abi.Hsp), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.POINTER(_abi.Alignment), C.POINTER(C.c_int32)]
        lib.oracle_cells_visited.restype = C.c_int64
        lib.oracle_cells_visited.argtypes = [C.c_int]
        _lib = lib
    return _lib


def alignment_to_dict(model, a, fmt, qid="qy", qlen=0, tlen=0, qstrand=b"+", tstrand=b"+"):
    """fmt(model, alignment, what, ...) is either the oracle's or the product's formatter."""
    out = {"score": a.score, "region": list(a.region.astuple()),
           "ops": [[a.op_transition[i], a.op_length[i]] for i in range(a.n_ops)]}
    buf = C.create_string_buffer(1 << 16)
    for what, key in ((0, "sugar"), (1, "cigar"), (2, "vulgar")):
        n = fmt(model, a, what, qid.encode(), qlen, qstrand, b"tg", tlen, tstrand, 1, buf, len(buf))
        assert n >= 0
        out[key] = buf.value.decode()
    return out


def find_path(model, params, q, t, dpmemory=32, threshold=_abi.IMPOSSIBLY_LOW_SCORE, qid="qy"):
    lib = load()
    a = _abi.Alignment()
    ok = lib.oracle_find_path(model, params, q, len(q), t, len(t), dpmemory, threshold, a)
    if not ok:
        return None
    d = alignment_to_dict(model, a, lib.oracle_alignment_format, qid, len(q), len(t))
    lib.oracle_alignment_clear(a)
    return d


def find_path_region(model, params, q, t, region, dpmemory=32, threshold=_abi.IMPOSSIBLY_LOW_SCORE, qid="qy"):
    """Optimal_find_path over a region of the rectangle (--refine region's call)."""
    lib = load()
    a = _abi.Alignment()
    ok = lib.oracle_find_path_region(model, params, q, len(q), t, len(t), _abi.Region(*region), dpmemory, threshold,
                                     None, a)
    if not ok:
        return None
    d = alignment_to_dict(model, a, lib.oracle_alignment_format, qid, len(q), len(t))
    lib.oracle_alignment_clear(a)
    return d


def find_score(model, params, q, t):
    return load().oracle_find_score(model, params, q, len(q), t, len(t))


def set_annotation(cds):
    """exonerate's --annotation for the calls that follow: cds = (cds_start, cds_length) of the query, or None."""
    lib = load()
    lib.oracle_set_annotation.argtypes = [C.c_int32, C.c_int32]
    lib.oracle_set_annotation.restype = None
    lib.oracle_set_annotation(*(cds if cds else (0, 0)))


def subopt_points(so):
    lib = load()
    n = lib.oracle_subopt_points(so, None, None, 0)
    q, t = (C.c_int32 * max(1, n))(), (C.c_int32 * max(1, n))()
    lib.oracle_subopt_points(so, q, t, n)
    return [[q[i], t[i]] for i in range(n)]


def find_paths_subopt(model, params, q, t, dpmemory, threshold, max_paths, qid="qy"):
    """GAM_Result_exhaustive_create's loop (gam.c:1139-1180): successive best paths, each with the match
    cells of all earlier ones blocked.  Returns [(alignment dict, blocked points after adding it)]."""
    lib = load()
    so = lib.oracle_subopt_create(len(q), len(t))
    out = []
    try:
        for _ in range(max_paths):
            a = _abi.Alignment()
            if not lib.oracle_find_path_subopt(model, params, q, len(q), t, len(t), dpmemory, threshold, so, a):
                break
            d = alignment_to_dict(model, a, lib.oracle_alignment_format, qid, len(q), len(t))
            lib.oracle_subopt_add_alignment(so, model, a)
            lib.oracle_alignment_clear(a)
            out.append((d, subopt_points(so)))
    finally:
        lib.oracle_subopt_destroy(so)
    return out


def span_pair(src_model, dst_model, params, q, t):
    """The two DPs of a span with the identity exchange of tools/make_golden.py (refdump --cmd span): returns
    (src score, reported END cells as {(i, j): cell}, dst score, dst path dict)."""
    lib = load()
    Q, T, cs = len(q), len(t), 1 + src_model.total_shadow_designations
    region = _abi.Region(0, 0, Q, T)
    n = (Q + 1) * (T + 1) * cs
    mat = (C.c_int32 * n)()
    for x in range((Q + 1) * (T + 1)):
        mat[x * cs] = _abi.IMPOSSIBLY_LOW_SCORE
    vo = ViterbiOut()
    lib.oracle_viterbi_span(src_model, params, 0, q, Q, t, T, region, None, mat, vo)
    src_score = vo.score
    lib.oracle_viterbi_out_clear(vo)
    cells = {}
    for i in range(Q + 1):
        for j in range(T + 1):
            x = (i * (T + 1) + j) * cs
            if mat[x] != _abi.IMPOSSIBLY_LOW_SCORE:
                cells[(i, j)] = [mat[x + l] for l in range(cs)]
    vo = ViterbiOut()
    lib.oracle_viterbi_span(dst_model, params, 0, q, Q, t, T, region, mat, None, vo)
    dst_score = vo.score
    lib.oracle_viterbi_out_clear(vo)
    vo = ViterbiOut()
    lib.oracle_viterbi_span(dst_model, params, 1, q, Q, t, T, region, mat, None, vo)
    path = {"score": vo.score, "query_start": vo.query_start, "target_start": vo.target_start,
            "query_end": vo.query_end, "target_end": vo.target_end, "ops": [vo.ops[k] for k in range(vo.n_ops)]}
    lib.oracle_viterbi_out_clear(vo)
    return src_score, cells, dst_score, path


MATCH_TYPES = {"dna2dna": _abi.MATCH_DNA2DNA, "protein2protein": _abi.MATCH_PROTEIN2PROTEIN, "protein2dna": _abi.MATCH_PROTEIN2DNA}


def hsp_extend(params, match, q, t, seedlen, dropoff, qs, ts):
    """HSPset_seed_hsp's HSP of one seed with nothing in its way: [query_start, target_start, length, score, cobs]."""
    h = _abi.Hsp()
    load().oracle_hsp_extend(params, MATCH_TYPES[match], q, len(q), t, len(t), seedlen, dropoff, qs, ts, h)
    return h.aslist()


def hsp_set(params, match, q, t, seedlen, dropoff, threshold, seeds):
    """One HSPset fed the seeds in order: the list of HSPs it holds after HSPset_finalise."""
    n = len(seeds)
    sq = (C.c_int32 * max(1, n))(*[s[0] for s in seeds])
    st = (C.c_int32 * max(1, n))(*[s[1] for s in seeds])
    out = (_abi.Hsp * max(1, n))()
    k = load().oracle_hsp_set(params, MATCH_TYPES[match], q, len(q), t, len(t), seedlen, dropoff, threshold, sq, st, n, out)
    return [out[i].aslist() for i in range(k)]


def sdp(model, params, q, t, hsps, query_advance=1, target_advance=1, dropoff=50, singlepass=True, threshold=100,
        max_alignments=8, qid="qy"):
    """The loop of GAM_Result_SDP_create (gam.c:852) on one pair's HSPs ([query_start, target_start, length, score,
    cobs] each, in HSPset order): (use_boundary, [alignment dicts with score / region / ops / vulgar])."""
    lib = load()
    n = len(hsps)
    hs = (_abi.Hsp * max(1, n))(*[_abi.Hsp(*h) for h in hsps])
    out = (_abi.Alignment * max_alignments)()
    ub = C.c_int32(-1)
    k = lib.oracle_sdp(model, params, q, len(q), t, len(t), hs, n, query_advance, target_advance, dropoff,
                       1 if singlepass else 0, threshold, max_alignments, out, C.byref(ub))
    res = []
    for i in range(k):
        res.append(alignment_to_dict(model, out[i], lib.oracle_alignment_format, qid, len(q), len(t)))
        lib.oracle_alignment_clear(out[i])
    return ub.value, res


def seed_walk(rec):
    """The seeder's walk over one reference-generated record (tests/golden/seeds_*.jsonl): the (query, query position,
    target position) triples in the reference's call order (oracle_seed_walk, oracle/c4_oracle_seed.c)."""
    import numpy as np
    lib = load()
    words = sorted(range(len(rec["words"])), key=lambda w: rec["words"][w][0])           # ascending codes
    place = {w: k for k, w in enumerate(words)}
    codes = np.array([rec["words"][w][0] for w in words], dtype=np.uint64)
    seed_first, seeds, nbr_first, nbrs = [0], [], [0], []
    for w in words:
        _, own, nb = rec["words"][w]
        for q, p in own:
            seeds += [q, p]
        seed_first.append(len(seeds) // 2)
        nbrs += [place[v] for v in nb]
        nbr_first.append(len(nbrs))
    sf, sd = np.array(seed_first, dtype=np.int32), np.array(seeds + [0], dtype=np.int32)
    nf, nb_ = np.array(nbr_first, dtype=np.int32), np.array(nbrs + [0], dtype=np.int32)
    sym = np.array(rec["symbols"] + [0], dtype=np.uint8)
    lib.oracle_seed_walk.restype = C.c_int64
    ptr = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    args = [C.c_int32(rec["width"]), C.c_int32(rec["wordlen"]), C.c_int32(len(words)), ptr(codes, C.c_uint64), ptr(sf, C.c_int32),
            ptr(sd, C.c_int32), ptr(nf, C.c_int32), ptr(nb_, C.c_int32), ptr(sym, C.c_uint8), C.c_int32(len(rec["symbols"])),
            C.c_int32(rec["tpos_modifier"])]
    n = lib.oracle_seed_walk(*args, None, C.c_int64(0))
    out = np.zeros(3 * max(1, n), dtype=np.int32)
    got = lib.oracle_seed_walk(*args, ptr(out, C.c_int32), C.c_int64(n))
    assert got == n
    return out[:3 * n].reshape(n, 3).tolist()
