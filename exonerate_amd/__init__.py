"""exonerate_amd — thin Python plumbing over libc4gpu.so (the MI355X-native C4 Viterbi engine).

The product is the C-ABI shared library built from exonerate_amd/csrc (HIP kernels for gfx950 + host
orchestration).  This package only marshals arguments: it never computes an alignment itself and it
raises when the library or the GPU is missing — there is no Python / CPU fallback.

Names follow the reference (exonerate src/c4/optimal.h:47-66, src/model/modeltype.h):
    model = Model("est2genome")                   Model_Type_get_model
    eng   = Engine()                              one HIP device
    eng.find_score(model, pairs)                  Optimal_find_score   per pair
    eng.find_path(model, pairs, dpmemory=32)      Optimal_find_path    per pair -> Alignment
    aln.vulgar(...), aln.cigar(...), aln.sugar(...)  Alignment_display_{vulgar,cigar,sugar}
"""
import ctypes as C

from . import _abi
from ._abi import (ALPHABET_DNA, ALPHABET_PROTEIN, IMPOSSIBLY_LOW_SCORE, MODE_FIND_SCORE, MODE_FIND_PATH,
                   MODE_FIND_REGION, MODE_FIND_CHECKPOINTS)

__all__ = ["Model", "Engine", "Alignment", "ResidentBatch", "Stage", "default_params", "C4GpuError"]


class C4GpuError(RuntimeError):
    pass


def _lib():
    return _abi.load()


def _err(prefix):
    return C4GpuError("%s: %s" % (prefix, (_lib().c4gpu_last_error() or b"").decode()))


def default_params():
    p = _abi.Params()
    _lib().c4gpu_params_default(p)
    return p


class Model:
    """A closed C4 model flattened to tables (c4gpu_model)."""
    _ALPHABETS = {"est2genome": (0, 0), "protein2dna": (1, 0), "protein2dna:bestfit": (1, 0),
                  "protein2genome": (1, 0), "protein2genome:bestfit": (1, 0)}

    def __init__(self, model_type, query_alphabet=None, target_alphabet=None, params=None):
        self.params = params if params is not None else default_params()
        qa, ta = self._ALPHABETS.get(model_type, (ALPHABET_DNA, ALPHABET_DNA))
        qa = qa if query_alphabet is None else query_alphabet
        ta = ta if target_alphabet is None else target_alphabet
        self.c = _abi.Model()
        if _lib().c4gpu_model_get(model_type.encode(), qa, ta, self.params, self.c) != 0:
            raise C4GpuError("unknown or unsupported model type %r" % model_type)
        self.model_type = model_type

    @classmethod
    def derived(cls, model_type, src_state, dst_state, start_scope, end_scope, query_alphabet=None,
                target_alphabet=None, params=None):
        """C4_DerivedModel_create on a model type: BSDP's join / terminal models (c4gpu_model_get_derived)."""
        self = cls.__new__(cls)
        self.params = params if params is not None else default_params()
        qa, ta = cls._ALPHABETS.get(model_type, (ALPHABET_DNA, ALPHABET_DNA))
        qa = qa if query_alphabet is None else query_alphabet
        ta = ta if target_alphabet is None else target_alphabet
        self.c = _abi.Model()
        self.transition_map = (C.c_int32 * _abi.MAX_TRANSITIONS)()
        if _lib().c4gpu_model_get_derived(model_type.encode(), qa, ta, self.params, src_state, dst_state, start_scope,
                                          end_scope, self.c, self.transition_map) != 0:
            raise C4GpuError("no derived model %r %d -> %d" % (model_type, src_state, dst_state))
        self.model_type = model_type
        return self

    @property
    def name(self):
        return self.c.name.decode()

    def plugin_name(self, mode, continuation=False):
        buf = C.create_string_buffer(256)
        _lib().c4gpu_model_plugin_name(self.c, mode, int(continuation), buf, 256)
        return buf.value.decode()


class Alignment:
    def __init__(self, model, c_alignment, qlen, tlen):
        self.model = model
        self.score = c_alignment.score
        self.region = c_alignment.region.astuple()
        self.ops = [(c_alignment.op_transition[i], c_alignment.op_length[i]) for i in range(c_alignment.n_ops)]
        self.qlen, self.tlen = qlen, tlen

    @classmethod
    def from_parts(cls, model, score, region, ops, qlen, tlen):
        """An alignment from its numbers (what exonerate_amd.parallel gathers from the other ranks)."""
        self = cls.__new__(cls)
        self.model, self.score, self.region = model, score, tuple(region)
        self.ops = [(int(t), int(l)) for t, l in ops]
        self.qlen, self.tlen = qlen, tlen
        return self

    def _c(self):
        a = _abi.Alignment()
        a.score = self.score
        a.region = _abi.Region(*self.region)
        a.n_ops = len(self.ops)
        self._t = (C.c_int32 * max(1, len(self.ops)))(*[o[0] for o in self.ops])
        self._l = (C.c_int32 * max(1, len(self.ops)))(*[o[1] for o in self.ops])
        a.op_transition = C.cast(self._t, C.POINTER(C.c_int32))
        a.op_length = C.cast(self._l, C.POINTER(C.c_int32))
        a.valid = 1
        return a

    def _format(self, what, qid, tid, qstrand, tstrand, forward_coords):
        buf = C.create_string_buffer(64 + 24 * (len(self.ops) + 4) + len(qid) + len(tid))
        n = _lib().c4gpu_alignment_format(self.model.c, self._c(), what, qid.encode(), self.qlen,
                                          qstrand.encode(), tid.encode(), self.tlen, tstrand.encode(),
                                          int(forward_coords), buf, len(buf))
        if n < 0:
            raise C4GpuError("alignment line does not fit its buffer")
        return buf.value.decode()

    def sugar(self, qid="qy", tid="tg", qstrand="+", tstrand="+", forward_coords=True):
        return self._format(0, qid, tid, qstrand, tstrand, forward_coords)

    def cigar(self, qid="qy", tid="tg", qstrand="+", tstrand="+", forward_coords=True):
        return self._format(1, qid, tid, qstrand, tstrand, forward_coords)

    def vulgar(self, qid="qy", tid="tg", qstrand="+", tstrand="+", forward_coords=True):
        return self._format(2, qid, tid, qstrand, tstrand, forward_coords)

    def gff(self, query, target, qid="qy", tid="tg", qstrand="+", tstrand="+", on_query=False, genomic=None, result_id=1,
            date=None, version=None):
        """Alignment_display_gff (--showtargetgff / --showquerygff with on_query) as the reference prints it
        (c4gpu_alignment_format_gff).  genomic: gene features before the similarity line; default: a target report of a
        model with introns (Model_Type_has_genomic_target, gam.c:1229)."""
        if genomic is None:
            labels = {self.model.c.transitions[k].label for k in range(self.model.c.n_transitions)}
            genomic = (not on_query) and (_abi.LABEL_INTRON in labels)
        q = query if isinstance(query, bytes) else query.encode()
        t = target if isinstance(target, bytes) else target.encode()
        req = _abi.GffRequest(qid.encode(), tid.encode(), q, t, len(q), len(t), qstrand.encode(), tstrand.encode(), int(on_query),
                              int(genomic), result_id, date.encode() if date else None, version.encode() if version else None)
        cap = 4096 + 64 * (len(self.ops) + 8)
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            n = _lib().c4gpu_alignment_format_gff(self.model.c, self.model.params, self._c(), req, buf, cap)
            if n >= 0:
                return buf.value.decode()
            if n == -(2 ** 31):
                raise C4GpuError("no GFF dump for this model")
            cap = -n
        raise C4GpuError("GFF dump does not fit its buffer")

    def display(self, query, target, qid="qy", tid="tg", qstrand="+", tstrand="+", qdef=None, tdef=None, width=80,
                forward_coords=True, use_aa_tla=True):
        """Alignment_display: the human-readable block of --showalignment yes as the reference prints it
        (c4gpu_alignment_display)."""
        q = query if isinstance(query, bytes) else query.encode()
        t = target if isinstance(target, bytes) else target.encode()
        req = _abi.DisplayRequest(qid.encode(), qdef.encode() if qdef else None, tid.encode(), tdef.encode() if tdef else None,
                                  q, t, len(q), len(t), qstrand.encode(), tstrand.encode(), width, int(forward_coords),
                                  int(use_aa_tla))
        cap = 8192 + 8 * (self.region[2] + self.region[3]) + 256 * len(self.ops)
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            n = _lib().c4gpu_alignment_display(self.model.c, self.model.params, self._c(), req, buf, cap)
            if n >= 0:
                return buf.value.decode()
            if n == -(2 ** 31):
                raise C4GpuError("no alignment display for this model / alignment")
            cap = -n
        raise C4GpuError("alignment display does not fit its buffer")

    def ryo(self, fmt, query, target, qid="qy", tid="tg", qstrand="+", tstrand="+", qdef=None, tdef=None, rank=0,
            forward_coords=True):
        """Alignment_display_ryo: a --ryo format string printed for this alignment as the reference prints it
        (c4gpu_alignment_format_ryo)."""
        q = query if isinstance(query, bytes) else query.encode()
        t = target if isinstance(target, bytes) else target.encode()
        req = _abi.RyoRequest(qid.encode(), qdef.encode() if qdef else None, tid.encode(), tdef.encode() if tdef else None, q, t,
                              len(q), len(t), qstrand.encode(), tstrand.encode(), int(forward_coords), rank, fmt.encode())
        cap = 1 << 16
        for _ in range(2):
            buf = C.create_string_buffer(cap)
            n = _lib().c4gpu_alignment_format_ryo(self.model.c, self.model.params, self._c(), req, buf, cap)
            if n >= 0:
                return buf.raw[:n].decode()
            if n == -(2 ** 31):
                raise C4GpuError("this --ryo string cannot be printed for this model (unknown token, unbalanced braces, %pS on a codon match)")
            cap = -n
        raise C4GpuError("ryo output does not fit its buffer")

    def as_dict(self, qid="qy"):
        return {"score": self.score, "region": list(self.region), "ops": [list(o) for o in self.ops],
                "sugar": self.sugar(qid), "cigar": self.cigar(qid), "vulgar": self.vulgar(qid)}


class SubOpt:
    """c4gpu_subopt: the cells blocked by the alignments already reported for one pair (SubOpt, subopt.h)."""

    def __init__(self, query_length, target_length):
        self.h = _lib().c4gpu_subopt_create(query_length, target_length)

    def add_alignment(self, alignment):
        if _lib().c4gpu_subopt_add_alignment(self.h, alignment.model.c, alignment._c()) != 0:
            raise _err("c4gpu_subopt_add_alignment")

    def add_point(self, query_pos, target_pos):
        _lib().c4gpu_subopt_add_point(self.h, query_pos, target_pos)

    def points(self):
        n = _lib().c4gpu_subopt_points(self.h, None, None, 0)
        q, t = (C.c_int32 * max(1, n))(), (C.c_int32 * max(1, n))()
        _lib().c4gpu_subopt_points(self.h, q, t, n)
        return [[q[i], t[i]] for i in range(n)]

    def close(self):
        if self.h:
            _lib().c4gpu_subopt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _pairs(pairs):
    arr = (_abi.Pair * max(1, len(pairs)))()
    keep = []
    seen = {}                      # one object handed over in many pairs (one contig against many queries) is one buffer:
    #                                the library uploads each buffer once.  Keyed by identity: hashing 400 MB of residues to
    #                                find equal copies would cost more than uploading them
    def buf(x):
        got = seen.get(id(x))
        if got is None:
            got = seen[id(x)] = x if isinstance(x, bytes) else x.encode()
        return got
    for i, (q, t) in enumerate(pairs):
        q, t = buf(q), buf(t)
        keep.append((q, t))
        arr[i].query, arr[i].query_len, arr[i].target, arr[i].target_len = q, len(q), t, len(t)
    return arr, keep


def _viterbi_jobs(jobs):
    cj = (_abi.ViterbiJob * max(1, len(jobs)))()
    for i, j in enumerate(jobs):
        cj[i].pair = j["pair"]
        cj[i].region = _abi.Region(*j["region"])
        cont = j.get("continuation")
        cj[i].use_continuation = 1 if cont else 0
        if cont:
            cj[i].continuation.first_state = cont["first_state"]
            cj[i].continuation.final_state = cont["final_state"]
            for l, v in enumerate(cont.get("first_cell", [])):
                cj[i].continuation.first_cell[l] = v
        cj[i].checkpoint_count = j.get("checkpoints", 0)
        cj[i].subopt = j["subopt"].h if j.get("subopt") is not None else None
        # span models: ctypes int32 arrays over the region, [(i * (T+1)) + j][1 + designations]
        if j.get("start_cells") is not None:
            cj[i].start_cells = C.cast(j["start_cells"], C.POINTER(C.c_int32))
        if j.get("end_cells") is not None:
            cj[i].end_cells = C.cast(j["end_cells"], C.POINTER(C.c_int32))
    return cj


def _viterbi_results(res, n):
    out = []
    for i in range(n):
        r = res[i]
        out.append({"score": r.score, "query_start": r.query_start, "target_start": r.target_start,
                    "query_end": r.query_end, "target_end": r.target_end,
                    "final_cell": list(r.final_cell), "last_srp": r.last_srp,
                    "ops": [r.ops[k] for k in range(r.n_ops)]})
        _lib().c4gpu_viterbi_result_clear(r)
    return out


class Engine:
    """One HIP device (c4gpu_ctx).  Raises when there is no gfx950 GPU: nothing here runs on the CPU."""

    def __init__(self, device=0, stream=None):
        self.ctx = _lib().c4gpu_ctx_create(device)
        if not self.ctx:
            raise _err("c4gpu_ctx_create")
        if stream is not None:
            _lib().c4gpu_ctx_set_stream(self.ctx, C.c_void_p(stream))

    def close(self):
        if self.ctx:
            _lib().c4gpu_ctx_destroy(self.ctx)
            self.ctx = None

    def own_stream(self):
        """A non-blocking HIP stream of this context's own (c4gpu_ctx_own_stream): calls made on it from a second host thread
        run beside another context's instead of in the default stream's order."""
        if _lib().c4gpu_ctx_own_stream(self.ctx) != 0:
            raise _err("c4gpu_ctx_own_stream")

    def sdp_reserve(self, nbytes):
        """Takes the SDP record arena now and keeps it between batches (c4gpu_ctx_sdp_reserve); nbytes <= 0 gives it back."""
        if _lib().c4gpu_ctx_sdp_reserve(self.ctx, int(nbytes)) != 0:
            raise _err("c4gpu_ctx_sdp_reserve")

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mem = C.c_int(), C.c_int64()
        _lib().c4gpu_ctx_device_info(self.ctx, name, 256, ncu, mem)
        return {"name": name.value.decode(), "compute_units": ncu.value, "memory_bytes": mem.value}

    def find_score(self, model, pairs):
        arr, keep = _pairs(pairs)
        out = (C.c_int32 * max(1, len(pairs)))()
        if _lib().c4gpu_optimal_find_score_batch(self.ctx, model.c, model.params, arr, len(pairs), out) != 0:
            raise _err("c4gpu_optimal_find_score_batch")
        return list(out)[:len(pairs)]

    def find_path(self, model, pairs, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE):
        arr, keep = _pairs(pairs)
        out = (_abi.Alignment * max(1, len(pairs)))()
        if _lib().c4gpu_optimal_find_path_batch(self.ctx, model.c, model.params, arr, len(pairs), dpmemory,
                                                threshold, out) != 0:
            raise _err("c4gpu_optimal_find_path_batch")
        res = []
        for i in range(len(pairs)):
            if out[i].valid:
                res.append(Alignment(model, out[i], len(keep[i][0]), len(keep[i][1])))
            else:
                res.append(None)
            _lib().c4gpu_alignment_clear(out[i])
        return res

    def find_path_subopt(self, model, pairs, subopts, active=None, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE):
        """One round of the sub-optimal loop: subopts[i] is a SubOpt or None."""
        arr, keep = _pairs(pairs)
        n = len(pairs)
        out = (_abi.Alignment * max(1, n))()
        so = (C.c_void_p * max(1, n))(*[(s.h if s is not None else None) for s in subopts])
        act = None if active is None else (C.c_uint8 * max(1, n))(*[1 if a else 0 for a in active])
        if _lib().c4gpu_optimal_find_path_batch_subopt(self.ctx, model.c, model.params, arr, n, dpmemory,
                                                       threshold, so, act, out) != 0:
            raise _err("c4gpu_optimal_find_path_batch_subopt")
        res = []
        for i in range(n):
            res.append(Alignment(model, out[i], len(keep[i][0]), len(keep[i][1])) if out[i].valid else None)
            _lib().c4gpu_alignment_clear(out[i])
        return res

    def find_all_paths(self, model, pairs, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE, max_paths=1 << 30):
        """GAM_Result_exhaustive_create's loop (gam.c:1139-1180) for every pair at once: successive best
        paths, each round with the match cells of everything found so far blocked, until a pair's score
        drops below the threshold.  Returns a list of alignment lists."""
        n = len(pairs)
        found = [[] for _ in range(n)]
        subs = [SubOpt(len(q), len(t)) for q, t in pairs]
        active = [True] * n
        try:
            for _ in range(max_paths):
                if not any(active):
                    break
                res = self.find_path_subopt(model, pairs, subs, active, dpmemory, threshold)
                for i, a in enumerate(res):
                    if not active[i]:
                        continue
                    if a is None:
                        active[i] = False
                    else:
                        found[i].append(a)
                        subs[i].add_alignment(a)
        finally:
            for s in subs:
                s.close()
        return found

    def viterbi(self, model, mode, pairs, jobs):
        """Raw Viterbi_DP_Func level: jobs = list of dict(pair, region, continuation=None, checkpoints=0,
        subopt=None)."""
        arr, keep = _pairs(pairs)
        cj = _viterbi_jobs(jobs)
        res = (_abi.ViterbiResult * max(1, len(jobs)))()
        if _lib().c4gpu_viterbi_batch(self.ctx, model.c, model.params, mode, arr, len(pairs), cj, len(jobs),
                                      res) != 0:
            raise _err("c4gpu_viterbi_batch")
        return _viterbi_results(res, len(jobs))

    def hsp_extend(self, params, match, pairs, seedlen, dropoff, seeds):
        """HSPset_seed_hsp's ungapped X-drop extension (hspset.c:933) of every seed (pair index, query_start,
        target_start) in one launch: [query_start, target_start, length, score, cobs] per seed.  `match`:
        "dna2dna" | "protein2protein" | "protein2dna"."""
        arr, keep = _pairs(pairs)
        kind = {"dna2dna": _abi.MATCH_DNA2DNA, "protein2protein": _abi.MATCH_PROTEIN2PROTEIN,
                "protein2dna": _abi.MATCH_PROTEIN2DNA}[match]
        n = len(seeds)
        cs = (_abi.HspSeed * max(1, n))(*[_abi.HspSeed(*s) for s in seeds])
        out = (_abi.Hsp * max(1, n))()
        if _lib().c4gpu_hsp_extend_batch(self.ctx, params, kind, arr, len(pairs), seedlen, dropoff, cs, n, out) != 0:
            raise _err("c4gpu_hsp_extend_batch")
        return [out[i].aslist() for i in range(n)]

    def hsp_extend_chains(self, params, match, pairs, seedlen, dropoff, seeds, chain, horizon0):
        """The same with the diagonal horizon applied on the device (c4gpu_hsp_extend_chains): chain[k] names the horizon
        entry of seed k, horizon0[c] its value before the scan; a skipped seed comes back with length -1."""
        arr, keep = _pairs(pairs)
        kind = {"dna2dna": _abi.MATCH_DNA2DNA, "protein2protein": _abi.MATCH_PROTEIN2PROTEIN,
                "protein2dna": _abi.MATCH_PROTEIN2DNA}[match]
        n, nc = len(seeds), len(horizon0)
        cs = (_abi.HspSeed * max(1, n))(*[_abi.HspSeed(*s) for s in seeds])
        cc = (C.c_int32 * max(1, n))(*chain)
        h0 = (C.c_int32 * max(1, nc))(*horizon0)
        out = (_abi.Hsp * max(1, n))()
        if _lib().c4gpu_hsp_extend_chains(self.ctx, params, kind, arr, len(pairs), seedlen, dropoff, cs, n, cc, nc, h0, out) != 0:
            raise _err("c4gpu_hsp_extend_chains")
        return [out[i].aslist() for i in range(n)]

    def sdp(self, model, pairs, hsps, query_advance=1, target_advance=1, dropoff=50, threshold=100, max_alignments=4):
        """The reference's default gapped-extension heuristic (SDP, GAM_Result_SDP_create gam.c:852) for a batch: per pair
        the list of alignments found from its HSPs ([query_start, target_start, length, score, cobs] each), both
        Scheduler passes on the device (c4gpu_sdp_batch: one sparse wavefront per pair, both SDP flavours).  A pair the
        device could not serve (its traceback did not fit the device's memory) yields None."""
        arr, keep = _pairs(pairs)
        flat = [h for hs in hsps for h in hs]
        first = [0]
        for hs in hsps:
            first.append(first[-1] + len(hs))
        ch = (_abi.Hsp * max(1, len(flat)))(*[_abi.Hsp(*h) for h in flat])
        cf = (C.c_int32 * len(first))(*first)
        out = (_abi.Alignment * max(1, len(pairs) * max_alignments))()
        n_out = (C.c_int32 * max(1, len(pairs)))()
        if _lib().c4gpu_sdp_batch(self.ctx, model.c, model.params, arr, len(pairs), ch, cf, query_advance, target_advance,
                                  dropoff, threshold, max_alignments, out, n_out) != 0:
            raise _err("c4gpu_sdp_batch")
        res = []
        for i in range(len(pairs)):
            if n_out[i] < 0:
                res.append(None)
                continue
            mine = []
            for k in range(n_out[i]):
                a = out[i * max_alignments + k]
                mine.append(Alignment(model, a, len(pairs[i][0]), len(pairs[i][1])))
                _lib().c4gpu_alignment_clear(a)
            res.append(mine)
        return res

    def seed_scan(self, width, wordlen, words, symbols):
        """The seeder's word scan on the device (c4gpu_seed_scan).  words: {code: number of emissions} in emission-list
        order (the k-th word's emissions follow those of the words before it); symbols: bytes of automaton columns
        (0 = outside the alphabet).  Returns ([(position of the word's last symbol, emission index)], device ms)."""
        import numpy as np
        codes = (C.c_uint64 * max(1, len(words)))(*[c for c, _ in words])
        first = [0]
        for _, n in words:
            first.append(first[-1] + n)
        cf = (C.c_int32 * len(first))(*first)
        tab = _lib().c4gpu_wordtab_create(self.ctx, width, wordlen, codes, cf, len(words))
        if not tab:
            raise _err("c4gpu_wordtab_create")
        try:
            n_hits = C.c_int64(0)
            cap = 1 << 16
            while True:
                buf = np.empty((max(1, cap), 2), dtype=np.int32)
                if _lib().c4gpu_seed_scan(self.ctx, tab, symbols, len(symbols), buf.ctypes.data, cap, C.byref(n_hits)) != 0:
                    raise _err("c4gpu_seed_scan")
                if n_hits.value <= cap:
                    break
                cap = n_hits.value
            ms = C.c_double(0)
            _lib().c4gpu_wordtab_stats(tab, C.byref(ms), None, None, None)
            return [tuple(int(x) for x in r) for r in buf[:n_hits.value]], ms.value
        finally:
            _lib().c4gpu_wordtab_destroy(tab)

    @staticmethod
    def sdp_stats(reset=False):
        """Counters of this thread's sdp() calls: kernel ms of the passes / walks, staging / host ms, jobs, reruns, unserved."""
        d = [C.c_double() for _ in range(4)]
        n = [C.c_int64() for _ in range(3)]
        _lib().c4gpu_sdp_stats(1 if reset else 0, *[C.byref(x) for x in d + n])
        return dict(pass_ms=d[0].value, walk_ms=d[1].value, stage_ms=d[2].value, host_ms=d[3].value, jobs=n[0].value,
                    reruns=n[1].value, unserved=n[2].value)

    def splice_predict(self, params, target):
        t = target if isinstance(target, bytes) else target.encode()
        bufs = [(C.c_int32 * max(1, len(t)))() for _ in range(4)]
        ptrs = (C.POINTER(C.c_int32) * 4)(*[C.cast(b, C.POINTER(C.c_int32)) for b in bufs])
        if _lib().c4gpu_splice_predict(self.ctx, params, t, len(t), ptrs) != 0:
            raise _err("c4gpu_splice_predict")
        return [list(b)[:len(t)] for b in bufs]


class ResidentBatch:
    """Pairs uploaded once (c4gpu_batch): what bench.py times."""

    def __init__(self, engine, model, pairs):
        self.engine, self.model = engine, model
        arr, self._keep = _pairs(pairs)
        self.n = len(pairs)
        self.h = _lib().c4gpu_batch_create(engine.ctx, model.c, model.params, arr, len(pairs))
        if not self.h:
            raise _err("c4gpu_batch_create")

    def run(self, what=2, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE):
        if _lib().c4gpu_batch_run(self.h, what, dpmemory, threshold) != 0:
            raise _err("c4gpu_batch_run")

    def swap(self, stage):
        """Take the sequences `stage` has loaded (c4gpu_batch_swap_stage); the stage gets this batch's previous ones, whose
        buffers its next load reuses.  Earlier results of the batch are dropped."""
        if _lib().c4gpu_batch_swap_stage(self.h, stage.h) != 0:
            raise _err("c4gpu_batch_swap_stage")
        self._keep, stage._keep = stage._keep, self._keep
        self.n, stage.n = stage.n, self.n

    def run_regions(self, regions, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE, active=None):
        """Optimal_find_path of every pair over its own region (query_start, target_start, query_length,
        target_length) of the rectangle: what --refine region asks for (c4gpu_batch_run_regions)."""
        arr = (_abi.Region * max(1, self.n))(*[_abi.Region(*r) for r in regions])
        act = None if active is None else (C.c_uint8 * max(1, self.n))(*[1 if a else 0 for a in active])
        if _lib().c4gpu_batch_run_regions(self.h, arr, act, dpmemory, threshold) != 0:
            raise _err("c4gpu_batch_run_regions")

    def set_annotation(self, cds):
        """exonerate's --annotation: cds[i] = (cds_start, cds_length) of pair i's query, or None (c4gpu_batch_set_annotation);
        applies to the sequences the batch holds now."""
        if cds is None:
            rc = _lib().c4gpu_batch_set_annotation(self.h, None, None)
        else:
            st = (C.c_int32 * max(1, self.n))(*[(c[0] if c else 0) for c in cds])
            ln = (C.c_int32 * max(1, self.n))(*[(c[1] if c else 0) for c in cds])
            rc = _lib().c4gpu_batch_set_annotation(self.h, st, ln)
        if rc != 0:
            raise _err("c4gpu_batch_set_annotation")

    def set_thresholds(self, per_pair):
        """Per-pair score thresholds (exonerate's --percent); None switches them off."""
        arr = None if per_pair is None else (C.c_int32 * max(1, self.n))(*per_pair)
        _lib().c4gpu_batch_set_thresholds(self.h, arr)

    def next_paths(self, dpmemory=32, threshold=IMPOSSIBLY_LOW_SCORE):
        """Next round of the sub-optimal loop (c4gpu_batch_next_paths); returns the number found."""
        n = _lib().c4gpu_batch_next_paths(self.h, dpmemory, threshold)
        if n < 0:
            raise _err("c4gpu_batch_next_paths")
        return n

    def scores(self):
        s = (C.c_int32 * max(1, self.n))()
        r = (_abi.Region * max(1, self.n))()
        _lib().c4gpu_batch_scores(self.h, s, r)
        return list(s)[:self.n], [x.astuple() for x in r][:self.n]

    def export(self):
        """All alignments as one int32 numpy array (c4gpu_batch_export): a row of 7 ints per pair (valid, score, region (4),
        n_ops), then the (transition, length) pairs of all valid alignments in pair order."""
        import numpy as np
        need = _lib().c4gpu_batch_export(self.h, None, 0)
        buf = np.empty(max(1, need), dtype=np.int32)
        got = _lib().c4gpu_batch_export(self.h, buf.ctypes.data_as(C.POINTER(C.c_int32)), need)
        assert got == need
        return buf[:need]

    def alignment(self, i):
        a = _abi.Alignment()
        if _lib().c4gpu_batch_alignment(self.h, i, a) != 0:
            raise _err("c4gpu_batch_alignment")
        out = Alignment(self.model, a, len(self._keep[i][0]), len(self._keep[i][1])) if a.valid else None
        _lib().c4gpu_alignment_clear(a)
        return out

    def viterbi(self, mode, jobs, model=None):
        """Viterbi_DP_Func-level jobs on the resident pairs; `model` = another model of the same family (BSDP's
        derived terminal / join / span models) that shares the batch's sequence arrays."""
        cj = _viterbi_jobs(jobs)
        res = (_abi.ViterbiResult * max(1, len(jobs)))()
        if model is None:
            rc = _lib().c4gpu_batch_viterbi(self.h, mode, cj, len(jobs), res)
        else:
            rc = _lib().c4gpu_batch_viterbi_model(self.h, model.c, mode, cj, len(jobs), res)
        if rc != 0:
            raise _err("c4gpu_batch_viterbi")
        return _viterbi_results(res, len(jobs))

    def kernel_stats(self, mode, reset=False):
        ms, n, cells = C.c_double(), C.c_int64(), C.c_int64()
        _lib().c4gpu_batch_kernel_stats(self.h, mode, int(reset), ms, n, cells)
        return {"ms": ms.value, "launches": n.value, "cells": cells.value}

    def close(self):
        if self.h:
            _lib().c4gpu_batch_destroy(self.h)
            self.h = None


class Stage:
    """The next batch on its way to the device (c4gpu_stage): load() from one thread while ResidentBatch.run() is in
    progress on another (ctypes releases the GIL inside both), then ResidentBatch.swap(stage)."""

    def __init__(self, engine, model):
        self.engine, self.model = engine, model
        self._keep, self.n = [], 0
        self.h = _lib().c4gpu_stage_create(engine.ctx, model.c, model.params)
        if not self.h:
            raise _err("c4gpu_stage_create")

    def load(self, pairs):
        arr, self._keep = _pairs(pairs)
        self.n = len(pairs)
        if _lib().c4gpu_stage_load(self.h, arr, len(pairs)) != 0:
            raise _err("c4gpu_stage_load")
        return _lib().c4gpu_stage_load_ms(self.h)

    def close(self):
        if self.h:
            _lib().c4gpu_stage_destroy(self.h)
            self.h = None
