"""ctypes mirror of include/c4gpu.h / include/c4m.h (the C ABI of libc4gpu.so).

Only plumbing lives here: struct layouts, function prototypes and the loader.  The product is the shared
library; this module never computes anything itself and raises loudly when the library is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libc4gpu.so")

ABI_VERSION = 9            # C4GPU_ABI_VERSION of include/c4gpu.h these structures mirror
MAX_STATES, MAX_TRANSITIONS, MAX_CALCS, MAX_SHADOWS, NAME_LEN = 16, 48, 16, 4, 48
SPLICE_MAX_LEN = 32
CELL_MAX = 1 + MAX_SHADOWS + 3
IMPOSSIBLY_LOW_SCORE = -987654321

SCOPE_ANYWHERE, SCOPE_EDGE, SCOPE_QUERY, SCOPE_TARGET, SCOPE_CORNER = range(5)
(LABEL_NONE, LABEL_MATCH, LABEL_GAP, LABEL_NER, LABEL_5SS, LABEL_3SS, LABEL_INTRON, LABEL_SPLIT_CODON,
 LABEL_FRAMESHIFT) = range(9)
MODE_FIND_SCORE, MODE_FIND_PATH, MODE_FIND_REGION, MODE_FIND_CHECKPOINTS = range(4)
ALPHABET_DNA, ALPHABET_PROTEIN = 0, 1
SS5_FORWARD, SS3_FORWARD, SS3_REVERSE, SS5_REVERSE = range(4)
(CALC_CONST, CALC_MATCH_DNA, CALC_MATCH_PROTEIN, CALC_MATCH_P2D, CALC_SPLICE_PRE, CALC_SPLICE_POST,
 CALC_PHASE_PRE, CALC_PHASE_POST) = range(8)


class Calc(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("kind", C.c_int32), ("value", C.c_int32),
                ("param", C.c_int32), ("max_score", C.c_int32), ("protect", C.c_int32)]


class Transition(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("input", C.c_int32), ("output", C.c_int32),
                ("advance_query", C.c_int32), ("advance_target", C.c_int32), ("calc", C.c_int32),
                ("label", C.c_int32), ("dst_shadow_mask", C.c_uint32)]


class Shadow(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("designation", C.c_int32), ("on_target", C.c_int32),
                ("src_state_mask", C.c_uint32), ("dst_transition_mask", C.c_uint64)]


class Model(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN),
                ("n_states", C.c_int32), ("n_transitions", C.c_int32), ("n_calcs", C.c_int32),
                ("n_shadows", C.c_int32), ("start_state", C.c_int32), ("end_state", C.c_int32),
                ("start_scope", C.c_int32), ("end_scope", C.c_int32),
                ("max_query_advance", C.c_int32), ("max_target_advance", C.c_int32),
                ("total_shadow_designations", C.c_int32),
                ("query_alphabet", C.c_int32), ("target_alphabet", C.c_int32),
                ("state_names", (C.c_char * NAME_LEN) * MAX_STATES),
                ("calcs", Calc * MAX_CALCS),
                ("transitions", Transition * MAX_TRANSITIONS),
                ("shadows", Shadow * MAX_SHADOWS)]


class SpliceModel(C.Structure):
    _fields_ = [("model_length", C.c_int32), ("splice_after", C.c_int32),
                ("index", C.c_uint8 * 256), ("data", (C.c_float * 5) * SPLICE_MAX_LEN),
                ("gtag_only", C.c_int32), ("expect_one", C.c_uint8), ("expect_two", C.c_uint8),
                ("pad_", C.c_uint8 * 2)]


class Params(C.Structure):
    _fields_ = [("dna_submat", (C.c_int32 * 24) * 24), ("protein_submat", (C.c_int32 * 24) * 24),
                ("submat_index", C.c_uint8 * 256), ("nt2d", C.c_uint8 * 256),
                ("trans", C.c_uint8 * 4096), ("aa", C.c_uint8 * 40),
                ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("codon_gap_open", C.c_int32), ("codon_gap_extend", C.c_int32),
                ("min_intron", C.c_int32), ("max_intron", C.c_int32),
                ("intron_open_penalty", C.c_int32), ("frameshift_penalty", C.c_int32),
                ("splice", SpliceModel * 4)]


class Region(C.Structure):
    _fields_ = [("query_start", C.c_int32), ("target_start", C.c_int32),
                ("query_length", C.c_int32), ("target_length", C.c_int32)]

    def astuple(self):
        return (self.query_start, self.target_start, self.query_length, self.target_length)


class Pair(C.Structure):
    _fields_ = [("query", C.c_char_p), ("query_len", C.c_int32),
                ("target", C.c_char_p), ("target_len", C.c_int32)]


class Alignment(C.Structure):
    _fields_ = [("score", C.c_int32), ("region", Region), ("n_ops", C.c_int32),
                ("op_transition", C.POINTER(C.c_int32)), ("op_length", C.POINTER(C.c_int32)),
                ("valid", C.c_int32)]


class GffRequest(C.Structure):
    _fields_ = [("query_id", C.c_char_p), ("target_id", C.c_char_p), ("query", C.c_char_p), ("target", C.c_char_p),
                ("query_len", C.c_int32), ("target_len", C.c_int32), ("query_strand", C.c_char), ("target_strand", C.c_char),
                ("report_on_query", C.c_int32), ("report_on_genomic", C.c_int32), ("result_id", C.c_int32),
                ("date", C.c_char_p), ("version", C.c_char_p)]


class DisplayRequest(C.Structure):
    _fields_ = [("query_id", C.c_char_p), ("query_def", C.c_char_p), ("target_id", C.c_char_p), ("target_def", C.c_char_p),
                ("query", C.c_char_p), ("target", C.c_char_p), ("query_len", C.c_int32), ("target_len", C.c_int32),
                ("query_strand", C.c_char), ("target_strand", C.c_char), ("width", C.c_int32), ("forward_coords", C.c_int32),
                ("use_aa_tla", C.c_int32)]


class RyoRequest(C.Structure):
    _fields_ = [("query_id", C.c_char_p), ("query_def", C.c_char_p), ("target_id", C.c_char_p), ("target_def", C.c_char_p),
                ("query", C.c_char_p), ("target", C.c_char_p), ("query_len", C.c_int32), ("target_len", C.c_int32),
                ("query_strand", C.c_char), ("target_strand", C.c_char), ("forward_coords", C.c_int32), ("rank", C.c_int32),
                ("format", C.c_char_p)]


class HspSeed(C.Structure):
    _fields_ = [("pair", C.c_int32), ("query_start", C.c_int32), ("target_start", C.c_int32)]


class Hsp(C.Structure):
    _fields_ = [("query_start", C.c_int32), ("target_start", C.c_int32), ("length", C.c_int32),
                ("score", C.c_int32), ("cobs", C.c_int32)]

    def aslist(self):
        return [self.query_start, self.target_start, self.length, self.score, self.cobs]


MATCH_DNA2DNA, MATCH_PROTEIN2PROTEIN, MATCH_PROTEIN2DNA = 0, 1, 2


class Continuation(C.Structure):
    _fields_ = [("first_state", C.c_int32), ("final_state", C.c_int32),
                ("first_cell", C.c_int32 * CELL_MAX)]


class ViterbiJob(C.Structure):
    _fields_ = [("pair", C.c_int32), ("region", Region), ("use_continuation", C.c_int32),
                ("continuation", Continuation), ("checkpoint_count", C.c_int32), ("subopt", C.c_void_p),
                ("start_cells", C.POINTER(C.c_int32)), ("end_cells", C.POINTER(C.c_int32))]


class ViterbiResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("query_start", C.c_int32), ("target_start", C.c_int32),
                ("query_end", C.c_int32), ("target_end", C.c_int32),
                ("final_cell", C.c_int32 * CELL_MAX), ("last_srp", C.c_int32),
                ("n_ops", C.c_int32), ("ops", C.POINTER(C.c_int32)),
                ("checkpoints", C.POINTER(C.c_int32))]


# (name, restype, argtypes) for every symbol include/*.h declares
PROTOTYPES = [
    ("c4gpu_abi_version", C.c_int, []),
    ("c4gpu_config_reload", C.c_int, []),
    ("c4gpu_last_error", C.c_char_p, []),
    ("c4gpu_ctx_create", C.c_void_p, [C.c_int]),
    ("c4gpu_ctx_destroy", None, [C.c_void_p]),
    ("c4gpu_ctx_warm", None, [C.c_void_p]),
    ("c4gpu_ctx_warm_cancel", None, []),
    ("c4gpu_ctx_set_stream", None, [C.c_void_p, C.c_void_p]),
    ("c4gpu_ctx_own_stream", C.c_int, [C.c_void_p]),
    ("c4gpu_ctx_sdp_reserve", C.c_int, [C.c_void_p, C.c_int64]),
    ("c4gpu_ctx_device_info", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int64)]),
    ("c4gpu_params_default", None, [C.POINTER(Params)]),
    ("c4gpu_params_set_forcegtag", None, [C.POINTER(Params), C.c_int]),
    ("c4gpu_model_get", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(Model)]),
    ("c4gpu_model_get_derived", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.POINTER(Model), C.POINTER(C.c_int32)]),
    ("c4m_derive", C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("c4gpu_model_make_continuation", None, [C.POINTER(Model), C.POINTER(Model)]),
    ("c4gpu_model_plugin_name", C.c_int, [C.POINTER(Model), C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    ("c4gpu_model_is_accelerated", C.c_int, [C.POINTER(Model)]),
    ("c4gpu_use_reduced_space", C.c_int, [C.POINTER(Model), C.POINTER(Region), C.c_int]),
    ("c4gpu_checkpoint_rows", C.c_int, [C.POINTER(Model), C.POINTER(Region), C.c_int]),
    ("c4gpu_memrule_device", C.c_int, [C.c_void_p, C.POINTER(Model), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("c4gpu_splice_predict", C.c_int, [C.c_void_p, C.POINTER(Params), C.c_char_p, C.c_int32,
                                       C.POINTER(C.POINTER(C.c_int32))]),
    ("c4gpu_viterbi_batch", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Params), C.c_int,
                                      C.POINTER(Pair), C.c_int32, C.POINTER(ViterbiJob), C.c_int32,
                                      C.POINTER(ViterbiResult)]),
    ("c4gpu_viterbi_result_clear", None, [C.POINTER(ViterbiResult)]),
    ("c4gpu_optimal_find_score_batch", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Params),
                                                 C.POINTER(Pair), C.c_int32, C.POINTER(C.c_int32)]),
    ("c4gpu_optimal_find_path_batch", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Params),
                                                C.POINTER(Pair), C.c_int32, C.c_int, C.c_int32,
                                                C.POINTER(Alignment)]),
    ("c4gpu_optimal_find_path_batch_subopt", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Params),
                                                       C.POINTER(Pair), C.c_int32, C.c_int, C.c_int32,
                                                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint8),
                                                       C.POINTER(Alignment)]),
    ("c4gpu_alignment_clear", None, [C.POINTER(Alignment)]),
    ("c4gpu_subopt_create", C.c_void_p, [C.c_int32, C.c_int32]),
    ("c4gpu_subopt_destroy", None, [C.c_void_p]),
    ("c4gpu_subopt_add_alignment", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Alignment)]),
    ("c4gpu_subopt_add_point", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("c4gpu_subopt_points", C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]),
    ("c4gpu_batch_next_paths", C.c_int, [C.c_void_p, C.c_int, C.c_int32]),
    ("c4gpu_batch_create", C.c_void_p, [C.c_void_p, C.POINTER(Model), C.POINTER(Params),
                                        C.POINTER(Pair), C.c_int32]),
    ("c4gpu_batch_destroy", None, [C.c_void_p]),
    ("c4gpu_batch_run", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int32]),
    ("c4gpu_batch_run_regions", C.c_int, [C.c_void_p, C.POINTER(Region), C.POINTER(C.c_uint8), C.c_int, C.c_int32]),
    ("c4gpu_batch_set_thresholds", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    ("c4gpu_batch_viterbi", C.c_int, [C.c_void_p, C.c_int, C.POINTER(ViterbiJob), C.c_int32,
                                      C.POINTER(ViterbiResult)]),
    ("c4gpu_model_device_family", C.c_int, [C.POINTER(Model)]),
    ("c4gpu_hsp_extend_batch", C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(Pair), C.c_int32, C.c_int32,
                                         C.c_int32, C.POINTER(HspSeed), C.c_int32, C.POINTER(Hsp)]),
    ("c4gpu_hsp_extend_chains", C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(Pair), C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(HspSeed), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(Hsp)]),
    ("c4gpu_wordtab_create", C.c_void_p, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_int32]),
    ("c4gpu_wordtab_destroy", None, [C.c_void_p]),
    ("c4gpu_seed_scan", C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    ("c4gpu_wordtab_stats", None, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("c4gpu_sdp_lattice_cells", C.c_double, [C.POINTER(Hsp), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("c4gpu_sdp_batch", C.c_int, [C.c_void_p, C.POINTER(Model), C.POINTER(Params), C.POINTER(Pair), C.c_int32, C.POINTER(Hsp),
                                  C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(Alignment), C.POINTER(C.c_int32)]),
    ("c4gpu_sdp_stats", None, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("c4gpu_batch_viterbi_model", C.c_int, [C.c_void_p, C.POINTER(Model), C.c_int, C.POINTER(ViterbiJob), C.c_int32,
                                            C.POINTER(ViterbiResult)]),
    ("c4gpu_batch_scores", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(Region)]),
    ("c4gpu_batch_alignment", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Alignment)]),
    ("c4gpu_batch_export", C.c_int64, [C.c_void_p, C.POINTER(C.c_int32), C.c_int64]),
    ("c4gpu_batch_kernel_stats", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("c4gpu_packed_route_fits", C.c_int, [C.POINTER(Model), C.POINTER(Params), C.c_int32, C.c_int32]),
    ("c4gpu_batch_set_annotation", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("c4gpu_loop_sections", C.c_int, [C.POINTER(Model), C.POINTER(Params), C.POINTER(C.c_int32), C.c_int32]),
    ("c4gpu_stage_create", C.c_void_p, [C.c_void_p, C.POINTER(Model), C.POINTER(Params)]),
    ("c4gpu_stage_load", C.c_int, [C.c_void_p, C.POINTER(Pair), C.c_int32]),
    ("c4gpu_stage_load_ms", C.c_double, [C.c_void_p]),
    ("c4gpu_batch_swap_stage", C.c_int, [C.c_void_p, C.c_void_p]),
    ("c4gpu_stage_destroy", None, [C.c_void_p]),
    ("c4gpu_alignment_format", C.c_int, [C.POINTER(Model), C.POINTER(Alignment), C.c_int,
                                         C.c_char_p, C.c_int32, C.c_char, C.c_char_p, C.c_int32, C.c_char,
                                         C.c_int, C.c_char_p, C.c_size_t]),
    ("c4gpu_alignment_format_gff", C.c_int, [C.POINTER(Model), C.POINTER(Params), C.POINTER(Alignment), C.POINTER(GffRequest),
                                    C.c_char_p, C.c_size_t]),
    ("c4gpu_alignment_display", C.c_int, [C.POINTER(Model), C.POINTER(Params), C.POINTER(Alignment), C.POINTER(DisplayRequest),
                                 C.c_char_p, C.c_size_t]),
    ("c4gpu_alignment_format_ryo", C.c_int, [C.POINTER(Model), C.POINTER(Params), C.POINTER(Alignment), C.POINTER(RyoRequest),
                                    C.c_char_p, C.c_size_t]),
    ("c4gpu_splice_max_score", C.c_float, [C.POINTER(SpliceModel)]),
    # c4m.h
    ("c4m_model_create", C.c_void_p, [C.c_char_p]),
    ("c4m_model_destroy", None, [C.c_void_p]),
    ("c4m_model_rename", None, [C.c_void_p, C.c_char_p]),
    ("c4m_model_open", None, [C.c_void_p]),
    ("c4m_model_close", C.c_int, [C.c_void_p]),
    ("c4m_model_is_open", C.c_int, [C.c_void_p]),
    ("c4m_model_set_alphabets", None, [C.c_void_p, C.c_int, C.c_int]),
    ("c4m_add_state", C.c_int, [C.c_void_p, C.c_char_p]),
    ("c4m_add_calc", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("c4m_add_transition", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int]),
    ("c4m_add_shadow", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]),
    ("c4m_shadow_add_src_state", None, [C.c_void_p, C.c_int, C.c_int]),
    ("c4m_shadow_add_dst_transition", None, [C.c_void_p, C.c_int, C.c_int]),
    ("c4m_configure_start_state", None, [C.c_void_p, C.c_int]),
    ("c4m_configure_end_state", None, [C.c_void_p, C.c_int]),
    ("c4m_make_stereo", None, [C.c_void_p, C.c_char_p, C.c_char_p]),
    ("c4m_insert", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("c4m_select_single_transition", C.c_int, [C.c_void_p, C.c_int]),
    ("c4m_select_transitions", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("c4m_transition_input", C.c_int, [C.c_void_p, C.c_int]),
    ("c4m_transition_output", C.c_int, [C.c_void_p, C.c_int]),
    ("c4m_transition_id", C.c_int, [C.c_void_p, C.c_int]),
    ("c4m_flatten", C.c_int, [C.c_void_p, C.POINTER(Model)]),
    ("c4m_ungapped_create", C.c_void_p, [C.c_int, C.c_int, C.POINTER(Params)]),
    ("c4m_affine_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, C.POINTER(Params)]),
    ("c4m_intron_create", C.c_void_p, [C.c_char_p, C.c_int, C.POINTER(Params)]),
    ("c4m_est2genome_create", C.c_void_p, [C.POINTER(Params)]),
    ("c4m_protein2dna_create", C.c_void_p, [C.c_int, C.POINTER(Params)]),
    ("c4m_phase_create", C.c_void_p, [C.POINTER(Params)]),
    ("c4m_protein2genome_create", C.c_void_p, [C.c_int, C.POINTER(Params)]),
]

_lib = None


def load(path=None):
    """dlopen libc4gpu.so and attach prototypes.  Raises if the library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("C4GPU_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(
            "libc4gpu.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no Python/CPU fallback for the C4 engine" % p)
    # torch first, where there is one: it brings its own copy of the HIP runtime, and a process in which libc4gpu.so has pulled in
    # /opt/rocm's before torch loads its own ends up with two runtimes, the first of which no longer finds the device
    # ("no ROCm-capable device is detected" from c4gpu_ctx_create after `build(); smoke()` in one process).  Loaded second, the
    # library binds to the runtime that is already there (same soname).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    lib.c4gpu_abi_version.restype = C.c_int
    if lib.c4gpu_abi_version() != ABI_VERSION:
        raise OSError("libc4gpu.so has ABI version %d, exonerate_amd/_abi.py mirrors version %d: rebuild "
                      "(python -c 'import __graft_entry__ as g; g.build()')" % (lib.c4gpu_abi_version(), ABI_VERSION))
    for name, res, args in PROTOTYPES:
        fn = getattr(lib, name)          # AttributeError if the ABI header and the library diverge
        fn.restype = res
        fn.argtypes = args
    lib = _EnvSynced(lib)
    if path is None:
        _lib = lib
    return lib


class _EnvSynced:
    """The library reads its C4GPU_* switches once (csrc/c4_config.h); tests flip them between two calls.  Every call through
    this proxy first compares the process's C4GPU_* variables with what the library last read and asks it to read them again
    (c4gpu_config_reload) when they differ -- the explicit hook, used by this plumbing and by nothing in the library."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_seen", self._snapshot())
        object.__setattr__(self, "_calls", {})

    @staticmethod
    def _snapshot():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("C4GPU_")))

    def _sync(self):
        now = self._snapshot()
        if now != self._seen:
            object.__setattr__(self, "_seen", now)
            self._cdll.c4gpu_config_reload()

    def __getattr__(self, name):
        calls = object.__getattribute__(self, "_calls")
        if name in calls:
            return calls[name]
        fn = getattr(self._cdll, name)

        def call(*args, _fn=fn):
            self._sync()
            return _fn(*args)
        call.__name__ = name
        call.restype, call.argtypes = getattr(fn, "restype", None), getattr(fn, "argtypes", None)
        calls[name] = call
        return call
