/* c4gpu.h — C ABI of the MI355X-native C4 dynamic-programming engine.
 *
 * This is the drop-in boundary.  Everything here is plain C: POD structs, pointers and
 * sizes; no torch / HIP / GLib types appear in any signature.  Each entry point cites the reference
 * interface (file:line under the exonerate tree) it replaces; INTEGRATION.md shows the binding a
 * maintainer adds on the exonerate side (a `Viterbi_DP_Func` shim + a batching `JobQueue`).
 *
 * Layering (mirrors the reference):
 *   c4gpu_model / c4m_*      <->  C4_Model builder API               src/c4/c4.h:198-356, c4.c:1669
 *   c4gpu_model_get          <->  Model_Type_get_model               src/model/modeltype.c
 *   c4gpu_viterbi_batch      <->  Viterbi_DP_Func / Viterbi_calculate src/c4/viterbi.h:95-98, viterbi.c:846
 *   c4gpu_optimal_*_batch    <->  Optimal_find_score / _find_path    src/c4/optimal.c:123,368
 *   c4gpu_splice_predict     <->  SplicePredictor_predict_array_int  src/sequence/splice.c:383
 *   c4gpu_alignment_format   <->  Alignment_print_{sugar,cigar,vulgar}_block  src/c4/alignment.c:1622-1779
 *   c4gpu_hsp_extend_batch   <->  HSPset_seed_hsp (HSP_trim_ends/_init/_extend) src/comparison/hspset.c:933-997
 *   c4gpu_hsp_extend_chains  <->  the same with its horizon test              src/comparison/hspset.c:939-958,990
 *   c4gpu_alignment_display / _format_gff / _format_ryo  <->  Alignment_display / _display_gff / _display_ryo  src/c4/alignment.c:1343,3212,2659
 *   c4gpu_batch_run_regions  <->  Optimal_find_path with a region (--refine)   src/hub/gam.c:605-655
 *   c4gpu_seed_scan          <->  Seeder_add_target's automaton walk src/comparison/seeder.c:649-720,852-915
 *   c4gpu_sdp_batch          <->  GAM_Result_SDP_create's loop (SDP_Pair_next_path) src/hub/gam.c:852-890, src/sdp/sdp.c:743
 */
#ifndef INCLUDED_C4GPU_H
#define INCLUDED_C4GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C4GPU_ABI_VERSION 9

/* src/c4/c4.h:28-30 */
typedef int32_t c4gpu_score;
#define C4GPU_IMPOSSIBLY_LOW_SCORE  (-987654321)
#define C4GPU_IMPOSSIBLY_HIGH_SCORE (987654321)

#define C4GPU_MAX_STATES       16
#define C4GPU_MAX_TRANSITIONS  48
#define C4GPU_MAX_CALCS        16
#define C4GPU_MAX_SHADOWS       4
#define C4GPU_NAME_LEN         48
#define C4GPU_SUBMAT_SIZE      24   /* SUBMAT_ALPHABETSIZE, src/sequence/submat.h:32 */
#define C4GPU_SPLICE_MAX_LEN   32

/* C4_Scope, src/c4/c4.h:91-97 (same numeric values) */
enum { C4GPU_SCOPE_ANYWHERE = 0, C4GPU_SCOPE_EDGE, C4GPU_SCOPE_QUERY, C4GPU_SCOPE_TARGET, C4GPU_SCOPE_CORNER };
/* C4_Label, src/c4/c4.h:114-124 (same numeric values) */
enum { C4GPU_LABEL_NONE = 0, C4GPU_LABEL_MATCH, C4GPU_LABEL_GAP, C4GPU_LABEL_NER, C4GPU_LABEL_5SS,
       C4GPU_LABEL_3SS, C4GPU_LABEL_INTRON, C4GPU_LABEL_SPLIT_CODON, C4GPU_LABEL_FRAMESHIFT };
/* C4_Protect, src/c4/c4.h:69-73 */
enum { C4GPU_PROTECT_NONE = 0, C4GPU_PROTECT_OVERFLOW = 1, C4GPU_PROTECT_UNDERFLOW = 2 };
/* Viterbi_Mode, src/c4/viterbi.h:104-109 */
enum { C4GPU_MODE_FIND_SCORE = 0, C4GPU_MODE_FIND_PATH, C4GPU_MODE_FIND_REGION, C4GPU_MODE_FIND_CHECKPOINTS };
/* Alphabet_Type, src/sequence/alphabet.h:40-44 */
enum { C4GPU_ALPHABET_DNA = 0, C4GPU_ALPHABET_PROTEIN = 1 };
/* SpliceType order used for the per-target splice arrays */
enum { C4GPU_SS5_FORWARD = 0, C4GPU_SS3_FORWARD = 1, C4GPU_SS3_REVERSE = 2, C4GPU_SS5_REVERSE = 3 };

/* What a C4_Calc's calc_func computes, recognised from the calc (src/c4/c4.h:75-86).  A model whose
 * calcs are all one of these kinds can run on the device; anything else is "not accelerated". */
enum {
    C4GPU_CALC_CONST = 0,        /* returns `value` (gap open/extend affine.c:88-124, frameshift.c:49-59)   */
    C4GPU_CALC_MATCH_DNA,        /* Match_1_1_dna_score_func      match.c:271  dna submat[q][t]              */
    C4GPU_CALC_MATCH_PROTEIN,    /* Match_1_1_protein_score_func  match.c:287  protein submat[q][t]          */
    C4GPU_CALC_MATCH_P2D,        /* Match_1_3_score_func          match.c:347  protein submat[q][aa(t..t+2)] */
    C4GPU_CALC_SPLICE_PRE,       /* Intron_calc_* with is_pre     intron.c:138-161: value + ss[param][tpos]  */
    C4GPU_CALC_SPLICE_POST,      /* Intron_calc_* post: length check against shadow, then ss[param][tpos]    */
    C4GPU_CALC_PHASE_PRE,        /* phase.c split-codon calcs (protein2genome)                               */
    C4GPU_CALC_PHASE_POST
};

/* C4_Calc, src/c4/c4.h:75-86 */
typedef struct {
    char    name[C4GPU_NAME_LEN];
    int32_t kind;       /* C4GPU_CALC_* */
    int32_t value;      /* CONST: the score; SPLICE_PRE: intron open penalty */
    int32_t param;      /* SPLICE_*: which splice array (C4GPU_SS*); PHASE_*: phase (1|2) */
    int32_t max_score;  /* C4_Calc.max_score */
    int32_t protect;    /* C4GPU_PROTECT_* bit set */
} c4gpu_calc;

/* C4_Transition, src/c4/c4.h:126-137; `id` is the index in c4gpu_model.transitions, which is the closed
 * model's evaluation / tie-break order (C4_Model_topological_sort, c4.c:1418). */
typedef struct {
    char    name[C4GPU_NAME_LEN];
    int32_t input, output;                  /* state ids */
    int32_t advance_query, advance_target;
    int32_t calc;                           /* index into calcs, -1 for a NULL calc (scores 0) */
    int32_t label;                          /* C4GPU_LABEL_* */
    uint32_t dst_shadow_mask;               /* shadows whose end_func runs on this transition */
} c4gpu_transition;

/* C4_Shadow, src/c4/c4.h:139-149.  In-scope shadows carry a sequence position (intron.c:454-493). */
typedef struct {
    char    name[C4GPU_NAME_LEN];
    int32_t designation;                    /* cell slot = designation + 1 */
    int32_t on_target;                      /* 1: start_func returns target_pos; 0: query_pos */
    uint32_t src_state_mask;                /* states the shadow starts from */
    uint64_t dst_transition_mask;           /* transitions it ends on (ids up to C4GPU_MAX_TRANSITIONS - 1) */
} c4gpu_shadow;

/* A *closed* C4_Model flattened to a POD (src/c4/c4.h:172-194). */
typedef struct {
    char    name[C4GPU_NAME_LEN];
    int32_t n_states, n_transitions, n_calcs, n_shadows;
    int32_t start_state, end_state;         /* C4_StartState/C4_EndState .state ids */
    int32_t start_scope, end_scope;         /* C4GPU_SCOPE_* */
    int32_t max_query_advance, max_target_advance;
    int32_t total_shadow_designations;
    int32_t query_alphabet, target_alphabet; /* C4GPU_ALPHABET_* (what the match calcs expect) */
    char             state_names[C4GPU_MAX_STATES][C4GPU_NAME_LEN];
    c4gpu_calc       calcs[C4GPU_MAX_CALCS];
    c4gpu_transition transitions[C4GPU_MAX_TRANSITIONS];
    c4gpu_shadow     shadows[C4GPU_MAX_SHADOWS];
} c4gpu_model;

/* One splice-site PSSM after SplicePredictor_create (splice.c:242-295): log-odds floats. */
typedef struct {
    int32_t model_length, splice_after;
    uint8_t index[256];                     /* residue -> column 0..4 */
    float   data[C4GPU_SPLICE_MAX_LEN][5];
    int32_t gtag_only;                      /* --forcegtag (splice.c:42-44,290-293): sites whose two bases */
    uint8_t expect_one, expect_two;         /* (upper-cased) are not these score -987654321.0f (splice.c:335) */
    uint8_t pad_[2];
} c4gpu_splice_model;

/* The scoring data the per-cell calcs read: what `user_data` + the static ArgumentSets hold in the
 * reference (match.c:39, affine.c:21, intron.c:21, frameshift.c, submat.c, translate.c). */
typedef struct {
    int32_t dna_submat[C4GPU_SUBMAT_SIZE][C4GPU_SUBMAT_SIZE];
    int32_t protein_submat[C4GPU_SUBMAT_SIZE][C4GPU_SUBMAT_SIZE];
    uint8_t submat_index[256];              /* residue -> row (24 = not in alphabet: rejected) */
    uint8_t nt2d[256];                      /* translate.h:40-50 */
    uint8_t trans[4096];
    uint8_t aa[40];
    int32_t gap_open, gap_extend, codon_gap_open, codon_gap_extend;   /* affine.c:24-35 */
    int32_t min_intron, max_intron, intron_open_penalty;              /* intron.c:24-32 */
    int32_t frameshift_penalty;
    c4gpu_splice_model splice[4];           /* indexed by C4GPU_SS* */
} c4gpu_params;

/* Region, src/c4/region.h:26-32 */
typedef struct { int32_t query_start, target_start, query_length, target_length; } c4gpu_region;

/* One query x target pair handed over at the boundary: residues already flattened to bytes
 * (Sequence_strncpy, sequence.c:588).  Pointers are HOST pointers for the *_batch calls below. */
typedef struct {
    const uint8_t *query;  int32_t query_len;
    const uint8_t *target; int32_t target_len;
} c4gpu_pair;

/* AlignmentOperation list, src/c4/alignment.h; ops are (transition id, run length). */
typedef struct {
    c4gpu_score  score;
    c4gpu_region region;                    /* in sequence coordinates */
    int32_t      n_ops;
    int32_t     *op_transition;             /* malloc'd by the library, freed by c4gpu_alignment_clear */
    int32_t     *op_length;
    int32_t      valid;                     /* 0: no alignment (score below threshold) */
} c4gpu_alignment;

/* ---- one Viterbi call (Viterbi_DP_Func level) -------------------------------------------------- */

/* Viterbi_Continuation, src/c4/viterbi.h:56-61 */
typedef struct {
    int32_t first_state, final_state;
    c4gpu_score first_cell[1 + C4GPU_MAX_SHADOWS + 3];
} c4gpu_continuation;

/* SubOpt, src/c4/subopt.h:33-48: the cells blocked by the alignments already reported for one pair
 * (sequence coordinates).  Opaque; see c4gpu_subopt_* below. */
typedef struct c4gpu_subopt c4gpu_subopt;

typedef struct {
    int32_t      pair;                      /* index into the pair array */
    c4gpu_region region;
    int32_t      use_continuation;          /* model is run with CORNER/CORNER scopes (viterbi.c:68-76) */
    c4gpu_continuation continuation;
    int32_t      checkpoint_count;          /* FIND_CHECKPOINTS: Viterbi_checkpoint_rows (viterbi.c:207) */
    const c4gpu_subopt *subopt;             /* NULL, or what Viterbi_calculate builds its SubOpt_Index from
                                               (viterbi.c:846-865): MATCH transitions skip blocked cells */
    /* BSDP's span models (heuristic.c:385-443): the model's cell_start_func / cell_end_func as matrices over the
     * job's region, [(i * (target_length + 1)) + j][1 + shadow designations] ints, host memory.
     * start_cells: what cell_start_func returns at (i, j) (viterbi.c:728-741); FIND_SCORE / FIND_PATH.
     * end_cells: receives the END cell of every (i, j) that reaches END (viterbi.c:793-799); other entries are
     * left as the caller initialised them; FIND_SCORE.  Both NULL for every other model. */
    const c4gpu_score *start_cells;
    c4gpu_score *end_cells;
} c4gpu_viterbi_job;

typedef struct {
    c4gpu_score score;
    int32_t query_start, target_start, query_end, target_end;  /* vd->curr_* (viterbi.c:464-478), region-relative */
    c4gpu_score final_cell[1 + C4GPU_MAX_SHADOWS + 3];          /* continuation->final_cell (viterbi.c:828-832) */
    int32_t last_srp;                                           /* vd->checkpoint->last_srp (viterbi.c:813) */
    int32_t n_ops;                                              /* FIND_PATH: raw transition path, start->end */
    int32_t *ops;                                               /* malloc'd transition ids (one per step) */
    c4gpu_score *checkpoints;                                   /* FIND_CHECKPOINTS: [cp][row][i][state][cell] */
} c4gpu_viterbi_result;

/* ---- library ----------------------------------------------------------------------------------- */

typedef struct c4gpu_ctx c4gpu_ctx;

int         c4gpu_abi_version(void);
/* The library's C4GPU_* switches (shapes kept for measurement, test hooks, fallbacks kept for A/B runs: csrc/c4_config.h) are read
 * from the environment ONCE -- when the first context opens, or at the first question -- by one function; no call path of the
 * library calls getenv (the reference reads its own environment through its ArgumentSets at start-up, argument.c, and so does
 * the drop-in, integration/c4gpu_shim.c: shim_env).  c4gpu_config_reload reads the environment again: the hook of a test that
 * flips a variable between two calls.  Not to be called while another thread is inside the library.  Returns the number of
 * C4GPU_* variables it found set. */
int         c4gpu_config_reload(void);
const char *c4gpu_last_error(void);

/* Opens HIP device `device_ordinal`.  Returns NULL (and sets last_error) when no gfx950 device or the
 * HIP runtime is unavailable: there is NO CPU fallback in this library. */
c4gpu_ctx  *c4gpu_ctx_create(int device_ordinal);
void        c4gpu_ctx_destroy(c4gpu_ctx *ctx);
/* Loads the code objects a heuristic run touches first (sequence preparation, HSP extension, word scan, SDP passes) without
 * launching anything; thread-safe beside other calls on the context: meant for a background thread of a caller that still
 * has host work to do before its first batch (integration/c4gpu_shim.c starts the context that way). */
void        c4gpu_ctx_warm(c4gpu_ctx *ctx);
/* Ends a warm-up that is running on another thread before its next code-object load and turns later ones into no-ops: a caller
 * that is about to leave the process joins its warm-up thread after this and runs its exit handlers with no thread inside the
 * HIP runtime (integration/c4gpu_shim.c, shim_quiesce). */
void        c4gpu_ctx_warm_cancel(void);
/* Use an externally owned HIP stream (e.g. torch's current stream) for all launches; NULL = default. */
void        c4gpu_ctx_set_stream(c4gpu_ctx *ctx, void *hip_stream);
/* Gives the context a non-blocking HIP stream of its own (destroyed with the context): the calls of a SECOND context of one
 * process, made from a second host thread, then run beside the first one's instead of in the default stream's order
 * (integration/c4gpu_sdp.c: a batch of SDP passes beside the word scans and HSP extensions of the comparisons behind it).
 * 0, or -1 with last_error set. */
int         c4gpu_ctx_own_stream(c4gpu_ctx *ctx);
/* Takes the arena the SDP passes write their step records to (c4gpu_sdp_batch) NOW, `bytes` large (at most 0.6 of the free device
 * memory), and keeps it with the context between batches instead of giving a large one back after each: a context that serves
 * nothing but SDP batches (the flight context of integration/c4gpu_sdp.c) allocates once, ahead of its first batch and beside the
 * caller's other work -- the first hipMalloc of tens of GB in a process takes 0.3 ms or, after another process has just given
 * memory back, 1.7-3 s (profiles/r05_c5_cold.md).  A batch that needs more still grows it.  bytes <= 0: give it back, batches
 * allocate for themselves again.  Not beside a running batch of the same context.  0, or -1 with last_error set. */
int         c4gpu_ctx_sdp_reserve(c4gpu_ctx *ctx, int64_t bytes);
int         c4gpu_ctx_device_info(c4gpu_ctx *ctx, char *name, size_t name_len, int *n_cu, int64_t *mem_bytes);

/* Defaults of the reference's ArgumentSets: nucleic / blosum62 matrices, standard genetic code,
 * primate splice PSSMs, -12/-4, -18/-8, intron 30..200000/-30, frameshift -28. */
void        c4gpu_params_default(c4gpu_params *out);
/* --forcegtag yes|no: SplicePredictor_GTAGonly_create (splice.c:213-240) for the four site types. */
void        c4gpu_params_set_forcegtag(c4gpu_params *params, int on);

/* Model_Type_get_model (modeltype.c): "affine:local", "affine:global", "affine:bestfit",
 * "affine:overlap", "ungapped", "est2genome", "protein2dna", "protein2dna:bestfit", "protein2genome",
 * "protein2genome:bestfit".  Returns 0 on success. */
int         c4gpu_model_get(const char *model_type, int query_alphabet, int target_alphabet,
                            const c4gpu_params *params, c4gpu_model *out);
/* C4_DerivedModel_create (c4.c:2292-2337) on the model of that type: the closed sub-model of every path from
 * state src_state to state dst_state (state ids of the original model; 0 = START, 1 = END), with the given
 * scopes — BSDP's join models (match state -> match state, CORNER/CORNER, heuristic.c:255) and terminal
 * models (START -> match state with the model's start scope / CORNER; match state -> END, heuristic.c:301-309).
 * transition_map (may be NULL, C4GPU_MAX_TRANSITIONS entries) receives per derived transition id the original
 * transition id.  Returns 0, or -1 when the type is unknown or no path exists. */
int         c4gpu_model_get_derived(const char *model_type, int query_alphabet, int target_alphabet,
                                    const c4gpu_params *params, int src_state, int dst_state,
                                    int start_scope, int end_scope, c4gpu_model *out, int32_t *transition_map);
/* Viterbi_create's continuation copy: same tables, CORNER/CORNER scopes (viterbi.c:68-76). */
void        c4gpu_model_make_continuation(const c4gpu_model *model, c4gpu_model *out);
/* Codegen_clean_path_component("optimal:<name> find <what>") — the Bootstrapper_lookup key
 * (codegen.c:39-55, optimal.c:31-67).  Returns length written. */
int         c4gpu_model_plugin_name(const c4gpu_model *model, int mode, int use_continuation,
                                    char *buf, size_t buf_len);
/* 1 when every calc kind / shadow of `model` is implemented by the device engine. */
int         c4gpu_model_is_accelerated(const c4gpu_model *model);

/* Index of the compiled device family whose closed tables equal `model`'s (the device-side counterpart of
 * Bootstrapper_lookup finding a compiled function, bootstrapper.c:85-90), or -1.  Host-only: needs no device. */
int         c4gpu_model_device_family(const c4gpu_model *model);

/* Viterbi_use_reduced_space (viterbi.c:128-150) and Viterbi_checkpoint_rows (viterbi.c:207-218):
 * identical decisions to the reference for a given --dpmemory (Mb). */
int         c4gpu_use_reduced_space(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb);
int         c4gpu_checkpoint_rows(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb);
/* The same two decisions evaluated ON THE DEVICE for n region sizes (the sub-alignment jobs of a checkpoint pass are
 * listed there, c4_engine.hip fuse_expand_kernel: the rule's arithmetic is compiled for host and device from one source,
 * csrc/c4_memrule.h): reduced[k] = Viterbi_use_reduced_space, rows[k] = Viterbi_checkpoint_rows of a region of
 * query_length[k] x target_length[k].  Test hook: must equal the host functions above for every size. */
int         c4gpu_memrule_device(c4gpu_ctx *ctx, const c4gpu_model *model, int dpmemory_mb, const int32_t *query_length,
                                 const int32_t *target_length, int32_t n, int32_t *reduced, int32_t *rows);

/* SplicePredictor_predict_array_int (splice.c:383-397) for the 4 splice types over whole targets, on
 * the device.  out[k] (k = C4GPU_SS*) receives target_len int32 each. */
int         c4gpu_splice_predict(c4gpu_ctx *ctx, const c4gpu_params *params,
                                 const uint8_t *target, int32_t target_len, int32_t *out[4]);

/* SubOpt_create / _destroy / _add_alignment (subopt.c:24-148).  add_alignment blocks the cells every
 * MATCH-labelled operation of `a` leaves from (and the lead-in cells of multi-residue matches), skipping
 * cells already present.  add_point is for callers that hold the reference's own SubOpt / SubOpt_Index
 * and only mirror it.  points() returns the total and writes up to `max` points sorted
 * by target then query — the order SubOpt_Index_create (subopt.c:250) sorts them in. */
c4gpu_subopt *c4gpu_subopt_create(int32_t query_length, int32_t target_length);
void        c4gpu_subopt_destroy(c4gpu_subopt *subopt);
int         c4gpu_subopt_add_alignment(c4gpu_subopt *subopt, const c4gpu_model *model, const c4gpu_alignment *a);
int         c4gpu_subopt_add_point(c4gpu_subopt *subopt, int32_t query_pos, int32_t target_pos);
int32_t     c4gpu_subopt_points(const c4gpu_subopt *subopt, int32_t *query_pos, int32_t *target_pos, int32_t max);

/* Viterbi_DP_Func (viterbi.h:95-98) for a batch of independent jobs in one mode.  A job's `subopt` plays
 * the part of the soi argument: the device evaluates SubOpt_Index_is_blocked for the job's region. */
int         c4gpu_viterbi_batch(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                int mode, const c4gpu_pair *pairs, int32_t n_pairs,
                                const c4gpu_viterbi_job *jobs, int32_t n_jobs,
                                c4gpu_viterbi_result *results);
void        c4gpu_viterbi_result_clear(c4gpu_viterbi_result *r);

/* Optimal_find_score (optimal.c:123) over full pair rectangles. */
int         c4gpu_optimal_find_score_batch(c4gpu_ctx *ctx, const c4gpu_model *model,
                                           const c4gpu_params *params,
                                           const c4gpu_pair *pairs, int32_t n_pairs,
                                           c4gpu_score *scores);
/* Optimal_find_path (optimal.c:368): region -> checkpoints -> sub-alignments, same orchestration and
 * the same memory decisions as the reference at `dpmemory_mb`; alignments[i].valid = 0 when the score
 * is below `threshold`. */
int         c4gpu_optimal_find_path_batch(c4gpu_ctx *ctx, const c4gpu_model *model,
                                          const c4gpu_params *params,
                                          const c4gpu_pair *pairs, int32_t n_pairs,
                                          int dpmemory_mb, c4gpu_score threshold,
                                          c4gpu_alignment *alignments);
/* The same with sub-optimal blocking: subopts[i] (may be NULL) is pair i's SubOpt, as the `subopt`
 * argument of Optimal_find_path (optimal.c:368); active[i] == 0 (active may be NULL = all) skips pair i.
 * One round of GAM_Result_exhaustive_create's do/while loop (gam.c:1158-1172) for every pair at once. */
int         c4gpu_optimal_find_path_batch_subopt(c4gpu_ctx *ctx, const c4gpu_model *model,
                                                 const c4gpu_params *params,
                                                 const c4gpu_pair *pairs, int32_t n_pairs,
                                                 int dpmemory_mb, c4gpu_score threshold,
                                                 const c4gpu_subopt *const *subopts, const uint8_t *active,
                                                 c4gpu_alignment *alignments);
void        c4gpu_alignment_clear(c4gpu_alignment *a);

/* Does a pair of this size pass the guard of the packed 16-bit passes (every score a path can reach
 * fits 16 000, the intron length test cannot fail on the upper side) under this model and these parameters?  1 / 0; -1: the
 * model is not device-accelerated.  Host-only, no device needed: what a caller (or a test) can ask to know which kernels a
 * launch will take; the results are the reference's either way. */
int         c4gpu_packed_route_fits(const c4gpu_model *model, const c4gpu_params *params, int32_t query_length,
                                    int32_t target_length);
/* Host only: for which states of (model, params) a path sub-alignment WITHOUT a query row that starts and ends in the state is
 * answered without a dynamic programme -- the state's loop over one target column, repeated (csrc/c4_viterbi_kernel.h,
 * KParams::loop_tr).  loop_transition[s] = the loop's transition id, or -1 where the parameters do not prove that nothing else
 * can score as much: the state must be entered and left through splice calcs only, every other transition that moves without a
 * query row must cost (constant <= 0), and the state's best 3' site + best 5' site + opening constant must be negative
 * (13 + 15 - 30 under exonerate's defaults: the best 5' site sums to 13.09, the best 3' site to 15.50, rounded as the predictor rounds; optimal.c:266-313 runs a Viterbi over such a section like over any other).
 * Returns the number of states with a loop, or -1 (last_error). */
int         c4gpu_loop_sections(const c4gpu_model *model, const c4gpu_params *params, int32_t *loop_transition, int32_t n_states);

/* Device-resident batches (bench / shim hot loop): upload once, run many times. */
typedef struct c4gpu_batch c4gpu_batch;
c4gpu_batch *c4gpu_batch_create(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                const c4gpu_pair *pairs, int32_t n_pairs);
void         c4gpu_batch_destroy(c4gpu_batch *b);
/* One pass of the hot path over the resident batch. `what`: 0 = score pass only (FIND_SCORE),
 * 1 = region pass only, 2 = full Optimal_find_path.  Results stay on the object. */
int          c4gpu_batch_run(c4gpu_batch *b, int what, int dpmemory_mb, c4gpu_score threshold);
/* Optimal_find_path (optimal.c:368) of every pair with active[i] != 0 (NULL = all) over ITS OWN region of the pair's
 * rectangle instead of the whole of it: the call GAM_Result_refine_alignment makes under --refine region (the
 * heuristic alignment's bounding box grown by --refineboundary, gam.c:618-640).  Results as after
 * c4gpu_batch_run(b, 2, ...): c4gpu_batch_alignment; alignment regions are in sequence coordinates. */
int          c4gpu_batch_run_regions(c4gpu_batch *b, const c4gpu_region *regions, const uint8_t *active,
                                     int dpmemory_mb, c4gpu_score threshold);
/* --annotation (src/comparison/match.c:276-281, Match_1_1_dna_score_func): a DNA query that carries a CDS annotation
 * (Sequence_Annotation: cds_start, cds_length, sequence.h:49-54) may not take part in a 1:1 DNA match inside its CDS -- the match
 * scores MATCH_IMPOSSIBLY_LOW_SCORE there.  cds_start[i] / cds_length[i]: the annotation of pair i's query (length <= 0: none;
 * both NULL: no pair has one).  Applies to the sequences the batch holds NOW (after c4gpu_batch_create or
 * c4gpu_batch_swap_stage; call it again after the next swap) and to every run that follows: annotated positions carry a matrix
 * row of their own, the batch's passes take the kernels that keep every validity mask and the 32-bit arithmetic (a score of
 * -987654321 is outside every packed-pass guard), one launch lane.  Models without a 1:1 DNA match calc are not affected (the
 * reference's other match functions have their own, coding-model rules: match.c:490-546, not on the accelerated path). */
int         c4gpu_batch_set_annotation(c4gpu_batch *b, const int32_t *cds_start, const int32_t *cds_length);

/* Per-pair score thresholds for the full runs (what = 2) and c4gpu_batch_next_paths: what
 * GAM_get_query_threshold gives a query under --percent (gam.c:466-487,677-705; never below --score).  A pair
 * is held to max(threshold argument, per_pair[i]).  NULL switches them off. */
int          c4gpu_batch_set_thresholds(c4gpu_batch *b, const c4gpu_score *per_pair);
/* c4gpu_viterbi_batch on the pairs already resident (jobs[i].pair indexes them): what a caller that makes
 * many Viterbi_DP_Func calls on the same pair uses to upload it once. */
int          c4gpu_batch_viterbi(c4gpu_batch *b, int mode, const c4gpu_viterbi_job *jobs, int32_t n_jobs,
                                 c4gpu_viterbi_result *results);
/* The same for ANOTHER model on the batch's resident pairs: BSDP's derived models (terminal, join, span src / dst;
 * heuristic.c:242-330,461-472) of the batch's model all read the residue codes and splice arrays the batch already
 * holds, and a heuristic run asks for all of them on every pair (sar.c:393,697,898).  The model's engine is set up on
 * first use and kept with the batch.  Fails when `model` needs arrays the batch's own model does not. */
int          c4gpu_batch_viterbi_model(c4gpu_batch *b, const c4gpu_model *model, int mode,
                                       const c4gpu_viterbi_job *jobs, int32_t n_jobs, c4gpu_viterbi_result *results);
/* The sub-optimal loop on the resident batch: after c4gpu_batch_run(b, 2, ...), each call blocks the
 * alignments found so far (SubOpt_add_alignment, gam.c:673) and finds the next best path of every pair
 * that still had one in the previous round; pairs whose score drops below `threshold` leave the loop.
 * Returns the number of alignments found in this round (0 = loop finished), -1 on error.
 * c4gpu_batch_alignment then returns this round's alignment (valid = 0 for pairs that have left). */
int          c4gpu_batch_next_paths(c4gpu_batch *b, int dpmemory_mb, c4gpu_score threshold);
int          c4gpu_batch_scores(c4gpu_batch *b, c4gpu_score *scores, c4gpu_region *regions);
int          c4gpu_batch_alignment(c4gpu_batch *b, int32_t i, c4gpu_alignment *out);
/* Every alignment the batch holds after c4gpu_batch_run, as one int32 stream: first one row of 7 ints per pair (valid,
 * score, query_start, target_start, query_length, target_length, n_ops), then the (transition, length) pairs of all valid
 * alignments in pair order.  Returns the number of ints the stream takes; it is written only when that fits `cap`
 * (out = NULL sizes the buffer).  What a rank of a sharded run hands to the result gather (bench.py --gpus N). */
int64_t      c4gpu_batch_export(c4gpu_batch *b, int32_t *out, int64_t cap);
/* Accumulated device time (ms), launches and lattice cells of the Viterbi kernel of one mode
 * (C4GPU_MODE_*) since the last reset, measured with HIP events on the launch stream.  The first call
 * switches the measurement on. */
int          c4gpu_batch_kernel_stats(c4gpu_batch *b, int mode, int reset, double *ms, int64_t *launches,
                                      int64_t *cells);

/* A stream of batches: the NEXT batch is staged while the current one is aligned.  What the reference pays per pair before
 * its first DP cell -- Sequence_strncpy of both sequences (src/sequence/sequence.c:588) and the splice site prediction of
 * the target (Intron_Data / SplicePredictor_predict_array_int, src/model/intron.c:259-269, src/sequence/splice.c) -- is,
 * per batch, here: gathering the residues into page-locked host memory, the copy over PCIe, residue coding and the splice
 * arrays built on the device.  A c4gpu_stage does that on a stream of its own from whatever thread calls
 * c4gpu_stage_load, while another thread is inside c4gpu_batch_run on the same context; c4gpu_batch_swap_stage then hands
 * the loaded sequences to the batch (which keeps its engine, launch lanes and launch buffers) and the batch's previous
 * sequences to the stage, whose next load reuses their device arrays and page-locked buffers: no allocation, no page
 * fault and no pageable copy per batch after the first two.
 *   c4gpu_stage_load   blocks until the batch is resident (0 / -1, c4gpu_last_error); the pairs' buffers are read during
 *                      the call only.  One load at a time per stage; never concurrently with c4gpu_batch_swap_stage.
 *   c4gpu_batch_swap_stage  between two runs of `b` (not while one is in progress); the batch's earlier results are
 *                      dropped.  Model and parameters of stage and batch must be bytewise equal.
 *   c4gpu_stage_load_ms     wall time of the last load (host clock around the whole call). */
typedef struct c4gpu_stage c4gpu_stage;
c4gpu_stage *c4gpu_stage_create(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params);
int          c4gpu_stage_load(c4gpu_stage *st, const c4gpu_pair *pairs, int32_t n_pairs);
double       c4gpu_stage_load_ms(const c4gpu_stage *st);
int          c4gpu_batch_swap_stage(c4gpu_batch *b, c4gpu_stage *st);
void         c4gpu_stage_destroy(c4gpu_stage *st);

/* ---- HSP seeding (src/comparison/hspset.c) ------------------------------------------------------------------------ */

/* Match_Type of an HSPset (src/comparison/match.h:45-51): what one HSP position scores.  DNA2DNA: dna submat[q][t];
 * PROTEIN2PROTEIN: protein submat[q][t]; PROTEIN2DNA: protein submat[q][aa(t, t+1, t+2)], target advance 3. */
enum { C4GPU_MATCH_DNA2DNA = 0, C4GPU_MATCH_PROTEIN2PROTEIN = 1, C4GPU_MATCH_PROTEIN2DNA = 2 };

/* one word hit handed to HSPset_seed_hsp (hspset.c:933): positions in pair `pair` */
typedef struct { int32_t pair, query_start, target_start; } c4gpu_hsp_seed;
/* the HSP that seed grows into: HSP_trim_ends, HSP_init and HSP_extend without masking (hspset.c:737-812,837-870);
 * length in match-state visits, cobs = HSP_find_cobs (hspset.c:426-441) */
typedef struct { int32_t query_start, target_start, length, score, cobs; } c4gpu_hsp;

/* The ungapped X-drop extension of every seed, on the device: what HSPset_seed_hsp computes for a seed its horizon lets
 * through (the horizon, the threshold and the store order stay with the caller: they need the HSPs in seed order).
 * seedlen / dropoff: HSP_Param.seedlen / .dropoff.  Fails (-1) on residues outside the submat alphabet. */
int         c4gpu_hsp_extend_batch(c4gpu_ctx *ctx, const c4gpu_params *params, int match_type,
                                   const c4gpu_pair *pairs, int32_t n_pairs, int32_t seedlen, int32_t dropoff,
                                   const c4gpu_hsp_seed *seeds, int32_t n_seeds, c4gpu_hsp *out);
/* The same with the diagonal horizon applied on the device (hspset.c:952-958,990): chain[k] names the horizon entry seed k
 * is tested against and updates -- one per (HSPset, diagonal section, query frame, target frame) -- and horizon0[c] is that
 * entry's value before the scan.  The seeds of a chain are taken in index order by one lane: a seed whose target_start lies
 * below the chain's horizon is NOT extended (out[k].length = -1), every other one is, and moves the horizon to its HSP's
 * target end, exactly as HSPset_seed_hsp does seed by seed.  A long identical diagonal then costs one extension instead of one
 * per word hit.  The caller's replay makes the same decisions from the same numbers and never reads a skipped entry. */
int         c4gpu_hsp_extend_chains(c4gpu_ctx *ctx, const c4gpu_params *params, int match_type,
                                    const c4gpu_pair *pairs, int32_t n_pairs, int32_t seedlen, int32_t dropoff,
                                    const c4gpu_hsp_seed *seeds, int32_t n_seeds, const int32_t *chain, int32_t n_chains,
                                    const int32_t *horizon0, c4gpu_hsp *out);


/* The seeder's word scan (Seeder_add_target -> FSM_traverse / Seeder_VFSM_traverse_single -> Seeder_FSM_traverse_func,
 * src/comparison/seeder.c:649-720,852-915; src/struct/fsm.c:186-198).  A word table holds the words the queries put into
 * the seeder's automaton — all of one length `wordlen` — as codes over the automaton's columns: code = sum of
 * column[i] * width^(wordlen - 1 - i), columns 1 .. width - 1 (column 0: a symbol outside the alphabet, which resets the
 * automaton).  Word k reports emissions emit_first[k] .. emit_first[k + 1] - 1 (the caller's list: the word's own seeds,
 * then its neighbours' seeds, seeder.c:676-692).  c4gpu_seed_scan walks one symbol string (a target, or one translated
 * frame of it, mapped through the automaton's traversal filter) and returns every word hit in the reference's order:
 * position of the word's LAST symbol ascending, then emission order.  hits: room for `cap` hits; *n_hits: how many there
 * are (when that exceeds cap nothing is written: call again with room for all of them). */
typedef struct c4gpu_wordtab c4gpu_wordtab;
typedef struct { int32_t pos, emit; } c4gpu_word_hit;
c4gpu_wordtab *c4gpu_wordtab_create(c4gpu_ctx *ctx, int32_t width, int32_t wordlen, const uint64_t *codes,
                                    const int32_t *emit_first, int32_t n_words);
void        c4gpu_wordtab_destroy(c4gpu_wordtab *t);
int         c4gpu_seed_scan(c4gpu_ctx *ctx, c4gpu_wordtab *t, const uint8_t *symbols, int32_t n, c4gpu_word_hit *hits,
                            int64_t cap, int64_t *n_hits);
/* device time of the scans of a table (HIP events), their number, symbols scanned, hits found */
void        c4gpu_wordtab_stats(const c4gpu_wordtab *t, double *scan_ms, int64_t *scans, int64_t *symbols, int64_t *hits);

/* SDP, the default gapped-extension heuristic (SDP_Pair_next_path src/sdp/sdp.c:743 in the loop of GAM_Result_SDP_create
 * src/hub/gam.c:852-890), for every pair of a batch, in the flavour SDP_create picks for the model (sdp.c:322-366):
 * bidirectional from the seeds (no shadows, no spans, one match transition: the affine and protein2dna families) or
 * boundary + span freeze / thaw (est2genome, protein2genome).  -1 for models outside those (spans on the query axis).
 * hsps: the HSPs of all pairs, pair i's are [hsp_first[i], hsp_first[i+1]) in the order SDP_Pair_create_seed_list meets
 * them (sdp.c:447-463); query_advance / target_advance: the match advances of the HSPset (1 / 1, or 1 / 3 for
 * protein2dna); dropoff: --extensionthreshold.  Both passes of Scheduler_Pair_calculate (scheduler.c:1445) run on the
 * device for all pairs at once — one wavefront per pair that visits, like the reference's scheduler, only the cells
 * inside the X-drop (memory and time follow the visited cells, not the lattice sizes) — and the --singlepass yes loop
 * over the seeds follows on the host.  out[i * max_alignments + k] is pair i's k-th alignment (clear each with
 * c4gpu_alignment_clear), n_out[i] their number, or -1 for a pair the device could not serve (its traceback did not fit
 * the memory left on the device): the caller runs the reference's SDP for that pair; the others are unaffected. */
/* The cells of the box around a pair's HSPs widened by 384 positions: an upper bound of what the passes visit when no
 * extension runs past that margin (for callers that plan by size; c4gpu_sdp_batch itself has no size limit): */
double      c4gpu_sdp_lattice_cells(const c4gpu_hsp *hsps, int32_t n_hsps, int32_t query_advance, int32_t target_advance,
                                    int32_t query_len, int32_t target_len);
/* Counters of the calling thread's c4gpu_sdp_batch calls (kernel ms of the passes and of the walks by HIP events,
 * staging and host ms, jobs launched, pairs run again with a larger arena, pairs not served); reset != 0 clears them. */
void        c4gpu_sdp_stats(int reset, double *pass_ms, double *walk_ms, double *stage_ms, double *host_ms, int64_t *jobs,
                            int64_t *reruns, int64_t *unserved);
int         c4gpu_sdp_batch(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                            const c4gpu_pair *pairs, int32_t n_pairs, const c4gpu_hsp *hsps, const int32_t *hsp_first,
                            int32_t query_advance, int32_t target_advance, int32_t dropoff, c4gpu_score threshold,
                            int32_t max_alignments, c4gpu_alignment *out, int32_t *n_out);

/* Alignment_print_{sugar,cigar,vulgar}_block (alignment.c:1622-1779); coordinates are region
 * coordinates on the given strands ('+', '-', '.'), flipped to the forward strand when
 * forward_coords != 0 as --forwardcoordinates does (alignment.c:177-205).  `what`: 0 sugar, 1 cigar,
 * 2 vulgar.  Returns length written (excluding NUL) or -1. */
int         c4gpu_alignment_format(const c4gpu_model *model, const c4gpu_alignment *a, int what,
                                   const char *query_id, int32_t query_len, char query_strand,
                                   const char *target_id, int32_t target_len, char target_strand,
                                   int forward_coords, char *buf, size_t buf_len);

/* Alignment_display_gff (alignment.c:2710-3236): the GFF2 dump of --showtargetgff / --showquerygff -- header,
 * gene / utr / cds / exon / intron / splice features where the report side is genomic (gam.c:1224-1230:
 * report_on_genomic = Model_Type_has_genomic_target for the target report, never for the query report), similarity line --
 * as the reference prints it, byte for byte.  Needs the residues (identity / similarity of the gene and exon attributes:
 * alignment.c:1382-1560, splice-site dinucleotides) and the scoring data (similarity = match calc > 0; codons through
 * Translate_base).  The sequences are the ones that were aligned (a reverse-complemented strand as such); a strand of
 * '-' turns the coordinates back as Alignment_display_gff_line does.  Host-only. */
typedef struct {
    const char    *query_id, *target_id;
    const uint8_t *query, *target;
    int32_t        query_len, target_len;
    char           query_strand, target_strand;   /* '+', '-' or '.' (Sequence_get_strand_as_char) */
    int32_t        report_on_query;               /* --showquerygff: 1; --showtargetgff: 0 */
    int32_t        report_on_genomic;             /* gene features before the similarity line */
    int32_t        result_id;                     /* gene_id / alignment_id */
    const char    *date;                          /* "YYYY-MM-DD" of the ##date line; NULL: today, local time */
    const char    *version;                       /* VERSION of the ##source-version line; NULL: "2.4.0" */
} c4gpu_gff_request;
/* returns the length written (without the terminating NUL), or -(length needed + 1) when buf_len is too small, or
 * INT32_MIN for a model whose match transitions are not one of the accelerated kinds */
int         c4gpu_alignment_format_gff(const c4gpu_model *model, const c4gpu_params *params, const c4gpu_alignment *a,
                                       const c4gpu_gff_request *req, char *buf, size_t buf_len);

/* Alignment_display (alignment.c:234-1380): the human-readable block of --showalignment yes, the reference's
 * default output -- header, then rows of query / [translation] / match line / [translation] / target with the running
 * coordinates, gaps, codon gaps, frameshifts, splice sites with their consensus marks, collapsed introns, split codons and
 * the reverse-translation marks of protein-vs-DNA matches (match.c:224-236,385-417) -- byte for byte, for the models the
 * library accelerates (1:1 DNA or protein matches, protein against DNA / genome).  Host-only. */
typedef struct {
    const char    *query_id, *query_def, *target_id, *target_def;    /* def: the rest of the FASTA header line, or NULL */
    const uint8_t *query, *target;                                   /* as aligned (a reverse-complemented strand as such) */
    int32_t        query_len, target_len;
    char           query_strand, target_strand;
    int32_t        width;                 /* --alignmentwidth; 0: 80 */
    int32_t        forward_coords;        /* --forwardcoordinates (default in the reference: 1) */
    int32_t        use_aa_tla;            /* --useaatla (default in the reference: 1): Ala / ^A^ */
} c4gpu_display_request;
/* returns as c4gpu_alignment_format_gff */
int         c4gpu_alignment_display(const c4gpu_model *model, const c4gpu_params *params, const c4gpu_alignment *a,
                                    const c4gpu_display_request *req, char *buf, size_t buf_len);

/* Alignment_display_ryo (alignment.c:1781-2669): a --ryo format string printed for one alignment -- every token
 * of the reference: %[qt][idlsSt], %[qt]a[bels], %[qt]c[bels], %s %m %r, %p[cIisS], %e[tism], %g, %S %C %V, the escapes, and
 * the per-transition section { ... } with %P[qt][sabe] and %P[nsl] (transition scores through the calcs: match, constants,
 * splice sites with the intron-length test on the tracked shadow, split codons; the splice predictions are computed on the
 * host here).  Not covered: %pS for protein-against-DNA models (the reference scores the query against itself through a
 * codon match there) -- INT32_MIN, as for an unknown token or unbalanced braces (the reference aborts). */
typedef struct {
    const char    *query_id, *query_def, *target_id, *target_def;
    const uint8_t *query, *target;
    int32_t        query_len, target_len;
    char           query_strand, target_strand;
    int32_t        forward_coords;        /* --forwardcoordinates */
    int32_t        rank;                  /* %r; -1 prints the placeholder the reference's --bestn pass fills in */
    const char    *format;
} c4gpu_ryo_request;
int         c4gpu_alignment_format_ryo(const c4gpu_model *model, const c4gpu_params *params, const c4gpu_alignment *a,
                                       const c4gpu_ryo_request *req, char *buf, size_t buf_len);




#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_C4GPU_H */
