/* c4_oracle.h — CPU restatement of the reference's C4 Viterbi path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (exonerate_amd/, libc4gpu.so) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 * It shares nothing with the product but the POD types of include/c4gpu.h.
 *
 * Pinned against the reference itself: tests/test_oracle_golden.py replays committed vectors generated
 * by oracle/_ref/refdump (the reference's own Optimal_find_score / Optimal_find_path), including the
 * reference's model known-answer tests (src/model/{affine,est2genome,protein2dna}.test.c).
 */
#ifndef INCLUDED_C4_ORACLE_H
#define INCLUDED_C4_ORACLE_H

#include "c4gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SplicePredictor_predict_array_int, src/sequence/splice.c:383-397 over a whole sequence */
void oracle_splice_predict(const c4gpu_splice_model *sp, const uint8_t *seq, int32_t len, int32_t *pred);

/* Optimal_find_score, src/c4/optimal.c:123 over the full rectangle of one pair */
c4gpu_score oracle_find_score(const c4gpu_model *model, const c4gpu_params *params,
                              const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen);

/* Optimal_find_path, src/c4/optimal.c:368 (region -> checkpoints -> sub-alignments), with the reference's
 * own memory decisions at `dpmemory_mb`.  Returns 1 and fills `out` (caller frees with
 * oracle_alignment_clear) or 0 when the score is below threshold. */
int  oracle_find_path(const c4gpu_model *model, const c4gpu_params *params,
                      const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                      int dpmemory_mb, c4gpu_score threshold, c4gpu_alignment *out);
void oracle_alignment_clear(c4gpu_alignment *a);
/* --annotation (match.c:276-281): the CDS of the query that the calls after this one align; cds_length <= 0: none */
void oracle_set_annotation(int32_t cds_start, int32_t cds_length);

/* SubOpt, src/c4/subopt.c: the blocked points of the alignments already reported for a pair (sequence
 * coordinates).  oracle_subopt_add_alignment = SubOpt_add_alignment (subopt.c:131); the _subopt variants
 * of find_path / viterbi skip MATCH-labelled transitions at blocked cells (viterbi.c:701-704) exactly as
 * SubOpt_Index_create / _set_row / _is_blocked_fast (subopt.c:250-392, subopt.h:77-80) make them. */
typedef struct oracle_subopt oracle_subopt;
oracle_subopt *oracle_subopt_create(int32_t query_length, int32_t target_length);
void    oracle_subopt_destroy(oracle_subopt *so);
void    oracle_subopt_add_alignment(oracle_subopt *so, const c4gpu_model *model, const c4gpu_alignment *a);
/* points sorted by target then query; returns the total number */
int32_t oracle_subopt_points(const oracle_subopt *so, int32_t *q, int32_t *t, int32_t max);
int  oracle_find_path_subopt(const c4gpu_model *model, const c4gpu_params *params,
                      const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                      int dpmemory_mb, c4gpu_score threshold, const oracle_subopt *subopt,
                      c4gpu_alignment *out);
/* the same over a region of the rectangle (Optimal_find_path's `region` argument; --refine region, gam.c:618-640) */
int  oracle_find_path_region(const c4gpu_model *model, const c4gpu_params *params,
                      const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                      const c4gpu_region *region, int dpmemory_mb, c4gpu_score threshold,
                      const oracle_subopt *subopt, c4gpu_alignment *out);

/* HSPset_seed_hsp's HSP for one seed with nothing in its way (src/comparison/hspset.c:933-997: HSP_trim_ends :837-870,
 * HSP_init :722-741, HSP_extend without masking :743-812) and HSP_find_cobs (:426-441) */
void oracle_hsp_extend(const c4gpu_params *params, int match_type, const uint8_t *query, int32_t qlen,
                       const uint8_t *target, int32_t tlen, int32_t seedlen, int32_t dropoff,
                       int32_t query_start, int32_t target_start, c4gpu_hsp *out);
/* one HSPset fed `n` seeds in order (seed_repeat 1, no filter): horizon test (hspset.c:952-958), threshold
 * (HSP_store :885-888), cobs at finalise.  Returns the number of HSPs written to out (at most n). */
int32_t oracle_hsp_set(const c4gpu_params *params, int match_type, const uint8_t *query, int32_t qlen,
                       const uint8_t *target, int32_t tlen, int32_t seedlen, int32_t dropoff, int32_t threshold,
                       const int32_t *seed_q, const int32_t *seed_t, int32_t n, c4gpu_hsp *out);

/* SDP (src/sdp/sdp.c:743 SDP_Pair_next_path in the loop of GAM_Result_SDP_create, gam.c:852-890) on the HSPs of one pair:
 * up to max_alignments alignments (out[] must hold that many; clear each with oracle_alignment_clear), the number found
 * is returned.  query_advance / target_advance: the match advances of the HSPset (1/1, or 1/3 for protein2dna);
 * dropoff: --extensionthreshold; singlepass: --singlepass.  *use_boundary (may be NULL): SDP.use_boundary (sdp.c:322). */
int32_t oracle_sdp(const c4gpu_model *model, const c4gpu_params *params,
                   const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                   const c4gpu_hsp *hsps, int32_t n_hsps, int32_t query_advance, int32_t target_advance,
                   int32_t dropoff, int32_t singlepass, c4gpu_score threshold, int32_t max_alignments,
                   c4gpu_alignment *out, int32_t *use_boundary);

/* one raw Viterbi call in any mode (Viterbi_interpreted, src/c4/viterbi.c:655-837); used by the parity
 * tests of c4gpu_viterbi_batch.  checkpoints (may be NULL) receives
 * [cp][row < max_target_advance][i <= Q][state][cell_size] ints; ops receives the raw transition path. */
typedef struct {
    c4gpu_score score;
    int32_t query_start, target_start, query_end, target_end;
    c4gpu_score final_cell[1 + C4GPU_MAX_SHADOWS + 3];
    int32_t last_srp;
    int32_t n_ops;
    int32_t *ops;            /* malloc'd */
    int32_t *checkpoints;    /* malloc'd */
    int32_t cell_size;
} oracle_viterbi_out;

int  oracle_viterbi(const c4gpu_model *model, const c4gpu_params *params, int mode,
                    const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                    const c4gpu_region *region, const c4gpu_continuation *continuation,
                    int checkpoint_count, oracle_viterbi_out *out);
int  oracle_viterbi_subopt(const c4gpu_model *model, const c4gpu_params *params, int mode,
                    const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                    const c4gpu_region *region, const c4gpu_continuation *continuation,
                    int checkpoint_count, const oracle_subopt *subopt, oracle_viterbi_out *out);
/* span models (BSDP): the cell_start_func / cell_end_func seam of the Viterbi (viterbi.c:728-741,793-799) as
 * matrices over the region, [(i * (T+1)) + j][cell_size]; end_cells must be initialised by the caller */
int  oracle_viterbi_span(const c4gpu_model *model, const c4gpu_params *params, int mode,
                    const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                    const c4gpu_region *region, const c4gpu_score *start_cells, c4gpu_score *end_cells,
                    oracle_viterbi_out *out);
void oracle_viterbi_out_clear(oracle_viterbi_out *out);

/* Viterbi_use_reduced_space (viterbi.c:128) / Viterbi_checkpoint_rows (viterbi.c:207) */
int  oracle_use_reduced_space(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb);
int  oracle_checkpoint_rows(const c4gpu_model *model, const c4gpu_region *region, int dpmemory_mb);

/* Alignment_display_{sugar,cigar,vulgar}, src/c4/alignment.c:2671-2706 (the --showsugar/--showcigar/
 * --showvulgar lines, built from the *_block printers alignment.c:1622-1779).  what: 0 sugar 1 cigar 2 vulgar */
int  oracle_alignment_format(const c4gpu_model *model, const c4gpu_alignment *a, int what,
                             const char *qid, int32_t qlen, char qstrand,
                             const char *tid, int32_t tlen, char tstrand,
                             int forward_coords, char *buf, size_t buf_len);

/* cells visited by the last oracle_find_path / oracle_find_score on this thread (all passes) */
int64_t oracle_cells_visited(int reset);

#ifdef __cplusplus
}
#endif
/* The seeder's automaton walk (seeder.c:649-720,852-915; c4_oracle_seed.c): the (query, query position, target position)
 * triples of the HSPset_seed_hsp calls the reference's walk makes over `symbols`, in its order. */
int64_t oracle_seed_walk(int32_t width, int32_t wordlen, int32_t n_words, const uint64_t *codes, const int32_t *seed_first,
                         const int32_t *seeds, const int32_t *nbr_first, const int32_t *nbrs, const uint8_t *symbols,
                         int32_t n_symbols, int32_t tpos_modifier, int32_t *out, int64_t cap);

#endif
