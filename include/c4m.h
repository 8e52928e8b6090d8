/* c4m.h — host-side model builder: the C4_Model plug-in surface (src/c4/c4.h:198-356) re-implemented
 * as a small C API that produces the flattened c4gpu_model tables the device engine consumes.
 *
 * Same operations, same argument meaning and same closing semantics as the reference:
 *   c4m_model_create / open / close      C4_Model_create c4.c:510, C4_Model_open :1300, C4_Model_close :1669
 *   c4m_add_state/calc/transition/shadow C4_Model_add_* c4.c:344-508
 *   c4m_make_stereo / c4m_insert         C4_Model_make_stereo c4.c:681, C4_Model_insert c4.c:963
 *   c4m_configure_start/end_state        c4.c:1059,1083
 * close() = id assignment + C4_Model_topological_sort (c4.c:1418: emitting transitions in reverse list
 * order, then silent ones producers-first) + shadow designation packing (c4.c:1638) + max advances.
 * Handles are small integers; state handles equal state ids (states are never reordered), transition
 * handles are creation indices and stay valid across close() (ids change, handles do not).
 */
#ifndef INCLUDED_C4M_H
#define INCLUDED_C4M_H

#include "c4gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c4m_model c4m_model;

#define C4M_START (-1)   /* NULL input  => START state (c4.c:437) */
#define C4M_END   (-1)   /* NULL output => END state   (c4.c:439) */

c4m_model *c4m_model_create(const char *name);             /* created open */
void       c4m_model_destroy(c4m_model *m);
void       c4m_model_rename(c4m_model *m, const char *name);
void       c4m_model_open(c4m_model *m);
int        c4m_model_close(c4m_model *m);                  /* 0 ok, -1 invalid (cycle, orphan state) */
int        c4m_model_is_open(const c4m_model *m);
void       c4m_model_set_alphabets(c4m_model *m, int query_alphabet, int target_alphabet);

int  c4m_add_state(c4m_model *m, const char *name);
int  c4m_add_calc(c4m_model *m, const char *name, int kind, int value, int param,
                  int max_score, int protect);
int  c4m_add_transition(c4m_model *m, const char *name, int input_state, int output_state,
                        int advance_query, int advance_target, int calc, int label);
/* src_state C4M_START => START; dst_transition -1 => every transition into END (c4.c:467-479) */
int  c4m_add_shadow(c4m_model *m, const char *name, int src_state, int dst_transition, int on_target);
void c4m_shadow_add_src_state(c4m_model *m, int shadow, int state);
void c4m_shadow_add_dst_transition(c4m_model *m, int shadow, int transition);

void c4m_configure_start_state(c4m_model *m, int scope);
void c4m_configure_end_state(c4m_model *m, int scope);

void c4m_make_stereo(c4m_model *m, const char *suffix_a, const char *suffix_b);
/* insert a CLOSED model between two states of an OPEN one; -1/-1 = START/END */
int  c4m_insert(c4m_model *target, const c4m_model *insert, int src_state, int dst_state);

/* C4_DerivedModel_create (c4.c:2292-2337, on C4_Model_select c4.c:2217): the closed sub-model of every path
 * from src_state to dst_state of a CLOSED model (C4M_START / C4M_END allowed), with the given scopes — BSDP's
 * join and terminal models (heuristic.c:242-330); named Segment("src"->"dst"):[model].  transition_map (may
 * be NULL) receives, per derived transition id, the id of the original transition (C4_DerivedModel's
 * transition_map).  NULL when no path exists.  cell_start/cell_end callbacks (span models) are not modelled. */
c4m_model *c4m_derive(const c4m_model *model, int src_state, int dst_state, int start_scope, int end_scope,
                      int *transition_map, int transition_map_len);

/* queries on the (open or closed) model */
int  c4m_select_single_transition(const c4m_model *m, int label);   /* handle or -1 */
int  c4m_select_transitions(const c4m_model *m, int label, int *handles, int max);
int  c4m_transition_input(const c4m_model *m, int transition);
int  c4m_transition_output(const c4m_model *m, int transition);
int  c4m_transition_id(const c4m_model *m, int transition);          /* id after close */

/* flatten a CLOSED model */
int  c4m_flatten(const c4m_model *m, c4gpu_model *out);

/* the reference's model constructors, on top of the builder (for callers that want the c4m object) */
c4m_model *c4m_ungapped_create(int query_alphabet, int target_alphabet, const c4gpu_params *p);  /* ungapped.c:122 */
c4m_model *c4m_affine_create(int scope_type, int query_alphabet, int target_alphabet,
                             const c4gpu_params *p);                                           /* affine.c:150  */
c4m_model *c4m_intron_create(const char *suffix, int is_forward, const c4gpu_params *p);        /* intron.c:588 (target introns) */
c4m_model *c4m_est2genome_create(const c4gpu_params *p);                                        /* est2genome.c:58 */
c4m_model *c4m_protein2dna_create(int scope_type, const c4gpu_params *p);                       /* protein2dna.c:56 */
c4m_model *c4m_phase_create(const c4gpu_params *p);                                             /* phase.c:354 (protein query, target introns) */
c4m_model *c4m_protein2genome_create(int scope_type, const c4gpu_params *p);                    /* protein2genome.c:44 */

/* Affine_Model_Type, src/model/affine.h */
enum { C4M_AFFINE_GLOBAL = 0, C4M_AFFINE_BESTFIT, C4M_AFFINE_LOCAL, C4M_AFFINE_OVERLAP };

float c4gpu_splice_max_score(const c4gpu_splice_model *sp);   /* SplicePredictor_get_max_score splice.c:399 */

#ifdef __cplusplus
}
#endif
#endif
