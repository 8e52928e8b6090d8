/* c4gpu_bsdp.c — the heuristic (BSDP) seam of the drop-in: speculative batch-confirm of the sub-alignment regions'
 * small DPs of MANY pairs in a few device launches.  Part of the exonerate-gpu shim (integration/Makefile), written
 * against the reference's own headers; INTEGRATION.md section 3.
 *
 * What the reference does (src/hub/gam.c:797-850, src/bsdp/{hpair,bsdp,sar}.c): per comparison it builds an HPair —
 * one BSDP node per HSP with a start and an end terminal region (SAR_Terminal), one edge per joinable HSP pair with a
 * join region (SAR_Join) or a pair of span regions (SAR_Span) — every one of them carrying an UPPER BOUND; then
 * BSDP_next_path confirms lazily, one tiny DP at a time (SAR_*_find_score -> Optimal_find_score on a derived model
 * over <= 49 x 49 cells), until the best path holds confirmed scores only.  One such DP per launch would leave a
 * 256-CU device idle, and the confirm loop is sequential per pair.
 *
 * What this file does, with the reference's code doing all the bookkeeping:
 *   1. GAM_Result_heuristic_create (gam.c:1107) only COLLECTS (gam, comparison);
 *   2. at a flush every collected pair gets a DRY RUN through the reference's own GAM_Result_heuristic_create: the
 *      fronts of BSDP_add_node / BSDP_add_edge (bsdp.c:195,301) record every candidate region as HPair creates it,
 *      and the front of BSDP_initialise (bsdp.c:520) passes an unreachable threshold on, so the dry run ends without
 *      a single DP;
 *   3. ALL candidates of ALL pairs go to the device: one resident batch of the pairs' sequences, one launch per
 *      derived model (terminals, joins, span sources with their END-cell matrices, span destinations reading the
 *      integrated matrices — Heuristic_Span_integrate itself, heuristic.c:589, runs here on the host between the two);
 *   4. every pair is REPLAYED in submission order through GAM_Result_heuristic_create: the fronts of
 *      SAR_{Terminal,Join,Span}_find_score (sar.c:393,697,898) answer from the batch — unless the region holds cells
 *      blocked by an alignment reported since (then SubOpt_Index_create is not NULL and the DP differs: the
 *      reference's own function runs) — and the fronts of the path calls likewise (shim: Optimal_find_path).
 * The confirm loop, the thresholds, the sub-optimal loop and the printing stay the reference's: output is byte-identical
 * by construction as long as each served score equals what the CPU DP returns, which tests/test_integration_gpu.py
 * checks end to end (and C4GPU_BSDP_HOST=1 checks without a device: step 3 on the host's Optimal_find_score).
 */
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

#include "bsdp.h"
#include "hpair.h"
#include "sar.h"
#include "heuristic.h"
#include "gam.h"
#include "optimal.h"
#include "modeltype.h"
#include "ungapped.h"
#include "comparison.h"

#include "c4gpu.h"
#include "c4gpu_shim.h"

extern gint BSDP_add_node_cpu(BSDP *bsdp, gpointer node_data, C4_Score node_score, gboolean is_valid_start,
                              gboolean is_valid_end, C4_Score start_bound, C4_Score end_bound);
extern void BSDP_add_edge_cpu(BSDP *bsdp, gpointer edge_data, gint src_node_id, gint dst_node_id, C4_Score bound_score);
extern void BSDP_initialise_cpu(BSDP *bsdp, C4_Score threshold);
extern BSDP_Path *BSDP_next_path_cpu(BSDP *bsdp, C4_Score threshold);
extern C4_Score SAR_Terminal_find_score_cpu(SAR_Terminal *sar_terminal, Optimal *optimal, HPair *hpair);
extern C4_Score SAR_Join_find_score_cpu(SAR_Join *sar_join, HPair *hpair);
extern C4_Score SAR_Span_find_score_cpu(SAR_Span *sar_span, HPair *hpair);
extern GAM_Result *GAM_Result_heuristic_create_cpu(GAM *gam, Comparison *comparison);
extern Alignment *Optimal_find_path_cpu(Optimal *optimal, Region *region, gpointer user_data,
                                        C4_Score threshold, SubOpt *subopt);

/* hpair.c:22-27,60-63: the node / edge payloads HPair hands to BSDP (file-local types there) */
typedef struct { Heuristic_Match *match; gint hsp_id; SAR_Terminal *sar_start; SAR_Terminal *sar_end; } ShimNodeData;
typedef struct { SAR_Join *sar_join; SAR_Span *sar_span; } ShimEdgeData;

/* one candidate DP: a terminal or join region under `optimal`, or a span (src region under span->src_optimal, dst
 * region under span->dst_optimal) */
typedef struct {
    Optimal *optimal;              /* terminals, joins: the DP's Optimal; spans: span->dst_optimal (the key) */
    Heuristic_Span *span;          /* spans only */
    gint r[8];                     /* region, and the dst region of a span (else zeros) */
    C4_Score raw;                  /* Optimal_find_score's result, before the HSP components are taken off */
    /* the path of a terminal / join DP (Optimal_find_path), START -> END, one transition id of the derived model per step */
    gint n_ops; gint *ops; gint path_region[4];
    gboolean have_path;
} ShimSar;

typedef struct {
    GAM *gam;
    Comparison *comparison;
    GArray *sars;                  /* ShimSar, in creation order */
    GHashTable *index;             /* (optimal, regions) -> ShimSar* */
    /* --refine: the pair's FIRST refinement (GAM_Result_refine_alignment, gam.c:605-655) is part of a batch too */
    gboolean refine_seen, refine_have, refine_used;
    gint refine_region[4];
    c4gpu_alignment refined;
} ShimHPending;

enum { BSDP_OFF = 0, BSDP_COLLECT, BSDP_REFINE_COLLECT, BSDP_REPLAY };
static gint bsdp_mode = BSDP_OFF;
static ShimHPending *bsdp_cur = NULL;
static GPtrArray *bsdp_pending = NULL;
static struct { long pairs, candidates, spans, score_calls, score_served, path_calls, path_served, flushes,
                     refine_calls, refine_served;
                double device_ms, dry_ms, replay_ms, refine_ms; } st;

static guint sar_hash(gconstpointer p){
    const ShimSar *s = p;
    register guint64 h = (guint64)(gsize)s->optimal * 0x9E3779B97F4A7C15ULL;
    register gint i;
    for(i = 0; i < 8; i++) h = (h ^ (guint)s->r[i]) * 1099511628211ULL;
    return (guint)(h ^ (h >> 32));
    }
static gboolean sar_equal(gconstpointer a, gconstpointer b){
    const ShimSar *x = a, *y = b;
    return (x->optimal == y->optimal) && !memcmp(x->r, y->r, sizeof(x->r));
    }

static void sar_key(ShimSar *k, Optimal *optimal, Region *r1, Region *r2){
    memset(k, 0, sizeof(*k));
    k->optimal = optimal;
    k->r[0] = r1->query_start; k->r[1] = r1->target_start; k->r[2] = r1->query_length; k->r[3] = r1->target_length;
    if(r2){ k->r[4] = r2->query_start; k->r[5] = r2->target_start; k->r[6] = r2->query_length; k->r[7] = r2->target_length; }
    }

static void bsdp_record(Optimal *optimal, Heuristic_Span *span, Region *r1, Region *r2){
    ShimSar s;
    sar_key(&s, optimal, r1, r2);
    s.span = span;
    s.raw = C4_IMPOSSIBLY_LOW_SCORE;
    g_array_append_val(bsdp_cur->sars, s);
    }

/* a path-only entry (the src traceback of a span: its region is only known once the dst path is) */
static void bsdp_record_path(ShimHPending *hp, Optimal *optimal, Region *region, C4_Score score, gint n_ops,
                             const gint *ops, gint qs, gint ts, gint ql, gint tl){
    ShimSar s;
    sar_key(&s, optimal, region, NULL);
    s.raw = score;
    s.n_ops = n_ops;
    s.ops = g_new(gint, n_ops + 1);
    memcpy(s.ops, ops, sizeof(gint) * n_ops);
    s.path_region[0] = qs; s.path_region[1] = ts; s.path_region[2] = ql; s.path_region[3] = tl;
    s.have_path = TRUE;
    g_array_append_val(hp->sars, s);
    }

static gint *bsdp_alignment_ops(Alignment *a, gint *n_ops){
    register guint o;
    register gint l, total = 0, *ops;
    for(o = 0; o < a->operation_list->len; o++)
        total += ((AlignmentOperation*)a->operation_list->pdata[o])->length;
    ops = g_new(gint, total + 1);
    *n_ops = 0;
    for(o = 0; o < a->operation_list->len; o++){
        register AlignmentOperation *ao = a->operation_list->pdata[o];
        for(l = 0; l < ao->length; l++)
            ops[(*n_ops)++] = ao->transition->id;
        }
    return ops;
    }

/* ---- fronts seen by hpair.o ---------------------------------------------------------------------------------- */

gint BSDP_add_node(BSDP *bsdp, gpointer node_data, C4_Score node_score, gboolean is_valid_start,
                   gboolean is_valid_end, C4_Score start_bound, C4_Score end_bound){
    if(bsdp_mode == BSDP_COLLECT){
        register ShimNodeData *nd = node_data;
        if(nd->sar_start)
            bsdp_record(nd->match->start_terminal->optimal, NULL, nd->sar_start->region, NULL);
        if(nd->sar_end)
            bsdp_record(nd->match->end_terminal->optimal, NULL, nd->sar_end->region, NULL);
        }
    return BSDP_add_node_cpu(bsdp, node_data, node_score, is_valid_start, is_valid_end, start_bound, end_bound);
    }

void BSDP_add_edge(BSDP *bsdp, gpointer edge_data, gint src_node_id, gint dst_node_id, C4_Score bound_score){
    if(bsdp_mode == BSDP_COLLECT){
        register ShimEdgeData *ed = edge_data;
        if(ed->sar_join)
            bsdp_record(ed->sar_join->pair->join->optimal, NULL, ed->sar_join->region, NULL);
        else
            bsdp_record(ed->sar_span->span->dst_optimal, ed->sar_span->span,
                        ed->sar_span->src_region, ed->sar_span->dst_region);
        }
    BSDP_add_edge_cpu(bsdp, edge_data, src_node_id, dst_node_id, bound_score);
    return;
    }

void BSDP_initialise(BSDP *bsdp, C4_Score threshold){
    /* dry run: no node reaches this threshold, so BSDP_next_path returns NULL at once (bsdp.c:562-569,640-642) */
    BSDP_initialise_cpu(bsdp, (bsdp_mode == BSDP_COLLECT) ? C4_IMPOSSIBLY_HIGH_SCORE : threshold);
    return;
    }

BSDP_Path *BSDP_next_path(BSDP *bsdp, C4_Score threshold){
    /* the refinement dry run only wants the first alignment's refinement request (see shim_bsdp_find_path) */
    if((bsdp_mode == BSDP_REFINE_COLLECT) && bsdp_cur && bsdp_cur->refine_seen)
        return NULL;
    return BSDP_next_path_cpu(bsdp, threshold);
    }

/* Viterbi_calculate builds its SubOpt_Index from (subopt, region) and runs the plain DP when that is NULL
 * (viterbi.c:846-865, subopt.c:250-266): exactly then a score computed without blocked cells is the call's result */
static gboolean bsdp_region_unblocked(SubOpt *subopt, Region *region){
    register SubOpt_Index *soi;
    if(!subopt)
        return TRUE;
    soi = SubOpt_Index_create(subopt, region);
    if(!soi)
        return TRUE;
    SubOpt_Index_destroy(soi);
    return FALSE;
    }

static ShimSar *bsdp_lookup(Optimal *optimal, Region *r1, Region *r2){
    ShimSar k;
    if(((bsdp_mode != BSDP_REPLAY) && (bsdp_mode != BSDP_REFINE_COLLECT)) || (!bsdp_cur) || (!bsdp_cur->index))
        return NULL;
    sar_key(&k, optimal, r1, r2);
    return g_hash_table_lookup(bsdp_cur->index, &k);
    }

C4_Score SAR_Terminal_find_score(SAR_Terminal *sar_terminal, Optimal *optimal, HPair *hpair){
    register ShimSar *s = bsdp_lookup(optimal, sar_terminal->region, NULL);
    st.score_calls += (bsdp_mode == BSDP_REPLAY);
    if(s && bsdp_region_unblocked(hpair->subopt, sar_terminal->region)){
        st.score_served += (bsdp_mode == BSDP_REPLAY);
        return s->raw - sar_terminal->component;                                  /* sar.c:393-398 */
        }
    return SAR_Terminal_find_score_cpu(sar_terminal, optimal, hpair);
    }

C4_Score SAR_Join_find_score(SAR_Join *sar_join, HPair *hpair){
    register ShimSar *s = bsdp_lookup(sar_join->pair->join->optimal, sar_join->region, NULL);
    st.score_calls += (bsdp_mode == BSDP_REPLAY);
    if(s && bsdp_region_unblocked(hpair->subopt, sar_join->region)){
        st.score_served += (bsdp_mode == BSDP_REPLAY);
        return s->raw - (sar_join->src_component + sar_join->dst_component);     /* sar.c:697-702 */
        }
    return SAR_Join_find_score_cpu(sar_join, hpair);
    }

C4_Score SAR_Span_find_score(SAR_Span *sar_span, HPair *hpair){
    register ShimSar *s = bsdp_lookup(sar_span->span->dst_optimal, sar_span->src_region, sar_span->dst_region);
    st.score_calls += (bsdp_mode == BSDP_REPLAY);
    if(s && bsdp_region_unblocked(hpair->subopt, sar_span->src_region)
         && bsdp_region_unblocked(hpair->subopt, sar_span->dst_region)){
        st.score_served += (bsdp_mode == BSDP_REPLAY);
        return s->raw - (sar_span->src_component + sar_span->dst_component);     /* sar.c:898-918 */
        }
    return SAR_Span_find_score_cpu(sar_span, hpair);
    }

/* Optimal_find_path's front (c4gpu_shim.c) asks here first: the terminal and join paths of SAR_Alignment_create /
 * SAR_Alignment_add_SAR_Join (sar.c:921-1040) are part of the batch */
Alignment *shim_bsdp_find_path(Optimal *optimal, Region *region, SubOpt *subopt){
    register ShimSar *s;
    register Alignment *alignment;
    register Region *ar;
    register C4_Model *model;
    register gint k, run;
    if((bsdp_mode != BSDP_REPLAY) && (bsdp_mode != BSDP_REFINE_COLLECT))
        return NULL;
    if(bsdp_cur && (optimal == bsdp_cur->gam->optimal)){
        /* GAM_Result_refine_alignment's call (gam.c:618-640).  Dry run: write the first one down and hand back an
         * alignment that loses against the unrefined one (gam.c:664-669), then BSDP_next_path ends the run.  Replay:
         * the first request, with nothing blocked yet, is answered from the refinement batch. */
        if(bsdp_mode == BSDP_REFINE_COLLECT){
            if(!bsdp_cur->refine_seen){
                bsdp_cur->refine_seen = TRUE;
                bsdp_cur->refine_region[0] = region->query_start;  bsdp_cur->refine_region[1] = region->target_start;
                bsdp_cur->refine_region[2] = region->query_length; bsdp_cur->refine_region[3] = region->target_length;
                }
            return Alignment_create(optimal->find_path->model, region, C4_IMPOSSIBLY_LOW_SCORE);
            }
        st.refine_calls++;
        if(bsdp_cur->refine_have && (!bsdp_cur->refine_used)
        && (bsdp_cur->refine_region[0] == region->query_start) && (bsdp_cur->refine_region[1] == region->target_start)
        && (bsdp_cur->refine_region[2] == region->query_length) && (bsdp_cur->refine_region[3] == region->target_length)
        && bsdp_region_unblocked(subopt, region) && bsdp_cur->refined.valid){
            register c4gpu_alignment *a = &bsdp_cur->refined;
            bsdp_cur->refine_used = TRUE;
            st.refine_served++;
            model = optimal->find_path->model;
            ar = Region_create(a->region.query_start, a->region.target_start, a->region.query_length, a->region.target_length);
            alignment = Alignment_create(model, ar, a->score);
            Region_destroy(ar);
            for(k = 0; k < a->n_ops; k++)
                Alignment_add(alignment, model->transition_list->pdata[a->op_transition[k]], a->op_length[k]);
            return alignment;
            }
        return NULL;
        }
    st.path_calls += (bsdp_mode == BSDP_REPLAY);
    s = bsdp_lookup(optimal, region, NULL);
    if((!s) && bsdp_cur && bsdp_cur->index){
        /* the dst path of a span (SAR_Alignment_add_SAR_Span, sar.c:1042-1062): its START cells came from the span's
         * src region, registered with the span just before this call */
        register guint x;
        for(x = 0; x < bsdp_cur->sars->len; x++){
            register ShimSar *c = &g_array_index(bsdp_cur->sars, ShimSar, x);
            if(c->span && (c->span->dst_optimal == optimal)){
                register Region *src = c->span->curr_src_region, *dst = c->span->curr_dst_region;
                if(src && dst && (dst->query_start == region->query_start) && (dst->target_start == region->target_start)
                && (dst->query_length == region->query_length) && (dst->target_length == region->target_length)
                && bsdp_region_unblocked(subopt, src))
                    s = bsdp_lookup(optimal, src, region);
                break;
                }
            }
        }
    if((!s) || (!s->have_path) || (!bsdp_region_unblocked(subopt, region)))
        return NULL;
    st.path_served += (bsdp_mode == BSDP_REPLAY);
    model = optimal->find_path->model;
    ar = Region_create(s->path_region[0], s->path_region[1], s->path_region[2], s->path_region[3]);
    alignment = Alignment_create(model, ar, s->raw);
    Region_destroy(ar);
    for(k = 0; k < s->n_ops; k += run){                       /* Alignment_add merges equal neighbours itself */
        for(run = 1; (k + run < s->n_ops) && (s->ops[k + run] == s->ops[k]); run++);
        Alignment_add(alignment, model->transition_list->pdata[s->ops[k]], run);
        }
    return alignment;
    }

/* ---- the device phase ----------------------------------------------------------------------------------------- */

typedef struct { Optimal *optimal; c4gpu_model fm; gboolean ok; GArray *members; /* (pair index, sar index) */ } ShimGroup;
typedef struct { gint pair, sar; } ShimMember;

static ShimGroup *bsdp_group(GPtrArray *groups, Optimal *optimal, Ungapped_Data *ud){
    register guint i;
    register ShimGroup *g;
    for(i = 0; i < groups->len; i++){
        g = groups->pdata[i];
        if(g->optimal == optimal)
            return g;
        }
    g = g_new0(ShimGroup, 1);
    g->optimal = optimal;
    g->ok = shim_flatten_any(optimal->find_score ? optimal->find_score->model : optimal->find_path->model, ud, &g->fm, TRUE);
    g->members = g_array_new(FALSE, FALSE, sizeof(ShimMember));
    g_ptr_array_add(groups, g);
    return g;
    }

/* step 3 without a device (C4GPU_BSDP_HOST=1, the CPU test of steps 1, 2 and 4): the reference's own DPs */
static void bsdp_host_scores(ShimHPending *hp){
    register guint i;
    gpointer ud = Model_Type_create_data(hp->gam->gas->type, hp->comparison->query, hp->comparison->target);
    register SubOpt *empty = SubOpt_create(hp->comparison->query->len, hp->comparison->target->len);
    HPair fake;
    memset(&fake, 0, sizeof(fake));
    fake.user_data = ud;
    fake.subopt = empty;
    register guint n_cand = hp->sars->len;            /* src traceback entries are appended behind the candidates */
    for(i = 0; i < n_cand; i++){
        register ShimSar *s = &g_array_index(hp->sars, ShimSar, i);
        Region *r1 = Region_create(s->r[0], s->r[1], s->r[2], s->r[3]);      /* the span keeps a share */
        if(s->span){
            SAR_Span tmp;
            Region *r2 = Region_create(s->r[4], s->r[5], s->r[6], s->r[7]);
            tmp.src_region = r1; tmp.dst_region = r2; tmp.src_component = tmp.dst_component = 0; tmp.span = s->span;
            s->raw = SAR_Span_find_score_cpu(&tmp, &fake);
            {   /* the two paths of the span, in SAR_Alignment_add_SAR_Span's order (sar.c:1042-1085) */
                register Heuristic_Data *hd = ud;
                register Alignment *da, *sa;
                hd->heuristic_span = s->span;
                Heuristic_Span_register(s->span, r1, r2);
                Optimal_find_score(s->span->src_optimal, r1, ud, empty);
                Heuristic_Span_integrate(s->span, r1, r2);
                da = Optimal_find_path_cpu(s->span->dst_optimal, r2, ud, C4_IMPOSSIBLY_LOW_SCORE, empty);
                if(da){
                    register gint qe = da->region->query_start - r2->query_start,
                                  te = da->region->target_start - r2->target_start;
                    register Heuristic_Span_Cell *sc = &s->span->dst_integration_matrix[qe][te];
                    s->ops = bsdp_alignment_ops(da, &s->n_ops);
                    s->path_region[0] = da->region->query_start;  s->path_region[1] = da->region->target_start;
                    s->path_region[2] = da->region->query_length; s->path_region[3] = da->region->target_length;
                    s->have_path = (da->score == s->raw);
                    if((sc->query_pos != -1) && (sc->target_pos != -1)){
                        Region *tr = Region_create(r1->query_start, r1->target_start, sc->query_pos - r1->query_start,
                                                   sc->target_pos - r1->target_start);
                        sa = Optimal_find_path_cpu(s->span->src_traceback_optimal, tr, ud, C4_IMPOSSIBLY_LOW_SCORE, empty);
                        if(sa){
                            gint n_ops, *ops = bsdp_alignment_ops(sa, &n_ops);
                            register Optimal *tbo = s->span->src_traceback_optimal;
                            bsdp_record_path(hp, tbo, tr, sa->score, n_ops, ops, sa->region->query_start,
                                             sa->region->target_start, sa->region->query_length, sa->region->target_length);
                            s = &g_array_index(hp->sars, ShimSar, i);        /* the array may have moved */
                            g_free(ops);
                            Alignment_destroy(sa);
                            }
                        Region_destroy(tr);
                        }
                    Alignment_destroy(da);
                    }
                hd->heuristic_span = NULL;
            }
            Region_destroy(r2);
        } else {
            register Alignment *a;
            s->raw = Optimal_find_score(s->optimal, r1, ud, empty);
            /* ... and the path, as the device's path pass would deliver it */
            if((s->optimal->type & Optimal_Type_PATH)
            && (a = Optimal_find_path_cpu(s->optimal, r1, ud, C4_IMPOSSIBLY_LOW_SCORE, empty))){
                s->ops = bsdp_alignment_ops(a, &s->n_ops);
                s->path_region[0] = a->region->query_start;  s->path_region[1] = a->region->target_start;
                s->path_region[2] = a->region->query_length; s->path_region[3] = a->region->target_length;
                s->have_path = (a->score == s->raw);
                Alignment_destroy(a);
                }
            }
        Region_destroy(r1);
        }
    /* what the device route would need of these models: every one flattens and has a compiled family */
    for(i = 0; i < n_cand; i++){
        register ShimSar *s = &g_array_index(hp->sars, ShimSar, i);
        Optimal *opts[4];
        register gint o, no = 0;
        static GHashTable *seen = NULL;
        c4gpu_model fm;
        if(!seen)
            seen = g_hash_table_new(g_direct_hash, g_direct_equal);
        if(s->span){ opts[no++] = s->span->src_optimal; opts[no++] = s->span->dst_optimal; opts[no++] = s->span->src_traceback_optimal; }
        else opts[no++] = s->optimal;
        for(o = 0; o < no; o++){
            register C4_Model *m = opts[o]->find_score ? opts[o]->find_score->model : opts[o]->find_path->model;
            if(g_hash_table_lookup(seen, opts[o]))
                continue;
            g_hash_table_insert(seen, opts[o], opts[o]);
            if((!shim_flatten_any(m, ud, &fm, TRUE)) || (c4gpu_model_device_family(&fm) < 0))
                g_warning("c4gpu bsdp: model [%s] has no device family", m->name);
            else if(shim_env("C4GPU_VERBOSE"))
                g_message("c4gpu bsdp: model [%s] -> device family %d", m->name, c4gpu_model_device_family(&fm));
            }
        }
    SubOpt_destroy(empty);
    Model_Type_destroy_data(hp->gam->gas->type, ud);
    return;
    }

/* what Heuristic_Span_dst_init_start_func (heuristic.c:414-443) returns for every cell of the dst region, as the
 * matrix the device reads its START cells from */
static void bsdp_span_start_cells(Heuristic_Span *span, Region *src, Region *dst, gint cs, c4gpu_score *out){
    register gint i, j, l;
    for(i = 0; i <= dst->query_length; i++)
        for(j = 0; j <= dst->target_length; j++){
            register Heuristic_Span_Cell *sc = &span->dst_integration_matrix[i][j];
            register C4_Score *cell = ((sc->query_pos == -1) || (sc->target_pos == -1)) ? span->dummy_cell
                : span->src_integration_matrix[sc->query_pos - src->query_start][sc->target_pos - src->target_start];
            for(l = 0; l < cs; l++)
                out[((gsize)i * (dst->target_length + 1) + j) * cs + l] = cell[l];
            }
    return;
    }

static gboolean bsdp_device_scores(GPtrArray *todo, c4gpu_batch **batch_out){
    register guint i, k, n = todo->len;
    register ShimHPending *hp = todo->pdata[0];
    register GAM *gam = hp->gam;
    register c4gpu_ctx *ctx = shim_get_ctx();
    register c4gpu_batch *batch = NULL;
    register gpointer ud;
    register GPtrArray *groups = g_ptr_array_new();
    register GHashTable *flat = g_hash_table_new(g_direct_hash, g_direct_equal);
    register gboolean ok = FALSE;
    register gint cs = 1 + gam->heuristic->model->total_shadow_designations;
    c4gpu_model fm;
    c4gpu_params params;
    c4gpu_pair *pair = g_new0(c4gpu_pair, n);
    GPtrArray *strs = g_ptr_array_new();
    if(!ctx)
        goto done;
    for(i = 0; i < n; i++){                              /* one flattened copy per Sequence */
        register gchar *qs, *ts;
        hp = todo->pdata[i];
        if(!(qs = g_hash_table_lookup(flat, hp->comparison->query))){
            qs = Sequence_get_str(hp->comparison->query);
            g_hash_table_insert(flat, hp->comparison->query, qs);
            g_ptr_array_add(strs, qs);
            }
        if(!(ts = g_hash_table_lookup(flat, hp->comparison->target))){
            ts = Sequence_get_str(hp->comparison->target);
            g_hash_table_insert(flat, hp->comparison->target, ts);
            g_ptr_array_add(strs, ts);
            }
        pair[i].query = (const uint8_t*)qs;  pair[i].query_len = hp->comparison->query->len;
        pair[i].target = (const uint8_t*)ts; pair[i].target_len = hp->comparison->target->len;
        }
    hp = todo->pdata[0];
    ud = Model_Type_create_data(gam->gas->type, hp->comparison->query, hp->comparison->target);
    if(shim_flatten_any(gam->heuristic->model, ud, &fm, FALSE)){
        memset(&params, 0, sizeof(params));
        shim_params(ud, &params);
        batch = c4gpu_batch_create(ctx, &fm, &params, pair, n);
        }
    if(!batch){
        Model_Type_destroy_data(gam->gas->type, ud);
        goto done;
        }
    /* group the candidates by DP model; spans appear under their src model (pass 1) and dst model (pass 2) */
    for(i = 0; i < n; i++){
        hp = todo->pdata[i];
        for(k = 0; k < hp->sars->len; k++){
            register ShimSar *s = &g_array_index(hp->sars, ShimSar, k);
            ShimMember m;
            m.pair = i; m.sar = k;
            g_array_append_val(bsdp_group(groups, s->span ? s->span->src_optimal : s->optimal, ud)->members, m);
            }
        }
    ok = TRUE;
    for(i = 0; ok && (i < groups->len); i++){
        register ShimGroup *g = groups->pdata[i];
        register guint cnt = g->members->len;
        register gboolean is_span;
        c4gpu_viterbi_job *job;
        c4gpu_viterbi_result *res;
        c4gpu_score **mat = NULL;
        if(!cnt)
            continue;
        if(!g->ok){ ok = FALSE; break; }
        {
            register ShimMember *m0 = &g_array_index(g->members, ShimMember, 0);
            register ShimHPending *h0 = todo->pdata[m0->pair];
            is_span = (g_array_index(h0->sars, ShimSar, m0->sar).span != NULL);
        }
        job = g_new0(c4gpu_viterbi_job, cnt);
        res = g_new0(c4gpu_viterbi_result, cnt);
        if(is_span)
            mat = g_new0(c4gpu_score*, cnt);
        for(k = 0; k < cnt; k++){
            register ShimMember *m = &g_array_index(g->members, ShimMember, k);
            register ShimHPending *h = todo->pdata[m->pair];
            register ShimSar *s = &g_array_index(h->sars, ShimSar, m->sar);
            job[k].pair = m->pair;
            job[k].region.query_start = s->r[0];  job[k].region.target_start = s->r[1];
            job[k].region.query_length = s->r[2]; job[k].region.target_length = s->r[3];
            if(is_span){                          /* pass 1: src DP, every END cell copied out (cell_end_func) */
                register gsize cells = (gsize)(s->r[2] + 1) * (s->r[3] + 1), x;
                mat[k] = g_new0(c4gpu_score, cells * cs);
                for(x = 0; x < cells; x++)
                    mat[k][x * cs] = C4_IMPOSSIBLY_LOW_SCORE;        /* Heuristic_Span_clear, heuristic.c:556-566 */
                job[k].end_cells = mat[k];
                }
            }
        /* terminals and joins: the path pass gives score, path and aligned region in one launch */
        if(c4gpu_batch_viterbi_model(batch, &g->fm, is_span ? C4GPU_MODE_FIND_SCORE : C4GPU_MODE_FIND_PATH,
                                     job, cnt, res) != 0)
            ok = FALSE;
        if(ok && !is_span){
            for(k = 0; k < cnt; k++){
                register ShimMember *m = &g_array_index(g->members, ShimMember, k);
                register ShimHPending *h = todo->pdata[m->pair];
                register ShimSar *s = &g_array_index(h->sars, ShimSar, m->sar);
                s->raw = res[k].score;
                s->n_ops = res[k].n_ops;
                s->ops = g_new(gint, res[k].n_ops + 1);
                memcpy(s->ops, res[k].ops, sizeof(gint) * res[k].n_ops);
                s->path_region[0] = s->r[0] + res[k].query_start;  s->path_region[1] = s->r[1] + res[k].target_start;
                s->path_region[2] = res[k].query_end - res[k].query_start;
                s->path_region[3] = res[k].target_end - res[k].target_start;
                s->have_path = TRUE;
                }
            }
        if(ok && is_span){
            /* pass 2: integrate on the host with the reference's own function, then the dst DPs read their START
             * cells from the integrated matrices (cell_start_func) */
            register ShimMember *m0 = &g_array_index(g->members, ShimMember, 0);
            register ShimHPending *h0 = todo->pdata[m0->pair];
            register Heuristic_Span *span = g_array_index(h0->sars, ShimSar, m0->sar).span;
            register ShimGroup *gd = bsdp_group(groups, span->dst_optimal, ud);     /* no members of its own */
            c4gpu_viterbi_job *djob = g_new0(c4gpu_viterbi_job, cnt);
            c4gpu_viterbi_result *dres = g_new0(c4gpu_viterbi_result, cnt);
            c4gpu_score **dmat = g_new0(c4gpu_score*, cnt);
            gint **dpos = g_new0(gint*, cnt);          /* per candidate: dst_integration_matrix (query_pos, target_pos) */
            for(k = 0; k < cnt; k++){
                register ShimMember *m = &g_array_index(g->members, ShimMember, k);
                register ShimHPending *h = todo->pdata[m->pair];
                register ShimSar *s = &g_array_index(h->sars, ShimSar, m->sar);
                register gint a, b, l;
                Region *src = Region_create(s->r[0], s->r[1], s->r[2], s->r[3]),
                       *dst = Region_create(s->r[4], s->r[5], s->r[6], s->r[7]);
                Heuristic_Span_register(s->span, src, dst);
                for(a = 0; a <= s->r[2]; a++)
                    for(b = 0; b <= s->r[3]; b++)
                        for(l = 0; l < cs; l++)
                            s->span->src_integration_matrix[a][b][l] = mat[k][((gsize)a * (s->r[3] + 1) + b) * cs + l];
                Heuristic_Span_integrate(s->span, src, dst);
                dmat[k] = g_new(c4gpu_score, (gsize)(s->r[6] + 1) * (s->r[7] + 1) * cs);
                bsdp_span_start_cells(s->span, src, dst, cs, dmat[k]);
                dpos[k] = g_new(gint, (gsize)(s->r[6] + 1) * (s->r[7] + 1) * 2);
                for(a = 0; a <= s->r[6]; a++)
                    for(b = 0; b <= s->r[7]; b++){
                        dpos[k][((gsize)a * (s->r[7] + 1) + b) * 2] = s->span->dst_integration_matrix[a][b].query_pos;
                        dpos[k][((gsize)a * (s->r[7] + 1) + b) * 2 + 1] = s->span->dst_integration_matrix[a][b].target_pos;
                        }
                djob[k].pair = m->pair;
                djob[k].region.query_start = s->r[4];  djob[k].region.target_start = s->r[5];
                djob[k].region.query_length = s->r[6]; djob[k].region.target_length = s->r[7];
                djob[k].start_cells = dmat[k];
                Region_destroy(src);
                Region_destroy(dst);
                }
            /* the path pass: score and dst path (SAR_Alignment_add_SAR_Span, sar.c:1052-1056) in one */
            if((!gd->ok) || (c4gpu_batch_viterbi_model(batch, &gd->fm, C4GPU_MODE_FIND_PATH, djob, cnt, dres) != 0))
                ok = FALSE;
            if(ok){
                /* pass 3: the src traceback of each span (CORNER to CORNER, from the src region's corner to the cell the
                 * dst path started from, sar.c:1057-1075) */
                register ShimGroup *gt = bsdp_group(groups, span->src_traceback_optimal, ud);
                c4gpu_viterbi_job *tjob = g_new0(c4gpu_viterbi_job, cnt);
                c4gpu_viterbi_result *tres = g_new0(c4gpu_viterbi_result, cnt);
                gint *towner = g_new(gint, cnt), nt = 0;
                for(k = 0; k < cnt; k++){
                    register ShimMember *m = &g_array_index(g->members, ShimMember, k);
                    register ShimHPending *h = todo->pdata[m->pair];
                    register ShimSar *s = &g_array_index(h->sars, ShimSar, m->sar);
                    s->raw = dres[k].score;
                    if(dres[k].n_ops > 0){
                        register gsize at = ((gsize)dres[k].query_start * (s->r[7] + 1) + dres[k].target_start) * 2;
                        register gint qp = dpos[k][at], tp = dpos[k][at + 1];
                        s->n_ops = dres[k].n_ops;
                        s->ops = g_new(gint, dres[k].n_ops + 1);
                        memcpy(s->ops, dres[k].ops, sizeof(gint) * dres[k].n_ops);
                        s->path_region[0] = s->r[4] + dres[k].query_start;  s->path_region[1] = s->r[5] + dres[k].target_start;
                        s->path_region[2] = dres[k].query_end - dres[k].query_start;
                        s->path_region[3] = dres[k].target_end - dres[k].target_start;
                        s->have_path = TRUE;
                        if((qp != -1) && (tp != -1) && gt->ok){
                            tjob[nt].pair = m->pair;
                            tjob[nt].region.query_start = s->r[0];       tjob[nt].region.target_start = s->r[1];
                            tjob[nt].region.query_length = qp - s->r[0]; tjob[nt].region.target_length = tp - s->r[1];
                            towner[nt++] = k;
                            }
                        }
                    }
                if(nt && (c4gpu_batch_viterbi_model(batch, &gt->fm, C4GPU_MODE_FIND_PATH, tjob, nt, tres) == 0)){
                    for(k = 0; k < (guint)nt; k++){
                        register ShimMember *m = &g_array_index(g->members, ShimMember, towner[k]);
                        register ShimHPending *h = todo->pdata[m->pair];
                        Region tr;
                        if(tres[k].n_ops <= 0)
                            continue;
                        Region_init_static(&tr, tjob[k].region.query_start, tjob[k].region.target_start,
                                           tjob[k].region.query_length, tjob[k].region.target_length);
                        bsdp_record_path(h, span->src_traceback_optimal, &tr, tres[k].score, tres[k].n_ops, tres[k].ops,
                                         tr.query_start + tres[k].query_start, tr.target_start + tres[k].target_start,
                                         tres[k].query_end - tres[k].query_start, tres[k].target_end - tres[k].target_start);
                        }
                    for(k = 0; k < (guint)nt; k++)
                        c4gpu_viterbi_result_clear(&tres[k]);
                    }
                g_free(tjob); g_free(tres); g_free(towner);
                }
            for(k = 0; k < cnt; k++){
                c4gpu_viterbi_result_clear(&dres[k]);
                g_free(dmat[k]);
                g_free(dpos[k]);
                }
            g_free(djob); g_free(dres); g_free(dmat); g_free(dpos);
            }
        for(k = 0; k < cnt; k++){
            c4gpu_viterbi_result_clear(&res[k]);
            if(mat)
                g_free(mat[k]);
            }
        g_free(job); g_free(res); g_free(mat);
        }
    Model_Type_destroy_data(gam->gas->type, ud);
done:
    if(batch && ok)
        *batch_out = batch;                     /* the refinement batch runs on the same resident pairs */
    else if(batch)
        c4gpu_batch_destroy(batch);
    for(i = 0; i < groups->len; i++){
        register ShimGroup *g = groups->pdata[i];
        g_array_free(g->members, TRUE);
        g_free(g);
        }
    g_ptr_array_free(groups, TRUE);
    for(i = 0; i < strs->len; i++)
        g_free(strs->pdata[i]);
    g_ptr_array_free(strs, TRUE);
    g_hash_table_destroy(flat);
    g_free(pair);
    return ok;
    }

/* ---- the first refinement of every pair (--refine region | full), one batch ------------------------------------------ */

static void bsdp_refine(GPtrArray *todo, c4gpu_batch *batch){
    register guint i, n = todo->len;
    register ShimHPending *hp = todo->pdata[0];
    register GAM *gam = hp->gam;
    register gint dpmemory = gam->optimal->find_path->vas->traceback_memory_limit;
    c4gpu_region *regions = g_new0(c4gpu_region, n);
    uint8_t *active = g_new0(uint8_t, n);
    register guint wanted = 0;
    /* dry runs with the sub-DP scores at hand: each stops at its first Optimal_find_path on the whole model */
    for(i = 0; i < n; i++){
        register GAM_Result *none;
        hp = todo->pdata[i];
        bsdp_mode = BSDP_REFINE_COLLECT;
        bsdp_cur = hp;
        none = GAM_Result_heuristic_create_cpu(hp->gam, hp->comparison);
        bsdp_cur = NULL;
        bsdp_mode = BSDP_OFF;
        if(none)
            GAM_Result_destroy(none);              /* never submitted */
        if(hp->refine_seen){
            regions[i].query_start = hp->refine_region[0];  regions[i].target_start = hp->refine_region[1];
            regions[i].query_length = hp->refine_region[2]; regions[i].target_length = hp->refine_region[3];
            active[i] = 1;
            wanted++;
            }
        }
    if(wanted && batch){
        /* GAM_Result_refine_alignment's threshold is 0 (gam.c:626,645) */
        if(c4gpu_batch_run_regions(batch, regions, active, dpmemory, 0) == 0){
            for(i = 0; i < n; i++){
                hp = todo->pdata[i];
                if(active[i] && (c4gpu_batch_alignment(batch, i, &hp->refined) == 0))
                    hp->refine_have = TRUE;
                }
        } else {
            g_warning("c4gpu: %s -- refinements stay one call at a time", c4gpu_last_error());
            }
    } else if(wanted){                              /* C4GPU_BSDP_HOST: the reference's own Optimal_find_path */
        for(i = 0; i < n; i++){
            register gpointer ud;
            register SubOpt *empty;
            register Alignment *a;
            Region *r;
            hp = todo->pdata[i];
            if(!active[i])
                continue;
            ud = Model_Type_create_data(gam->gas->type, hp->comparison->query, hp->comparison->target);
            empty = SubOpt_create(hp->comparison->query->len, hp->comparison->target->len);
            r = Region_create(regions[i].query_start, regions[i].target_start, regions[i].query_length, regions[i].target_length);
            a = Optimal_find_path_cpu(gam->optimal, r, ud, 0, empty);
            if(a){
                register guint o;
                hp->refined.score = a->score;
                hp->refined.region.query_start = a->region->query_start;   hp->refined.region.target_start = a->region->target_start;
                hp->refined.region.query_length = a->region->query_length; hp->refined.region.target_length = a->region->target_length;
                hp->refined.n_ops = a->operation_list->len;
                hp->refined.op_transition = malloc(sizeof(int32_t) * (a->operation_list->len + 1));
                hp->refined.op_length = malloc(sizeof(int32_t) * (a->operation_list->len + 1));
                for(o = 0; o < a->operation_list->len; o++){
                    register AlignmentOperation *ao = a->operation_list->pdata[o];
                    hp->refined.op_transition[o] = ao->transition->id;
                    hp->refined.op_length[o] = ao->length;
                    }
                hp->refined.valid = 1;
                hp->refine_have = TRUE;
                Alignment_destroy(a);
                }
            Region_destroy(r);
            SubOpt_destroy(empty);
            Model_Type_destroy_data(gam->gas->type, ud);
            }
        }
    g_free(regions);
    g_free(active);
    return;
    }

/* ---- collect / flush ---------------------------------------------------------------------------------------------- */

void shim_bsdp_flush(void){
    register guint i, k;
    register GPtrArray *todo = bsdp_pending;
    register gboolean have_scores;
    c4gpu_batch *batch = NULL;
    gint64 t0 = g_get_monotonic_time(), t1, t2, t3;
    if((!todo) || (!todo->len))
        return;
    bsdp_pending = NULL;
    st.flushes++;
    /* 2. dry runs: the reference builds each HPair, the fronts above write the candidates down */
    bsdp_mode = BSDP_COLLECT;
    for(i = 0; i < todo->len; i++){
        register GAM_Result *none;
        bsdp_cur = todo->pdata[i];
        none = GAM_Result_heuristic_create_cpu(bsdp_cur->gam, bsdp_cur->comparison);
        if(none)                                   /* cannot happen: no node reaches the dry threshold */
            GAM_Result_destroy(none);
        st.candidates += bsdp_cur->sars->len;
        }
    bsdp_cur = NULL;
    bsdp_mode = BSDP_OFF;
    t1 = g_get_monotonic_time();
    /* 3. all candidate DPs of all pairs */
    if(shim_env("C4GPU_BSDP_HOST")){
        for(i = 0; i < todo->len; i++)
            bsdp_host_scores(todo->pdata[i]);
        have_scores = TRUE;
    } else {
        have_scores = bsdp_device_scores(todo, &batch);
        if(!have_scores)
            g_warning("c4gpu: %s -- BSDP sub-DPs stay on the CPU for this batch", c4gpu_last_error());
        }
    t2 = g_get_monotonic_time();
    for(i = 0; have_scores && (i < todo->len); i++){
        register ShimHPending *hp = todo->pdata[i];
        hp->index = g_hash_table_new(sar_hash, sar_equal);
        for(k = 0; k < hp->sars->len; k++){
            register ShimSar *s = &g_array_index(hp->sars, ShimSar, k);
            if(s->span)
                st.spans++;
            g_hash_table_insert(hp->index, s, s);
            }
        }
    /* 3b. --refine: the first refinement of every pair as one more batch on the same resident pairs */
    if(have_scores && (((ShimHPending*)todo->pdata[0])->gam->gas->refinement != GAM_Refinement_NONE)
    && ((ShimHPending*)todo->pdata[0])->gam->optimal && (!shim_env("C4GPU_REFINE_BATCH_OFF")))
        bsdp_refine(todo, batch);
    if(batch)
        c4gpu_batch_destroy(batch);
    t3 = g_get_monotonic_time();
    /* 4. replay in submission order */
    for(i = 0; i < todo->len; i++){
        register ShimHPending *hp = todo->pdata[i];
        register GAM_Result *gam_result;
        bsdp_mode = BSDP_REPLAY;
        bsdp_cur = hp;
        gam_result = GAM_Result_heuristic_create_cpu(hp->gam, hp->comparison);
        bsdp_cur = NULL;
        bsdp_mode = BSDP_OFF;
        if(gam_result){
            GAM_Result_submit(gam_result);
            GAM_Result_destroy(gam_result);
            }
        if(hp->index)
            g_hash_table_destroy(hp->index);
        for(k = 0; k < hp->sars->len; k++)
            g_free(g_array_index(hp->sars, ShimSar, k).ops);
        g_array_free(hp->sars, TRUE);
        if(hp->refine_have)
            c4gpu_alignment_clear(&hp->refined);
        Comparison_destroy(hp->comparison);
        GAM_destroy(hp->gam);
        g_free(hp);
        st.pairs++;
        }
    g_ptr_array_free(todo, TRUE);
    st.dry_ms += (t1 - t0) / 1e3; st.device_ms += (t2 - t1) / 1e3; st.refine_ms += (t3 - t2) / 1e3;
    st.replay_ms += (g_get_monotonic_time() - t3) / 1e3;
    return;
    }

GAM_Result *GAM_Result_heuristic_create(GAM *gam, Comparison *comparison){
    register ShimHPending *hp;
    register gboolean batchable = (shim_batch_size() > 0) && (!gam->gas->use_gapped_extension) && gam->heuristic
        && (bsdp_mode == BSDP_OFF)
        && (!Comparison_Param_get_HSPSet_Argument_Set(comparison->param)->geneseed_threshold)
        && (shim_env("C4GPU_BSDP_HOST") || (shim_ctx_nowait() != NULL))
        && (!shim_env("C4GPU_BSDP_OFF"));
    /* --gappedextension yes: the SDP seam (c4gpu_sdp.c) takes the pairs of the models it covers */
    if(gam->gas->use_gapped_extension && (bsdp_mode == BSDP_OFF) && (!shim_sdp_replaying())
    && shim_sdp_collect(gam, comparison))
        return NULL;
    if(!batchable){
        if((bsdp_mode == BSDP_OFF) && (!shim_sdp_replaying())){
            shim_bsdp_flush();                    /* keep the output order */
            shim_sdp_flush();
            }
        return GAM_Result_heuristic_create_cpu(gam, comparison);
        }
    if(!Comparison_has_hsps(comparison))          /* gam.c:1122 (a comparison whose word hits grew no HSP: c4gpu_hsp.c) */
        return NULL;
    if(bsdp_pending && bsdp_pending->len && (((ShimHPending*)bsdp_pending->pdata[0])->gam != gam))
        shim_bsdp_flush();
    if(!bsdp_pending)
        bsdp_pending = g_ptr_array_new();
    hp = g_new0(ShimHPending, 1);
    hp->gam = GAM_share(gam);
    hp->comparison = Comparison_share(comparison);
    hp->sars = g_array_new(FALSE, TRUE, sizeof(ShimSar));
    g_ptr_array_add(bsdp_pending, hp);
    if((gint)bsdp_pending->len >= shim_batch_size())
        shim_bsdp_flush();
    return NULL;                                  /* submitted by the flush, in submission order */
    }

void shim_bsdp_report(void){
    if(shim_env("C4GPU_VERBOSE") && st.pairs)
        g_message("c4gpu bsdp: %ld pairs in %ld flush(es): %ld candidate sub-DPs (%ld spans) in device batches; "
                  "%ld of %ld score calls and %ld of %ld path calls served from them; %ld of %ld refinements from "
                  "refinement batches; dry runs %.0f ms, device %.0f ms, refinement %.0f ms, replay %.0f ms", st.pairs,
                  st.flushes, st.candidates, st.spans, st.score_served, st.score_calls, st.path_served, st.path_calls,
                  st.refine_served, st.refine_calls, st.dry_ms, st.device_ms, st.refine_ms, st.replay_ms);
    return;
    }
