/* c4_oracle.c — CPU restatement of the reference's C4 Viterbi path.  TEST INFRASTRUCTURE ONLY
 * (see c4_oracle.h).  Plain C, deliberately simple: one interpreted loop over the flattened model, the
 * same evaluation order, tie-breaks, shadow transport and reduced-space orchestration as the reference.
 * Every function cites the reference file:line it restates.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "c4_oracle.h"

#define LOW  C4GPU_IMPOSSIBLY_LOW_SCORE
#define HIGH C4GPU_IMPOSSIBLY_HIGH_SCORE
#define CELL_MAX (1 + C4GPU_MAX_SHADOWS + 3)

static __thread int64_t g_cells = 0;

int64_t oracle_cells_visited(int reset){
    int64_t c = g_cells;
    if(reset) g_cells = 0;
    return c;
    }

/* ---- splice prediction ----------------------------------------------------------------------- */

/* Splice_predict_position, src/sequence/splice.c:320-344 (float accumulation, left to right) and
 * SplicePredictor_round, splice.c:379-381 */
void oracle_splice_predict(const c4gpu_splice_model *sp, const uint8_t *seq, int32_t len, int32_t *pred){
    int32_t pos, i;
    for(pos = 0; pos < len; pos++){
        float score = 0.0f, pos_score;
        int seq_start = pos - sp->splice_after, model_start = 0, calc_length = sp->model_length;
        if(seq_start < 0){
            model_start = -seq_start;
            seq_start = 0;
            calc_length -= model_start;
            }
        if((seq_start + calc_length) > len)
            calc_length = len - seq_start;
        for(i = 0; i < calc_length; i++){
            pos_score = sp->data[model_start+i][sp->index[seq[seq_start+i]]];
            score += pos_score;
            }
        if(sp->gtag_only){                      /* splice.c:312-318,334-336: seq[pos+1] may be the NUL */
            int one = toupper(seq[pos]), two = (pos + 1 < len) ? toupper(seq[pos+1]) : 0;
            if((one != sp->expect_one) || (two != sp->expect_two))
                score = -987654321.0;
            }
        pred[pos] = (int32_t)((score < 0) ? (score - 0.5) : (score + 0.5));
        }
    }

/* ---- SubOpt: blocked points of earlier alignments of the same pair (src/c4/subopt.c) ---------------- */

struct oracle_subopt {
    int32_t query_length, target_length;
    int32_t n, cap;
    int32_t *q, *t;                 /* unique points, insertion order, sequence coordinates */
    int32_t path_count;
};

oracle_subopt *oracle_subopt_create(int32_t query_length, int32_t target_length){   /* subopt.c:24 */
    oracle_subopt *so = calloc(1, sizeof(*so));
    so->query_length = query_length; so->target_length = target_length;
    return so;
    }

void oracle_subopt_destroy(oracle_subopt *so){
    if(!so) return;
    free(so->q); free(so->t); free(so);
    }

static int subopt_check_pos(const oracle_subopt *so, int32_t q, int32_t t){          /* subopt.c:58 */
    int32_t k;
    for(k = 0; k < so->n; k++)
        if((so->q[k] == q) && (so->t[k] == t))
            return 1;
    return 0;
    }

static void subopt_add_point(oracle_subopt *so, int32_t q, int32_t t){
    if(so->n == so->cap){
        so->cap = so->cap ? so->cap * 2 : 256;
        so->q = realloc(so->q, sizeof(int32_t) * so->cap);
        so->t = realloc(so->t, sizeof(int32_t) * so->cap);
        }
    so->q[so->n] = q; so->t[so->n] = t; so->n++;
    }

static int gcd_of(int a, int b){                                                     /* subopt.c:51 */
    while(b){ int r = a % b; a = b; b = r; }
    return a;
    }

/* SubOpt_add_AlignmentOperation, subopt.c:64-128: the cells each step of a match operation LEAVES
 * from (and the sub-steps of a multi-residue match), then the lead-in positions */
static void subopt_add_operation(oracle_subopt *so, int aq, int at, int length, int32_t qpos, int32_t tpos){
    const int g = gcd_of(aq, at), q_move = aq / g, t_move = at / g;
    int32_t q_limit = qpos, t_limit = tpos, qp, tp;
    int i;
    for(i = 0; i < length; i++){
        qp = q_limit; tp = t_limit;
        q_limit += aq; t_limit += at;
        while(qp < q_limit){
            if(!subopt_check_pos(so, qp, tp))
                subopt_add_point(so, qp, tp);
            qp += q_move; tp += t_move;
            }
        }
    qp = qpos - aq + q_move;
    tp = tpos - at + t_move;
    while(qp < qpos){
        if(!subopt_check_pos(so, qp, tp))
            if((qp >= 0) && (tp >= 0))
                subopt_add_point(so, qp, tp);
        qp += q_move; tp += t_move;
        }
    }

/* SubOpt_add_alignment, subopt.c:131-148 */
void oracle_subopt_add_alignment(oracle_subopt *so, const c4gpu_model *model, const c4gpu_alignment *a){
    int32_t qpos = a->region.query_start, tpos = a->region.target_start;
    int k;
    for(k = 0; k < a->n_ops; k++){
        const c4gpu_transition *tr = &model->transitions[a->op_transition[k]];
        if(tr->label == C4GPU_LABEL_MATCH)
            subopt_add_operation(so, tr->advance_query, tr->advance_target, a->op_length[k], qpos, tpos);
        qpos += tr->advance_query * a->op_length[k];
        tpos += tr->advance_target * a->op_length[k];
        }
    so->path_count++;
    }

static int point_cmp(const void *a, const void *b){              /* subopt.c:239-248: target, then query */
    const int32_t *x = a, *y = b;
    if(x[0] != y[0]) return (x[0] < y[0]) ? -1 : 1;
    return (x[1] < y[1]) ? -1 : (x[1] > y[1]);
    }

int32_t oracle_subopt_points(const oracle_subopt *so, int32_t *q, int32_t *t, int32_t max){
    int32_t k, (*p)[2] = malloc(sizeof(int32_t[2]) * (so->n ? so->n : 1));
    for(k = 0; k < so->n; k++){ p[k][0] = so->t[k]; p[k][1] = so->q[k]; }
    qsort(p, so->n, sizeof(int32_t[2]), point_cmp);
    for(k = 0; (k < so->n) && (k < max); k++){ q[k] = p[k][1]; t[k] = p[k][0]; }
    free(p);
    return so->n;
    }

/* SubOpt_Index, subopt.c:250-392: the points inside one region, as rows (one per target position, in
 * region coordinates) of ascending query positions ending in the dummy position query_length+1 */
typedef struct { int32_t target_pos, total, *query_pos; } osoi_row;
typedef struct {
    osoi_row *rows;                 /* the last one is the blank row */
    int32_t n_rows;
    osoi_row *curr_row;
    int32_t curr_row_index, curr_query_index;
} osoi;

static osoi *osoi_create(const oracle_subopt *so, const c4gpu_region *region){
    int32_t k, n = 0, (*p)[2], r;
    osoi *x;
    if(!so) return NULL;
    p = malloc(sizeof(int32_t[2]) * (so->n ? so->n : 1));
    /* RangeTree_find with lengths +1 (subopt.c:258-261, rangetree.c:70-79): both ends inclusive */
    for(k = 0; k < so->n; k++)
        if((so->q[k] >= region->query_start) && (so->q[k] <= region->query_start + region->query_length)
        && (so->t[k] >= region->target_start) && (so->t[k] <= region->target_start + region->target_length)){
            p[n][0] = so->t[k] - region->target_start;
            p[n][1] = so->q[k] - region->query_start;
            n++;
            }
    if(!n){
        free(p);
        return NULL;                /* "Found no points in region" */
        }
    qsort(p, n, sizeof(int32_t[2]), point_cmp);
    x = calloc(1, sizeof(*x));
    x->rows = calloc(n + 1, sizeof(osoi_row));
    for(k = 0; k < n; k++){
        if((!x->n_rows) || (x->rows[x->n_rows-1].target_pos != p[k][0])){
            x->rows[x->n_rows].target_pos = p[k][0];
            x->rows[x->n_rows].query_pos = malloc(sizeof(int32_t) * (n + 1));
            x->n_rows++;
            }
        r = x->n_rows - 1;
        x->rows[r].query_pos[x->rows[r].total++] = p[k][1];
        }
    for(r = 0; r < x->n_rows; r++)
        x->rows[r].query_pos[x->rows[r].total] = so->query_length + 1;
    x->rows[x->n_rows].target_pos = so->target_length + 1;         /* blank row */
    x->rows[x->n_rows].total = 1;
    x->rows[x->n_rows].query_pos = malloc(sizeof(int32_t));
    x->rows[x->n_rows].query_pos[0] = so->query_length + 1;
    x->n_rows++;
    x->curr_row = &x->rows[x->n_rows-1];
    free(p);
    return x;
    }

static void osoi_destroy(osoi *x){
    int32_t r;
    if(!x) return;
    for(r = 0; r < x->n_rows; r++)
        free(x->rows[r].query_pos);
    free(x->rows); free(x);
    }

static void osoi_set_row(osoi *x, int32_t target_pos){                               /* subopt.c:340-375 */
    osoi_row *row;
    if(!x) return;
    row = &x->rows[x->curr_row_index];
    while(row && (row->target_pos < target_pos)){
        if(x->curr_row_index < x->n_rows - 1){
            x->curr_row_index++;
            row = &x->rows[x->curr_row_index];
        } else {
            row = NULL;
            }
        }
    if(row){
        while(row && (row->target_pos > target_pos)){
            if(x->curr_row_index > 0){
                x->curr_row_index--;
                row = &x->rows[x->curr_row_index];
            } else {
                row = NULL;
                }
            }
        x->curr_row = (row && (row->target_pos == target_pos)) ? row : &x->rows[x->n_rows-1];
        }
    x->curr_query_index = 0;
    }

/* SubOpt_Index_is_blocked_fast, subopt.h:77-80: the cursor moves by at most one per call */
static int osoi_is_blocked_fast(osoi *x, int32_t q_pos){
    if(x->curr_row->query_pos[x->curr_query_index] < q_pos)
        return x->curr_row->query_pos[++x->curr_query_index] == q_pos;
    return x->curr_row->query_pos[x->curr_query_index] == q_pos;
    }

/* ---- per-pair data (what `user_data` is in the reference) --------------------------------------- */

typedef struct {
    const c4gpu_model *model;
    const c4gpu_params *params;
    const uint8_t *query, *target;
    int32_t qlen, tlen;
    int32_t *ss[4];                 /* whole-target splice predictions, lazily built */
    int32_t curr_intron_start;      /* Intron_ChainData.curr_intron_start, intron.h */
    /* span models: what the model's cell_start_func returns / its cell_end_func receives (viterbi.c:728-741,
     * 793-799), as matrices over the region: [(i * (T+1)) + j][cell_size] */
    const c4gpu_score *start_cells;
    c4gpu_score *end_cells;
} odata;

static void odata_init(odata *od, const c4gpu_model *model, const c4gpu_params *params,
                       const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen){
    int c;
    memset(od, 0, sizeof(*od));
    od->model = model; od->params = params;
    od->query = query; od->qlen = qlen; od->target = target; od->tlen = tlen;
    /* calc init_funcs (Intron_init_*, intron.c:259-269) create the predictions the calcs read */
    for(c = 0; c < model->n_calcs; c++){
        int kind = model->calcs[c].kind;
        if((kind == C4GPU_CALC_SPLICE_PRE) || (kind == C4GPU_CALC_SPLICE_POST)){
            int k = model->calcs[c].param;
            if((!od->ss[k]) && (tlen > 0)){
                od->ss[k] = malloc(sizeof(int32_t) * tlen);
                oracle_splice_predict(&params->splice[k], target, tlen, od->ss[k]);
                }
            }
        }
    }

static void odata_clear(odata *od){
    int k;
    for(k = 0; k < 4; k++)
        free(od->ss[k]);
    }

/* --annotation: the CDS of the query every following call aligns (Sequence_Annotation, sequence.h:49-54; length <= 0: none).
 * Test infrastructure keeps it beside the calls instead of in their signatures. */
static int32_t oracle_cds_start = 0, oracle_cds_length = 0;
void oracle_set_annotation(int32_t cds_start, int32_t cds_length){
    oracle_cds_start = cds_start;
    oracle_cds_length = cds_length;
    }

/* C4_Calc_score, src/c4/c4.c:1700 + the calc functions of SURVEY.md section 8a-A4 */
static c4gpu_score calc_score(odata *od, int calc, int32_t qpos, int32_t tpos){
    const c4gpu_params *p = od->params;
    const c4gpu_calc *c;
    if(calc < 0)
        return 0;
    c = &od->model->calcs[calc];
    switch(c->kind){
        case C4GPU_CALC_CONST:
            return c->value;
        case C4GPU_CALC_MATCH_DNA:       /* Match_1_1_dna_score_func, match.c:271-285: inside an annotated CDS no 1:1 match */
            if((oracle_cds_length > 0) && (qpos >= oracle_cds_start) && (qpos < oracle_cds_start + oracle_cds_length))
                return LOW;
            return p->dna_submat[p->submat_index[od->query[qpos]]][p->submat_index[od->target[tpos]]];
        case C4GPU_CALC_MATCH_PROTEIN:   /* Match_1_1_protein_score_func, match.c:287 */
            return p->protein_submat[p->submat_index[od->query[qpos]]][p->submat_index[od->target[tpos]]];
        case C4GPU_CALC_MATCH_P2D: {     /* Match_1_3_score_func match.c:347 + Translate_base translate.h:73 */
            uint8_t aa = p->aa[p->trans[ p->nt2d[od->target[tpos]]
                                      | (p->nt2d[od->target[tpos+1]] << 4)
                                      | (p->nt2d[od->target[tpos+2]] << 8)]];
            return p->protein_submat[p->submat_index[od->query[qpos]]][p->submat_index[aa]];
            }
        case C4GPU_CALC_SPLICE_PRE:      /* Intron_calc_*, is_pre, intron.c:148-160 */
            return c->value + od->ss[c->param][tpos];
        case C4GPU_CALC_SPLICE_POST: {   /* Intron_calc_*, post, intron.c:150-160 */
            int32_t intron_length = tpos - od->curr_intron_start + 2;
            if((intron_length < p->min_intron) || (intron_length > p->max_intron))
                return LOW;
            return od->ss[c->param][tpos];
            }
        case C4GPU_CALC_PHASE_POST: {    /* Phase_{1,2}_PROTEIN2DNA_FALSE_TRUE_calc_func, phase.c:188-213 */
            const int phase = c->param, cis = od->curr_intron_start;
            int tp1, tp2, tp3;
            uint8_t aa;
            if(cis < phase)                  /* Phase_calc_is_valid, target chain has the intron */
                return LOW;
            if(phase == 1){ tp1 = cis - 1; tp2 = tpos; tp3 = tpos + 1; }
            else { tp1 = cis - 2; tp2 = cis - 1; tp3 = tpos; }
            aa = p->aa[p->trans[ p->nt2d[od->target[tp1]]
                              | (p->nt2d[od->target[tp2]] << 4)
                              | (p->nt2d[od->target[tp3]] << 8)]];
            return p->protein_submat[p->submat_index[od->query[qpos]]][p->submat_index[aa]];
            }
        default:
            fprintf(stderr, "oracle: calc kind %d not restated\n", c->kind);
            abort();
        }
    return 0;
    }

/* ---- layout ------------------------------------------------------------------------------------ */

/* Layout_model_has_state_active, src/c4/layout.c:21-88 */
static int state_active(const c4gpu_model *m, int state, int start_scope, int end_scope,
                        int qp, int tp, int ql, int tl){
    if((qp < 0) || (tp < 0) || (qp > ql) || (tp > tl))
        return 0;
    if(state == m->start_state){
        switch(start_scope){
            case C4GPU_SCOPE_ANYWHERE: break;
            case C4GPU_SCOPE_EDGE:   if((qp != 0) && (tp != 0)) return 0; break;
            case C4GPU_SCOPE_QUERY:  if(qp != 0) return 0; break;
            case C4GPU_SCOPE_TARGET: if(tp != 0) return 0; break;
            case C4GPU_SCOPE_CORNER: if((qp != 0) || (tp != 0)) return 0; break;
            }
        }
    if(state == m->end_state){
        switch(end_scope){
            case C4GPU_SCOPE_ANYWHERE: break;
            case C4GPU_SCOPE_EDGE:   if((qp != ql) && (tp != tl)) return 0; break;
            case C4GPU_SCOPE_QUERY:  if(qp != ql) return 0; break;
            case C4GPU_SCOPE_TARGET: if(tp != tl) return 0; break;
            case C4GPU_SCOPE_CORNER: if((qp != ql) || (tp != tl)) return 0; break;
            }
        }
    return 1;
    }

/* Layout_transition_is_valid, src/c4/layout.c:122-154 (what Layout_is_transition_valid caches) */
static int transition_valid(const c4gpu_model *m, int start_scope, int end_scope,
                            const c4gpu_transition *t, int i, int j, int ql, int tl){
    if(!state_active(m, t->input, start_scope, end_scope,
                     i - t->advance_query, j - t->advance_target, ql, tl))
        return 0;
    if(!state_active(m, t->output, start_scope, end_scope, i, j, ql, tl))
        return 0;
    return 1;
    }

/* ---- memory-size decisions --------------------------------------------------------------------- */

/* Matrix3d_size / Matrix4d_size, src/struct/matrix.c:74-100,137-171 (including the padding quirk) */
static size_t matrix3d_size(int a, int b, int c, size_t cell){
    unsigned long primary = a * sizeof(void*), secondary = b * sizeof(void*), row = c * cell,
                  block = secondary + (b * row), total;
    double cprimary = a * sizeof(void*), csecondary = b * sizeof(void*), crow = c * cell,
           cblock = csecondary + (b * crow), ctotal;
    block += (block % sizeof(void*));
    cblock += (block % sizeof(void*));
    total = primary + (a * block);
    ctotal = cprimary + (a * cblock);
    if((ctotal - total) > 1)
        return 0;
    return total;
    }

static size_t matrix4d_size(int a, int b, int c, int d, size_t cell){
    unsigned long primary = a * sizeof(void*), secondary = b * sizeof(void*),
                  tertiary = c * sizeof(void*), row = d * cell,
                  block = tertiary + (c * row), sheet, total;
    double cprimary = a * sizeof(void*), csecondary = b * sizeof(void*),
           ctertiary = c * sizeof(void*), crow = d * cell,
           cblock = ctertiary + (c * crow), csheet, ctotal;
    block += (block % sizeof(void*));
    cblock += (block % sizeof(void*));
    sheet = secondary + (b * block);
    csheet = csecondary + (b * cblock);
    sheet += (sheet % sizeof(void*));
    csheet += (sheet % sizeof(void*));
    total = primary + (a * sheet);
    ctotal = cprimary + (a * csheet);
    if((ctotal - total) > 1)
        return 0;
    return total;
    }

#define SIZEOF_VITERBI_ROW 24   /* sizeof(Viterbi_Row) on LP64, viterbi.h:41-47 */

/* Viterbi_get_row_size / Viterbi_Row_get_size, viterbi.c:108-118,195-205: note the gint truncation */
static size_t row_size(const c4gpu_model *m, const c4gpu_region *r, int cell_size){
    int mat_size = (int)matrix4d_size(m->max_target_advance+1, r->query_length+1, m->n_states,
                                      cell_size, sizeof(c4gpu_score));
    if(!mat_size)
        return 0;
    return SIZEOF_VITERBI_ROW + mat_size;
    }

/* Viterbi_use_reduced_space, viterbi.c:128-150; the Viterbi consulted is always optimal->find_path */
int oracle_use_reduced_space(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb){
    size_t row_memory = row_size(m, r, 1 + m->total_shadow_designations);
    size_t traceback_memory = matrix3d_size(r->query_length+1, r->target_length+1, m->n_states,
                                            sizeof(void*));
    size_t memory_limit = (size_t)(dpmemory_mb << 20);
    if(r->query_length <= (m->max_query_advance * 6))
        return 0;
    if(r->target_length <= (m->max_target_advance * 6))
        return 0;
    if((!row_memory) || (!traceback_memory))
        return 1;
    if((row_memory + traceback_memory) > memory_limit)
        return 1;
    return 0;
    }

/* Viterbi_checkpoint_rows, viterbi.c:207-218; the Viterbi is find_checkpoint_continuation whose
 * cell_size has the extra checkpoint slot (viterbi.c:53-54) */
int oracle_checkpoint_rows(const c4gpu_model *m, const c4gpu_region *r, int dpmemory_mb){
    size_t row_memory = row_size(m, r, 1 + m->total_shadow_designations + 1);
    int avail_rows = (int)((((size_t)(dpmemory_mb << 20)) / row_memory) - 1);
    int max_rows = (r->target_length / (m->max_target_advance << 1)) - 2;
    if(avail_rows < 1)
        return 1;
    return (avail_rows < max_rows) ? avail_rows : max_rows;
    }

/* ---- the Viterbi recurrence ----------------------------------------------------------------- */

typedef struct {
    const c4gpu_model *m;
    int mode, use_continuation;
    int start_scope, end_scope;
    int cell_size, rsq_id, rst_id, cp_id;
} oviterbi;

/* Viterbi_create + Viterbi_get_cell_size + Viterbi_Row_create, viterbi.c:42-96,154-185 */
static void oviterbi_init(oviterbi *v, const c4gpu_model *m, int mode, int use_continuation){
    v->m = m; v->mode = mode; v->use_continuation = use_continuation;
    v->start_scope = use_continuation ? C4GPU_SCOPE_CORNER : m->start_scope;
    v->end_scope = use_continuation ? C4GPU_SCOPE_CORNER : m->end_scope;
    v->cell_size = 1 + m->total_shadow_designations;
    v->rsq_id = v->rst_id = v->cp_id = -1;
    if(mode == C4GPU_MODE_FIND_REGION){
        if(v->start_scope != C4GPU_SCOPE_CORNER){
            if(v->start_scope != C4GPU_SCOPE_QUERY)
                v->rsq_id = v->cell_size++;
            if(v->start_scope != C4GPU_SCOPE_TARGET)
                v->rst_id = v->cell_size++;
            }
        }
    if(mode == C4GPU_MODE_FIND_CHECKPOINTS)
        v->cp_id = v->cell_size++;
    }

/* Viterbi_Checkpoint_SRP_encode / decode, viterbi.c:515-535 */
static int srp_encode(const c4gpu_model *m, int state, int row, int qpos){
    return (((qpos * m->n_states) + state) * m->max_target_advance) + row;
    }
static void srp_decode(const c4gpu_model *m, int srp, int *state, int *row, int *pos){
    int rem;
    *row = srp % m->max_target_advance;
    rem = srp / m->max_target_advance;
    *state = rem % m->n_states;
    *pos = rem / m->n_states;
    }

typedef struct {
    c4gpu_score *rows;           /* [(max_at+1)][Q+1][S][cell] */
    c4gpu_score **prev_row;      /* rotating row pointers */
    uint8_t *traceback;          /* [(Q+1)][(T+1)][S] transition id + 1 */
    int32_t *checkpoints;        /* [cp][max_at][Q+1][S][cell] */
    int cp_count, section_length, counter, last_srp;
    int curr_query_end, curr_target_end, curr_query_start, curr_target_start;
    c4gpu_score final_cell[CELL_MAX];
} ovdata;

#define CELL(v, rowp, i, s) ((rowp) + ((size_t)(i) * (v)->m->n_states + (s)) * (v)->cell_size)

/* Viterbi_interpreted, src/c4/viterbi.c:655-837 */
static c4gpu_score viterbi_run(const oviterbi *v, const c4gpu_region *region, ovdata *vd, odata *od,
                               const c4gpu_continuation *cont, osoi *soi){
    const c4gpu_model *m = v->m;
    const int Q = region->query_length, T = region->target_length, S = m->n_states,
              cs = v->cell_size, mta = m->max_target_advance;
    c4gpu_score t, score = LOW, *src, *dst, *swap, dummy_start[CELL_MAX];
    int i, j, k, l, end_is_set = 0, final_state, state_is_set[C4GPU_MAX_STATES];
    final_state = cont ? cont->final_state : m->end_state;
    for(j = 0; j <= T; j++){
        osoi_set_row(soi, j);                                       /* viterbi.c:689 */
        for(i = 0; i <= Q; i++){
            for(k = 0; k < S; k++){
                state_is_set[k] = 0;
                CELL(v, vd->prev_row[0], i, k)[0] = LOW;
                }
            for(k = 0; k < m->n_transitions; k++){
                const c4gpu_transition *tr = &m->transitions[k];
                if(!transition_valid(m, v->start_scope, v->end_scope, tr, i, j, Q, T))
                    continue;
                if((tr->label == C4GPU_LABEL_MATCH) && soi && osoi_is_blocked_fast(soi, i))
                    continue;                                       /* viterbi.c:701-704 */
                if(cont && (tr->input == m->start_state)){          /* viterbi.c:705-714 */
                    dst = CELL(v, vd->prev_row[0], 0, cont->first_state);
                    for(l = 0; l < cs; l++)
                        dst[l] = cont->first_cell[l];
                    state_is_set[cont->first_state] = 1;
                    }
                src = CELL(v, vd->prev_row[tr->advance_target], i - tr->advance_query, tr->input);
                dst = CELL(v, vd->prev_row[0], i, tr->output);
                t = 0;
                if(tr->input == m->start_state){
                    if(cont){
                        src = CELL(v, vd->prev_row[0], 0, m->start_state);
                        t = src[0];
                        }
                    else if(od->start_cells){                       /* cell_start_func, viterbi.c:728-741 */
                        const c4gpu_score *sc = od->start_cells
                            + ((size_t)(i - tr->advance_query) * (T+1) + (j - tr->advance_target)) * cs;
                        for(l = 0; l < cs; l++)
                            dummy_start[l] = sc[l];
                        src = dummy_start;
                        t = src[0];
                        }
                } else {
                    t = src[0];
                    }
                /* Viterbi_Row_shadow_end, viterbi.c:426-443 -> Intron_*_end_func, intron.c:468-476 */
                for(l = 0; l < m->n_shadows; l++)
                    if(tr->dst_shadow_mask & (1u << l))
                        od->curr_intron_start = src[m->shadows[l].designation + 1];
                t += calc_score(od, tr->calc, region->query_start + i - tr->advance_query,
                                              region->target_start + j - tr->advance_target);
                if(tr->calc >= 0){
                    if((m->calcs[tr->calc].protect & C4GPU_PROTECT_UNDERFLOW) && (t < LOW))
                        t = LOW;
                    if((m->calcs[tr->calc].protect & C4GPU_PROTECT_OVERFLOW) && (t > HIGH))
                        t = HIGH;
                    }
                if(state_is_set[tr->output]){
                    if(!(dst[0] < t))
                        continue;
                } else {
                    state_is_set[tr->output] = 1;
                    }
                /* Viterbi_Data_assign, viterbi.c:445-462 */
                dst[0] = t;
                if(tr->input == m->start_state){                     /* viterbi.c:403-412 */
                    if(v->rsq_id != -1)
                        src[v->rsq_id] = i - tr->advance_query;
                    if(v->rst_id != -1)
                        src[v->rst_id] = j - tr->advance_target;
                    }
                for(l = 0; l < m->n_shadows; l++)                    /* viterbi.c:414-422, intron.c:454 */
                    if(m->shadows[l].src_state_mask & (1u << tr->input))
                        src[m->shadows[l].designation + 1] = m->shadows[l].on_target
                            ? (region->target_start + j - tr->advance_target)
                            : (region->query_start + i - tr->advance_query);
                for(l = 1; l < cs; l++)
                    dst[l] = src[l];
                if(vd->traceback)
                    vd->traceback[((size_t)i * (T+1) + j) * S + tr->output] = (uint8_t)(k + 1);
                }
            if(state_is_set[m->end_state]){                          /* viterbi.c:778-799 */
                c4gpu_score *cell = CELL(v, vd->prev_row[0], i, final_state);
                t = cell[0];
                if((!end_is_set) || (score < t)){
                    score = t;
                    end_is_set = 1;
                    vd->curr_query_end = i;                           /* Viterbi_Data_register_end :464 */
                    vd->curr_target_end = j;
                    if(v->rsq_id != -1)
                        vd->curr_query_start = cell[v->rsq_id];
                    if(v->rst_id != -1)
                        vd->curr_target_start = cell[v->rst_id];
                    }
                if(od->end_cells){                                   /* cell_end_func, viterbi.c:793-799 */
                    c4gpu_score *ec = od->end_cells + ((size_t)i * (T+1) + j) * cs;
                    cell = CELL(v, vd->prev_row[0], i, m->end_state);
                    for(l = 0; l < cs; l++)
                        ec[l] = cell[l];
                    }
                }
            }
        g_cells += (int64_t)(Q + 1);
        /* Viterbi_Checkpoint_process, viterbi.c:605-631 */
        if((v->mode == C4GPU_MODE_FIND_CHECKPOINTS) && j && (!(j % vd->section_length))
        && (vd->counter < vd->cp_count)){
            int32_t *cp = vd->checkpoints + (size_t)vd->counter++ * mta * (Q+1) * S * cs;
            int r;
            for(r = 0; r < mta; r++)
                for(i = 0; i <= Q; i++)
                    for(k = 0; k < S; k++){
                        c4gpu_score *cell = CELL(v, vd->prev_row[r], i, k);
                        for(l = 0; l < cs; l++)
                            cp[(((size_t)r * (Q+1) + i) * S + k) * cs + l] = cell[l];
                        cell[cs-1] = srp_encode(m, k, r, i);
                        }
            }
        swap = vd->prev_row[mta];                                   /* rotate rows backwards */
        for(i = mta; i > 0; i--)
            vd->prev_row[i] = vd->prev_row[i-1];
        vd->prev_row[0] = swap;
        }
    if(v->mode == C4GPU_MODE_FIND_CHECKPOINTS)
        vd->last_srp = CELL(v, vd->prev_row[1], Q, final_state)[cs-1];
    src = CELL(v, vd->prev_row[1], Q, final_state);                  /* viterbi.c:828-832 */
    for(l = 0; l < cs; l++)
        vd->final_cell[l] = src[l];
    return score;
    }

static void ovdata_init(ovdata *vd, const oviterbi *v, const c4gpu_region *region, int cp_count){
    const c4gpu_model *m = v->m;
    const int Q = region->query_length, T = region->target_length, S = m->n_states;
    size_t row_ints = (size_t)(Q+1) * S * v->cell_size;
    int r;
    size_t x;
    memset(vd, 0, sizeof(*vd));
    vd->rows = calloc((size_t)(m->max_target_advance+1) * row_ints, sizeof(c4gpu_score));
    vd->prev_row = malloc(sizeof(c4gpu_score*) * (m->max_target_advance+1));
    for(r = 0; r <= m->max_target_advance; r++)
        vd->prev_row[r] = vd->rows + r * row_ints;
    for(x = 0; x < (size_t)(m->max_target_advance+1) * (Q+1) * S; x++)     /* viterbi.c:180-183 */
        vd->rows[x * v->cell_size] = LOW;
    if(v->mode == C4GPU_MODE_FIND_PATH)
        vd->traceback = calloc((size_t)(Q+1) * (T+1) * S, 1);
    if(v->mode == C4GPU_MODE_FIND_CHECKPOINTS){                      /* Viterbi_Checkpoint_create :231 */
        vd->cp_count = cp_count;
        vd->checkpoints = calloc((size_t)cp_count * m->max_target_advance * row_ints, sizeof(int32_t));
        vd->section_length = T / (cp_count + 1);
        }
    }

static void ovdata_clear(ovdata *vd){
    free(vd->rows);
    free(vd->prev_row);
    free(vd->traceback);
    free(vd->checkpoints);
    }

/* Viterbi_Data_create_Alignment's walk, viterbi.c:342-379: returns the raw transition path
 * (start -> end order) and the start offsets (i, j) */
static int traceback_walk(const oviterbi *v, const ovdata *vd, const c4gpu_region *region,
                          const c4gpu_continuation *cont, int32_t **ops_out, int *qs, int *ts){
    const c4gpu_model *m = v->m;
    const int T = region->target_length, S = m->n_states;
    int i = vd->curr_query_end, j = vd->curr_target_end, n = 0, cap = 64, a, z;
    int32_t *ops = malloc(sizeof(int32_t) * cap);
    int tr = vd->traceback[((size_t)i * (T+1) + j) * S + (cont ? cont->final_state : m->end_state)] - 1;
    do {
        if(n == cap) ops = realloc(ops, sizeof(int32_t) * (cap *= 2));
        ops[n++] = tr;
        i -= m->transitions[tr].advance_query;
        j -= m->transitions[tr].advance_target;
        tr = vd->traceback[((size_t)i * (T+1) + j) * S + m->transitions[tr].input] - 1;
        if(tr < 0)
            break;
        if(m->transitions[tr].input == m->start_state){
            if(n == cap) ops = realloc(ops, sizeof(int32_t) * (cap *= 2));
            ops[n++] = tr;
            i -= m->transitions[tr].advance_query;
            j -= m->transitions[tr].advance_target;
            break;
            }
        if(cont && (!(i|j)) && (m->transitions[tr].output == cont->first_state))
            break;
    } while(1);
    for(a = 0, z = n-1; a < z; a++, z--){
        int32_t s = ops[a]; ops[a] = ops[z]; ops[z] = s;
        }
    *ops_out = ops; *qs = i; *ts = j;
    return n;
    }

int oracle_viterbi(const c4gpu_model *model, const c4gpu_params *params, int mode,
                   const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                   const c4gpu_region *region, const c4gpu_continuation *continuation,
                   int checkpoint_count, oracle_viterbi_out *out){
    return oracle_viterbi_subopt(model, params, mode, query, qlen, target, tlen, region, continuation,
                                 checkpoint_count, NULL, out);
    }

/* a span DP: FIND_SCORE / FIND_PATH with the cell_start_func / cell_end_func seam (either may be NULL) */
static const c4gpu_score *g_span_start = NULL;
static c4gpu_score *g_span_end = NULL;
int oracle_viterbi_span(const c4gpu_model *model, const c4gpu_params *params, int mode,
                   const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                   const c4gpu_region *region, const c4gpu_score *start_cells, c4gpu_score *end_cells,
                   oracle_viterbi_out *out){
    int rc;
    g_span_start = start_cells; g_span_end = end_cells;
    rc = oracle_viterbi_subopt(model, params, mode, query, qlen, target, tlen, region, NULL, 0, NULL, out);
    g_span_start = NULL; g_span_end = NULL;
    return rc;
    }

/* Viterbi_calculate, viterbi.c:846-865: the index is built per call from the pair's SubOpt */
int oracle_viterbi_subopt(const c4gpu_model *model, const c4gpu_params *params, int mode,
                   const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                   const c4gpu_region *region, const c4gpu_continuation *continuation,
                   int checkpoint_count, const oracle_subopt *subopt, oracle_viterbi_out *out){
    oviterbi v;
    ovdata vd;
    odata od;
    osoi *soi = osoi_create(subopt, region);
    int l;
    memset(out, 0, sizeof(*out));
    odata_init(&od, model, params, query, qlen, target, tlen);
    od.start_cells = g_span_start; od.end_cells = g_span_end;
    oviterbi_init(&v, model, mode, continuation ? 1 : 0);
    ovdata_init(&vd, &v, region, checkpoint_count);
    out->score = viterbi_run(&v, region, &vd, &od, continuation, soi);
    osoi_destroy(soi);
    out->cell_size = v.cell_size;
    out->query_start = vd.curr_query_start; out->target_start = vd.curr_target_start;
    out->query_end = vd.curr_query_end; out->target_end = vd.curr_target_end;
    for(l = 0; l < v.cell_size; l++)
        out->final_cell[l] = vd.final_cell[l];
    out->last_srp = vd.last_srp;
    if(mode == C4GPU_MODE_FIND_PATH){
        int qs, ts;
        out->n_ops = traceback_walk(&v, &vd, region, continuation, &out->ops, &qs, &ts);
        out->query_start = qs; out->target_start = ts;
        }
    if(mode == C4GPU_MODE_FIND_CHECKPOINTS){
        out->checkpoints = vd.checkpoints;
        vd.checkpoints = NULL;
        }
    ovdata_clear(&vd);
    odata_clear(&od);
    return 0;
    }

void oracle_viterbi_out_clear(oracle_viterbi_out *out){
    free(out->ops);
    free(out->checkpoints);
    out->ops = NULL; out->checkpoints = NULL;
    }

/* ---- alignments ---------------------------------------------------------------------------------- */

/* Alignment_add, src/c4/alignment.c:75-102 */
static void alignment_add(c4gpu_alignment *a, int *cap, int transition, int length){
    if(a->n_ops && (a->op_transition[a->n_ops-1] == transition)){
        a->op_length[a->n_ops-1] += length;
        if(a->op_length[a->n_ops-1] == 0)
            a->n_ops--;
        return;
        }
    if(a->n_ops == *cap){
        *cap = (*cap) ? (*cap * 2) : 32;
        a->op_transition = realloc(a->op_transition, sizeof(int32_t) * (*cap));
        a->op_length = realloc(a->op_length, sizeof(int32_t) * (*cap));
        }
    a->op_transition[a->n_ops] = transition;
    a->op_length[a->n_ops++] = length;
    }

void oracle_alignment_clear(c4gpu_alignment *a){
    free(a->op_transition);
    free(a->op_length);
    memset(a, 0, sizeof(*a));
    }

/* ---- Optimal ------------------------------------------------------------------------------------- */

c4gpu_score oracle_find_score(const c4gpu_model *model, const c4gpu_params *params,
                              const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen){
    oracle_viterbi_out out;
    c4gpu_region region = {0, 0, qlen, tlen};
    c4gpu_score score;
    oracle_viterbi(model, params, C4GPU_MODE_FIND_SCORE, query, qlen, target, tlen, &region, NULL, 0, &out);
    score = out.score;
    oracle_viterbi_out_clear(&out);
    return score;
    }

/* Viterbi_SubAlignment, viterbi.h / viterbi.c:482-496 */
typedef struct {
    c4gpu_region region;
    int first_state;
    c4gpu_score final_cell[CELL_MAX];
} ovsa;

typedef struct { ovsa *v; int n, cap; } ovsa_list;

static void ovsa_push(ovsa_list *l, const ovsa *s){
    if(l->n == l->cap){
        l->cap = l->cap ? l->cap * 2 : 16;
        l->v = realloc(l->v, sizeof(ovsa) * l->cap);
        }
    l->v[l->n++] = *s;
    }

typedef struct {
    const c4gpu_model *model;
    const c4gpu_params *params;
    const uint8_t *query, *target;
    int32_t qlen, tlen;
    int dpmemory_mb;
    const oracle_subopt *subopt;
} octx;

/* Viterbi_Checkpoint_traceback, src/c4/viterbi.c:537-601 */
static void checkpoint_traceback(const octx *cx, const oracle_viterbi_out *vo, int cp_count,
                                 const c4gpu_region *region, int first_state, ovsa_list *out){
    const c4gpu_model *m = cx->model;
    const int Q = region->query_length, S = m->n_states, cs = vo->cell_size, mta = m->max_target_advance;
    const int section_length = region->target_length / (cp_count + 1);
    int state, row, pos, prev_row, i, l, query_start, target_start;
    const int32_t *cell;
    ovsa vsa, prev;
#define CP(c, r, q, s) (vo->checkpoints + ((((size_t)(c) * mta + (r)) * (Q+1) + (q)) * S + (s)) * cs)
    srp_decode(m, vo->last_srp, &state, &row, &pos);
    query_start = region->query_start + pos;
    target_start = region->target_start + (section_length * cp_count) - row;
    vsa.region.query_start = query_start;
    vsa.region.target_start = target_start;
    vsa.region.query_length = (region->query_start + region->query_length) - query_start;
    vsa.region.target_length = (region->target_start + region->target_length) - target_start;
    vsa.first_state = state;
    for(l = 0; l < cs; l++) vsa.final_cell[l] = vo->final_cell[l];
    ovsa_push(out, &vsa);
    for(i = cp_count-1; i >= 1; i--){
        prev = vsa;
        prev_row = row;
        cell = CP(i, prev_row, prev.region.query_start - region->query_start, prev.first_state);
        srp_decode(m, cell[cs-1], &state, &row, &pos);
        query_start = region->query_start + pos;
        target_start = prev.region.target_start - section_length - row + prev_row;
        vsa.region.query_start = query_start;
        vsa.region.target_start = target_start;
        vsa.region.query_length = prev.region.query_start - query_start;
        vsa.region.target_length = prev.region.target_start - target_start;
        vsa.first_state = state;
        for(l = 0; l < cs; l++) vsa.final_cell[l] = cell[l];
        ovsa_push(out, &vsa);
        }
    prev = vsa;
    cell = CP(0, row, prev.region.query_start - region->query_start, prev.first_state);
    vsa.region.query_start = region->query_start;
    vsa.region.target_start = region->target_start;
    vsa.region.query_length = query_start - region->query_start;
    vsa.region.target_length = target_start - region->target_start;
    vsa.first_state = first_state;
    for(l = 0; l < cs; l++) vsa.final_cell[l] = cell[l];
    ovsa_push(out, &vsa);
#undef CP
    }

/* Optimal_find_checkpoints_recur, src/c4/optimal.c:160-230.  Appends to vsa_list in path order. */
static c4gpu_score find_checkpoints_recur(const octx *cx, const c4gpu_region *region, ovsa_list *vsa_list,
                                          int first_state, const c4gpu_score *first_cell,
                                          int final_state){
    oracle_viterbi_out vo;
    c4gpu_continuation cont;
    ovsa_list sub = {NULL, 0, 0};
    int cp_count = oracle_checkpoint_rows(cx->model, region, cx->dpmemory_mb), i, l;
    c4gpu_score score;
    cont.first_state = first_state; cont.final_state = final_state;
    for(l = 0; l < CELL_MAX; l++) cont.first_cell[l] = first_cell[l];
    oracle_viterbi_subopt(cx->model, cx->params, C4GPU_MODE_FIND_CHECKPOINTS, cx->query, cx->qlen,
                   cx->target, cx->tlen, region, &cont, cp_count, cx->subopt, &vo);
    score = vo.score;
    checkpoint_traceback(cx, &vo, cp_count, region, first_state, &sub);
    oracle_viterbi_out_clear(&vo);
    for(i = sub.n-1; i >= 0; i--){
        ovsa *vsa = &sub.v[i];
        if(oracle_use_reduced_space(cx->model, &vsa->region, cx->dpmemory_mb)){
            const c4gpu_score *sub_first_cell = (i < sub.n-1) ? sub.v[i+1].final_cell : first_cell;
            int sub_final_state = i ? sub.v[i-1].first_state : final_state;
            find_checkpoints_recur(cx, &vsa->region, vsa_list, vsa->first_state, sub_first_cell,
                                   sub_final_state);
        } else {
            ovsa_push(vsa_list, vsa);
            }
        }
    free(sub.v);
    return score;
    }
/* note on (i < sub.n-1): in the reference `prev_vsa` is the previously *visited* element of the
 * reversed walk, i.e. sub_vsa_list->pdata[i+1] (optimal.c:204-206,225). */

static void path_to_alignment(c4gpu_alignment *a, int *cap, const int32_t *ops, int n){
    int i;
    for(i = 0; i < n; i++)
        alignment_add(a, cap, ops[i], 1);
    }

/* Optimal_find_path_reduced_space + Optimal_compute_subalignments, optimal.c:266-345 */
static void find_path_reduced_space(const octx *cx, const c4gpu_region *region, c4gpu_alignment *out){
    ovsa_list vsa_list = {NULL, 0, 0};
    c4gpu_score zero_cell[CELL_MAX] = {0};
    int cap = 0, n;
    c4gpu_score score = find_checkpoints_recur(cx, region, &vsa_list, cx->model->start_state, zero_cell,
                                               cx->model->end_state);
    memset(out, 0, sizeof(*out));
    out->score = score;
    out->region = *region;
    out->valid = 1;
    for(n = 0; n < vsa_list.n; n++){
        ovsa *vsa = &vsa_list.v[n];
        oracle_viterbi_out vo;
        c4gpu_continuation cont;
        int l;
        const c4gpu_score *first_cell = n ? vsa_list.v[n-1].final_cell : zero_cell;
        cont.first_state = vsa->first_state;
        cont.final_state = (n+1 < vsa_list.n) ? vsa_list.v[n+1].first_state : cx->model->end_state;
        for(l = 0; l < CELL_MAX; l++) cont.first_cell[l] = first_cell[l];
        /* Optimal_find_path_quadratic_space_continuation, optimal.c:232-264 */
        oracle_viterbi_subopt(cx->model, cx->params, C4GPU_MODE_FIND_PATH, cx->query, cx->qlen,
                       cx->target, cx->tlen, &vsa->region, &cont, 0, cx->subopt, &vo);
        path_to_alignment(out, &cap, vo.ops, vo.n_ops);
        oracle_viterbi_out_clear(&vo);
        }
    free(vsa_list.v);
    }

/* Optimal_find_path_quadratic_space, optimal.c:349-364 + Viterbi_Data_create_Alignment viterbi.c:380-391 */
static void find_path_quadratic_space(const octx *cx, const c4gpu_region *region, c4gpu_alignment *out){
    oracle_viterbi_out vo;
    int cap = 0;
    oracle_viterbi_subopt(cx->model, cx->params, C4GPU_MODE_FIND_PATH, cx->query, cx->qlen, cx->target, cx->tlen,
                   region, NULL, 0, cx->subopt, &vo);
    memset(out, 0, sizeof(*out));
    out->score = vo.score;
    out->region.query_start = region->query_start + vo.query_start;
    out->region.target_start = region->target_start + vo.target_start;
    out->region.query_length = vo.query_end - vo.query_start;
    out->region.target_length = vo.target_end - vo.target_start;
    out->valid = 1;
    path_to_alignment(out, &cap, vo.ops, vo.n_ops);
    oracle_viterbi_out_clear(&vo);
    }

static int model_is_global(const c4gpu_model *m){   /* C4_Model_is_global, c4.c:1959 */
    return (m->start_scope == C4GPU_SCOPE_CORNER) && (m->end_scope == C4GPU_SCOPE_CORNER);
    }

/* Optimal_find_path, optimal.c:368-413 */
int oracle_find_path(const c4gpu_model *model, const c4gpu_params *params,
                     const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                     int dpmemory_mb, c4gpu_score threshold, c4gpu_alignment *out){
    return oracle_find_path_subopt(model, params, query, qlen, target, tlen, dpmemory_mb, threshold, NULL, out);
    }

int oracle_find_path_subopt(const c4gpu_model *model, const c4gpu_params *params,
                     const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                     int dpmemory_mb, c4gpu_score threshold, const oracle_subopt *subopt,
                     c4gpu_alignment *out){
    c4gpu_region region = {0, 0, qlen, tlen};
    return oracle_find_path_region(model, params, query, qlen, target, tlen, &region, dpmemory_mb, threshold, subopt, out);
    }

/* Optimal_find_path (optimal.c:368-413) with its `region` argument: GAM_Result_refine_alignment's call under
 * --refine region (gam.c:618-640) */
int oracle_find_path_region(const c4gpu_model *model, const c4gpu_params *params,
                     const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                     const c4gpu_region *region_in, int dpmemory_mb, c4gpu_score threshold,
                     const oracle_subopt *subopt, c4gpu_alignment *out){
    octx cx = {model, params, query, target, qlen, tlen, dpmemory_mb, subopt};
    c4gpu_region region = *region_in;
    memset(out, 0, sizeof(*out));
    if(oracle_use_reduced_space(model, &region, dpmemory_mb)){
        c4gpu_region ar = region;
        if(!model_is_global(model)){                         /* Optimal_find_region, optimal.c:135-156 */
            oracle_viterbi_out vo;
            oracle_viterbi_subopt(model, params, C4GPU_MODE_FIND_REGION, query, qlen, target, tlen,
                           &region, NULL, 0, subopt, &vo);
            if(vo.score < threshold){
                oracle_viterbi_out_clear(&vo);
                return 0;
                }
            /* Viterbi_Data_finalise, viterbi.c:633-653 */
            if(model->start_scope != C4GPU_SCOPE_QUERY)
                ar.query_start = vo.query_start + region.query_start;
            if(model->start_scope != C4GPU_SCOPE_TARGET)
                ar.target_start = vo.target_start + region.target_start;
            ar.query_length = vo.query_end - vo.query_start;
            ar.target_length = vo.target_end - vo.target_start;
            oracle_viterbi_out_clear(&vo);
            }
        if(oracle_use_reduced_space(model, &ar, dpmemory_mb))
            find_path_reduced_space(&cx, &ar, out);
        else
            find_path_quadratic_space(&cx, &ar, out);
    } else {
        find_path_quadratic_space(&cx, &region, out);
        }
    if(out->score < threshold){
        oracle_alignment_clear(out);
        return 0;
        }
    return 1;
    }

/* ---- output formats ---------------------------------------------------------------------------- */

/* Alignment_get_coordinate, alignment.c:177-205 */
static int coordinate(const c4gpu_alignment *a, int on_query, int report_start, int len, char strand,
                      int forward_coords){
    int pos;
    if(on_query)
        pos = report_start ? a->region.query_start : (a->region.query_start + a->region.query_length);
    else
        pos = report_start ? a->region.target_start : (a->region.target_start + a->region.target_length);
    if(forward_coords && (strand == '-'))
        pos = len - pos;
    return pos;
    }

typedef struct { char *buf; size_t len, pos; int overflow; } obuf;
static void oprintf(obuf *b, const char *fmt, int x, int y, const char *gap, char c){
    char tmp[96];
    int n = snprintf(tmp, sizeof tmp, fmt, gap, c, x, y);
    if(b->pos + n + 1 > b->len){ b->overflow = 1; return; }
    memcpy(b->buf + b->pos, tmp, n + 1);
    b->pos += n;
    }

int oracle_alignment_format(const c4gpu_model *m, const c4gpu_alignment *a, int what,
                            const char *qid, int32_t qlen, char qstrand,
                            const char *tid, int32_t tlen, char tstrand,
                            int forward_coords, char *buf, size_t buf_len){
    /* Alignment_display_{sugar,cigar,vulgar}, alignment.c:2671-2706: "<what>: <sugar block>[ <block>]" */
    static const char *prefix[] = {"sugar: ", "cigar: ", "vulgar: "};
    obuf b = {buf, buf_len, 0, 0};
    int i, n;
    const char *gap = "";
    if((what < 0) || (what > 2) || (!buf_len))
        return -1;
    /* Alignment_print_sugar_block, alignment.c:1622-1639 */
    n = snprintf(buf, buf_len, "%s%s %d %d %c %s %d %d %c %d%s", prefix[what], qid,
                 coordinate(a, 1, 1, qlen, qstrand, forward_coords),
                 coordinate(a, 1, 0, qlen, qstrand, forward_coords), qstrand, tid,
                 coordinate(a, 0, 1, tlen, tstrand, forward_coords),
                 coordinate(a, 0, 0, tlen, tstrand, forward_coords), tstrand, a->score,
                 what ? " " : "");
    if(n >= (int)buf_len)
        return -1;
    b.pos = n;
    if(what == 0)
        return n;
    if(what == 1){      /* Alignment_print_cigar_block, alignment.c:1641-1681 */
        char type = 0, next_type;
        int move = 0, next_move;
        for(i = 0; i < a->n_ops; i++){
            const c4gpu_transition *t = &m->transitions[a->op_transition[i]];
            if(!t->advance_query){ next_move = t->advance_target * a->op_length[i]; next_type = 'D'; }
            else if(!t->advance_target){ next_move = t->advance_query * a->op_length[i]; next_type = 'I'; }
            else { next_move = ((t->advance_query > t->advance_target) ? t->advance_query
                                : t->advance_target) * a->op_length[i]; next_type = 'M'; }
            if(!i){ type = next_type; move = next_move; continue; }
            if(type == next_type){
                move += next_move;
            } else {
                if(move){ oprintf(&b, "%s%c %d", move, 0, gap, type); }
                move = next_move;
                type = next_type;
                gap = " ";
                }
            }
        if(move) oprintf(&b, "%s%c %d", move, 0, gap, type);
        return b.overflow ? -1 : (int)b.pos;
        }
    {   /* Alignment_print_vulgar_block, alignment.c:1683-1779 */
        static const char label_char[] = {0, 'M', 'G', 'N', '5', '3', 'I', 'S', 'F'};
        const c4gpu_transition *t = &m->transitions[a->op_transition[0]];
        int curr_label = t->label, curr_is_codon = 0;
        int aq = t->advance_query * a->op_length[0], at = t->advance_target * a->op_length[0];
        for(i = 1; i < a->n_ops; i++){
            t = &m->transitions[a->op_transition[i]];
            if((t->label == curr_label)
            && (aq || (!t->advance_query))
            && (at || (!t->advance_target))
            && (curr_is_codon == ((t->advance_query == 3) && (t->advance_target == 3)))){
                aq += t->advance_query * a->op_length[i];
                at += t->advance_target * a->op_length[i];
            } else {
                if(curr_label != C4GPU_LABEL_NONE){
                    char c = label_char[curr_label];
                    if((curr_label == C4GPU_LABEL_MATCH) && curr_is_codon) c = 'C';
                    oprintf(&b, "%s%c %d %d", aq, at, gap, c);
                    gap = " ";
                    }
                curr_label = t->label;
                curr_is_codon = ((t->advance_query == 3) && (t->advance_target == 3));
                aq = t->advance_query * a->op_length[i];
                at = t->advance_target * a->op_length[i];
                }
            }
        return b.overflow ? -1 : (int)b.pos;
        }
    }


/* ---- HSP seeding (src/comparison/hspset.c) ------------------------------------------------------------------------------ */

typedef struct { const c4gpu_params *p; int type; const uint8_t *q, *t; int qlen, tlen, aq, at; } hsp_ctx;

static int hsp_score(const hsp_ctx *h, int qpos, int tpos){            /* HSP_get_score -> match->score_func */
    const c4gpu_params *p = h->p;
    if(h->type == C4GPU_MATCH_DNA2DNA)                                 /* Match_1_1_dna_score_func, match.c:271 */
        return p->dna_submat[p->submat_index[h->q[qpos]]][p->submat_index[h->t[tpos]]];
    if(h->type == C4GPU_MATCH_PROTEIN2PROTEIN)                         /* match.c:287 */
        return p->protein_submat[p->submat_index[h->q[qpos]]][p->submat_index[h->t[tpos]]];
    {                                                                  /* Match_1_3_score_func, match.c:347 */
        uint8_t aa = p->aa[p->trans[ p->nt2d[h->t[tpos]] | (p->nt2d[h->t[tpos+1]] << 4) | (p->nt2d[h->t[tpos+2]] << 8)]];
        return p->protein_submat[p->submat_index[h->q[qpos]]][p->submat_index[aa]];
    }
    }

static void hsp_grow(const hsp_ctx *h, int seedlen, int dropoff, int qs, int ts, c4gpu_hsp *o){
    int length = seedlen, i, score, maxscore, extend, maxext, qpos, tpos;
    /* HSP_trim_ends, hspset.c:837-870 */
    for(i = 0; i < length; i++){
        if(hsp_score(h, qs, ts) > 0) break;
        qs += h->aq; ts += h->at;
        }
    length -= i;
    qpos = qs + length * h->aq - h->aq; tpos = ts + length * h->at - h->at;
    while(length > 0){
        if(hsp_score(h, qpos, tpos) > 0) break;
        length--; qpos -= h->aq; tpos -= h->at;
        }
    /* HSP_init, hspset.c:722-741 */
    score = 0;
    for(i = 0, qpos = qs, tpos = ts; i < length; i++, qpos += h->aq, tpos += h->at)
        score += hsp_score(h, qpos, tpos);
    /* HSP_extend(forbid_masked = FALSE), hspset.c:743-812: left ... */
    maxscore = score;
    qpos = qs - h->aq; tpos = ts - h->at;
    for(extend = 1, maxext = 0; (qpos >= 0) && (tpos >= 0); extend++){
        score += hsp_score(h, qpos, tpos);
        if(maxscore <= score){ maxscore = score; maxext = extend; }
        else { if(score < 0) break; if((maxscore - score) >= dropoff) break; }
        qpos -= h->aq; tpos -= h->at;
        }
    qpos = qs + length * h->aq; tpos = ts + length * h->at;            /* HSP_query_end / _target_end before the update */
    qs -= maxext * h->aq; ts -= maxext * h->at; length += maxext;
    score = maxscore;
    /* ... then right */
    for(extend = 1, maxext = 0; ((qpos + h->aq) <= h->qlen) && ((tpos + h->at) <= h->tlen); extend++){
        score += hsp_score(h, qpos, tpos);
        if(maxscore <= score){ maxscore = score; maxext = extend; }
        else { if(score < 0) break; if((maxscore - score) >= dropoff) break; }
        qpos += h->aq; tpos += h->at;
        }
    length += maxext;
    o->query_start = qs; o->target_start = ts; o->length = length; o->score = maxscore;
    /* HSP_find_cobs, hspset.c:426-441 */
    score = 0;
    for(i = 0, qpos = qs, tpos = ts; i < length; i++, qpos += h->aq, tpos += h->at){
        score += hsp_score(h, qpos, tpos);
        if(score >= (maxscore >> 1)) break;
        }
    o->cobs = i;
    }

void oracle_hsp_extend(const c4gpu_params *params, int match_type, const uint8_t *query, int32_t qlen,
                       const uint8_t *target, int32_t tlen, int32_t seedlen, int32_t dropoff,
                       int32_t query_start, int32_t target_start, c4gpu_hsp *out){
    hsp_ctx h = {params, match_type, query, target, qlen, tlen, 1, match_type == C4GPU_MATCH_PROTEIN2DNA ? 3 : 1};
    hsp_grow(&h, seedlen, dropoff, query_start, target_start, out);
    }

int32_t oracle_hsp_set(const c4gpu_params *params, int match_type, const uint8_t *query, int32_t qlen,
                       const uint8_t *target, int32_t tlen, int32_t seedlen, int32_t dropoff, int32_t threshold,
                       const int32_t *seed_q, const int32_t *seed_t, int32_t n, c4gpu_hsp *out){
    hsp_ctx h = {params, match_type, query, target, qlen, tlen, 1, match_type == C4GPU_MATCH_PROTEIN2DNA ? 3 : 1};
    /* horizon[section_pos][query_frame][target_frame], hspset.c:321-327,936-958 (seed_repeat == 1) */
    int *horizon = calloc((size_t)qlen * h.aq * h.at + 1, sizeof(int));
    int k, total = 0;
    for(k = 0; k < n; k++){
        const int diag = seed_t[k] * h.aq - seed_q[k] * h.at, qf = seed_q[k] % h.aq, tf = seed_t[k] % h.at;
        const int section = (diag + qlen) % qlen;
        /* diag + qlen < 0 (a seed more than a query length above the main diagonal: query_start * target advance >
         * target_start + query length, only possible with a target advance of 3) gives the reference a NEGATIVE
         * section_pos (hspset.c:943-944; its g_assert is compiled out) and an out-of-bounds horizon entry: undefined
         * there, so no parity target — such a seed neither reads nor writes a horizon here */
        int *hz = (section >= 0) ? &horizon[((size_t)section * h.aq + qf) * h.at + tf] : NULL;
        c4gpu_hsp o;
        if(hz && (seed_t[k] < *hz)) continue;
        hsp_grow(&h, seedlen, dropoff, seed_q[k], seed_t[k], &o);
        if(o.score >= threshold) out[total++] = o;                     /* HSP_store, hspset.c:885-888 */
        if(hz) *hz = o.target_start + o.length * h.at;                 /* HSP_target_end */
        }
    free(horizon);
    return total;
    }

/* ---- SDP (src/sdp/): restated in its own file, same translation unit ------------------------------------------- */
#include "c4_oracle_sdp.c"

/* ---- the seeder's automaton walk (src/comparison/seeder.c:649-720,852-915): its own file, same translation unit ---- */
#include "c4_oracle_seed.c"
