/* c4_oracle_sdp.c — CPU restatement of the reference's SDP (seeded dynamic programming, src/sdp/): the default
 * gapped-extension heuristic behind GAM_Result_SDP_create (src/hub/gam.c:852-890).  TEST INFRASTRUCTURE ONLY (see
 * c4_oracle.h); #included at the end of c4_oracle.c (it uses that file's calc_score / SubOpt restatements).
 *
 * What is restated, in the reference's own order:
 *   Scheduler_Pair_calculate   src/sdp/scheduler.c:1445-1500  rows (target positions) in ascending order, seeds joined
 *                                                             as the rows reach them
 *   Scheduler_Row_process      scheduler.c:1113-1166          cells (query positions) of a row in ascending order
 *   Scheduler_Cell_process     scheduler.c:859-1065           push-style cell: transitions from last to first, span
 *                                                             freeze (:890-921) / thaw (:940-986), pruning (:1020-1027),
 *                                                             "keep higher, no tie-break on max" (:1047-1051)
 *   Scheduler_Cell_assign      scheduler.c:763-837            score / max / seed id, shadow start + transport, the
 *                                                             start / end call-backs on a new path maximum
 *   Scheduler_SpanData_*       scheduler.c:567-645            the best span seed in scope, the per-query-position store
 *   Scheduler_Row_destroy      scheduler.c:1168-1247          boundary cells of the reverse pass
 *   Boundary_*                 src/sdp/boundary.c:153-166,437-490
 *   SDP_Pair_*                 src/sdp/sdp.c:479-815          seed list, the two passes, single-pass sub-optimal loop
 * The reference's Lookahead windows / SList used+unused lists (lookahead.c, scheduler.c:1249-1321) implement an ordered set
 * of cells per row and of rows per pair; seeds always reach a row before any pushed cell does (a seed is taken as soon as
 * it is within max_target_advance of the row being processed, scheduler.c:1470-1476, and nothing is pushed further than
 * that), so dense per-row arrays scanned in ascending order visit the same cells in the same order.  Traceback cells
 * (straceback.c) are a persistent list; the run-length merge of scheduler.c:1180-1196 only touches unshared cells and is
 * not observable through Alignment_add.
 */

#define SDP_SH 3                    /* scheduler->shadow_start: score, max, seed (scheduler.c:72) */
#define SDP_MAX_SPANS 4

typedef struct { int32_t transition, length, prev; } sdp_tbcell;           /* STraceback_Cell, straceback.h:30 */
typedef struct { sdp_tbcell *c; int32_t n, cap; } sdp_tb;

static int32_t sdp_tb_add(sdp_tb *tb, int transition, int length, int32_t prev){     /* STraceback_add, straceback.c:43 */
    if(tb->n == tb->cap){
        tb->cap = tb->cap ? tb->cap * 2 : 4096;
        tb->c = realloc(tb->c, sizeof(sdp_tbcell) * tb->cap);
        }
    tb->c[tb->n].transition = transition;
    tb->c[tb->n].length = length;
    tb->c[tb->n].prev = prev;
    return tb->n++;
    }

typedef struct { int32_t state, min_query, max_query, min_target, max_target, query_loop, target_loop; } sdp_span;

typedef struct {                                                          /* Scheduler_SpanSeed, scheduler.h:112 */
    int present;
    c4gpu_score score, max;
    int32_t seed_id, query_entry, target_entry, cell;
    c4gpu_score shadow[C4GPU_MAX_SHADOWS];
} sdp_span_seed;

typedef struct {                                                          /* Scheduler_Cell, scheduler.h:174 */
    int32_t query_pos;                                                    /* negated in the reverse pass */
    int permit_span_thaw;
    c4gpu_score score[C4GPU_MAX_STATES][SDP_SH + C4GPU_MAX_SHADOWS];
    int32_t tb[C4GPU_MAX_STATES];
} sdp_cell;

typedef struct { int32_t target_pos; sdp_cell **cells; } sdp_row;         /* Scheduler_Row: cells by relative position */

typedef struct { int32_t query_pos, length, seed_id; } sdp_interval;      /* Boundary_Interval, boundary.h */
typedef struct { int32_t target_pos, n, cap; sdp_interval *iv; } sdp_brow;
typedef struct { sdp_brow *rows; int32_t n, cap; } sdp_boundary;

typedef struct { int32_t query_pos, target_pos; c4gpu_score score; int32_t cell; } sdp_terminal;   /* sdp.h:48 */
typedef struct { int32_t seed_id; const c4gpu_hsp *hsp; int32_t qcobs, tcobs; sdp_terminal max_start, max_end; } sdp_seed;
typedef struct { int32_t query_pos, target_pos, seed_id; c4gpu_score start_score; } sdp_sseed;      /* Scheduler_Seed */

typedef struct {
    const c4gpu_model *m;
    odata *od;
    int is_forward, has_traceback, use_boundary, start_func, end_func;
    c4gpu_score dropoff;
    int32_t qlen, tlen;
    int n_spans;
    sdp_span spans[SDP_MAX_SPANS];
    int span_map[C4GPU_MAX_STATES];
    sdp_tb *tb;
    sdp_row **rows;                         /* by relative target position (+ tlen in the reverse pass) */
    sdp_span_seed *cache[SDP_MAX_SPANS];    /* span_seed_cache: [query position] */
    int32_t curr[SDP_MAX_SPANS];            /* curr_span_seed: query position of the stored seed it points at, or -1 */
    osoi *soi;
    sdp_boundary *boundary;
    sdp_seed *seeds;
} sdp_sched;

/* SubOpt_Index_is_blocked, subopt.c:376-390 (the cursor may move both ways, unlike the _fast macro) */
static int osoi_is_blocked(osoi *x, int32_t q_pos){
    while(x->curr_row->query_pos[x->curr_query_index] < q_pos)
        x->curr_query_index++;
    while(x->curr_row->query_pos[x->curr_query_index] > q_pos){
        if(!x->curr_query_index)
            break;
        x->curr_query_index--;
        }
    return x->curr_row->query_pos[x->curr_query_index] == q_pos;
    }

static int sdp_is_span(const c4gpu_transition *tr){                       /* C4_Transition_is_span, c4.h:246 */
    return (tr->input == tr->output) && (tr->calc < 0);
    }

/* C4_Span list of the model (intron.c:660-672: one per intron state, target span min..max intron) in span_list order;
 * Scheduler_get_span_map, scheduler.c:31 */
static void sdp_find_spans(sdp_sched *S, const c4gpu_params *p){
    const c4gpu_model *m = S->m;
    int s, k;
    S->n_spans = 0;
    for(s = 0; s < m->n_states; s++){
        int ql = -1, tl = -1;
        S->span_map[s] = -1;
        for(k = 0; k < m->n_transitions; k++){
            const c4gpu_transition *tr = &m->transitions[k];
            if(sdp_is_span(tr) && (tr->input == s)){
                if(tr->advance_query) ql = k;
                if(tr->advance_target) tl = k;
                }
            }
        if((ql >= 0) || (tl >= 0)){
            sdp_span *sp = &S->spans[S->n_spans];
            sp->state = s;
            sp->query_loop = ql; sp->target_loop = tl;
            sp->min_query = (ql >= 0) ? p->min_intron : 0; sp->max_query = (ql >= 0) ? p->max_intron : 0;
            sp->min_target = (tl >= 0) ? p->min_intron : 0; sp->max_target = (tl >= 0) ? p->max_intron : 0;
            S->span_map[s] = S->n_spans++;
            }
        }
    }

/* SDP_create, sdp.c:322-341: bidirectional seeded SDP only for a model without shadows and spans whose one portal has one
 * transition (affine.c:245: the match self-transition); everything else goes through the boundary */
static int sdp_use_boundary(const sdp_sched *S){
    const c4gpu_model *m = S->m;
    int k, loops = 0;
    if(m->n_shadows || S->n_spans)
        return 1;
    for(k = 0; k < m->n_transitions; k++){
        const c4gpu_transition *tr = &m->transitions[k];
        if((tr->input == tr->output) && (tr->label == C4GPU_LABEL_MATCH) && (tr->calc >= 0))
            loops++;
        }
    return loops != 1;
    }

static sdp_row *sdp_row_get(sdp_sched *S, int32_t rel_t, int create){
    const int32_t idx = rel_t + (S->is_forward ? 0 : S->tlen);
    if((!S->rows[idx]) && create){
        S->rows[idx] = calloc(1, sizeof(sdp_row));
        S->rows[idx]->target_pos = rel_t;
        S->rows[idx]->cells = calloc((size_t)S->qlen + 1, sizeof(sdp_cell*));
        }
    return S->rows[idx];
    }

static sdp_cell **sdp_cell_slot(sdp_sched *S, sdp_row *row, int32_t rel_q){
    return &row->cells[rel_q + (S->is_forward ? 0 : S->qlen)];
    }

static sdp_cell *sdp_cell_create(sdp_sched *S, int32_t rel_q, int permit_span_thaw){   /* Scheduler_Cell_init :684 */
    sdp_cell *cell = calloc(1, sizeof(sdp_cell));
    int s;
    cell->query_pos = rel_q;
    cell->permit_span_thaw = permit_span_thaw;
    for(s = 0; s < S->m->n_states; s++){
        cell->score[s][0] = LOW;
        cell->tb[s] = -1;
        }
    return cell;
    }

/* ---- boundary (boundary.c) ---- */
static sdp_brow *sdp_boundary_add_row(sdp_boundary *b, int32_t target_pos){           /* Boundary_add_row :437 */
    if(b->n == b->cap){
        b->cap = b->cap ? b->cap * 2 : 64;
        b->rows = realloc(b->rows, sizeof(sdp_brow) * b->cap);
        }
    memset(&b->rows[b->n], 0, sizeof(sdp_brow));
    b->rows[b->n].target_pos = target_pos;
    return &b->rows[b->n++];
    }

static void sdp_brow_prepend(sdp_brow *r, int32_t query_pos, int32_t seed_id){         /* Boundary_Row_prepend :153 */
    sdp_interval *last = r->n ? &r->iv[r->n-1] : NULL;
    if(last && (last->seed_id == seed_id) && ((last->query_pos - 1) == query_pos)){
        last->query_pos = query_pos;
        last->length++;
        return;
        }
    if(r->n == r->cap){
        r->cap = r->cap ? r->cap * 2 : 8;
        r->iv = realloc(r->iv, sizeof(sdp_interval) * r->cap);
        }
    r->iv[r->n].query_pos = query_pos; r->iv[r->n].length = 1; r->iv[r->n].seed_id = seed_id;
    r->n++;
    }

static void sdp_boundary_reverse(sdp_boundary *b){                                     /* Boundary_reverse :474 */
    int32_t a, z, i;
    for(a = 0, z = b->n - 1; a < z; a++, z--){
        sdp_brow sw = b->rows[a]; b->rows[a] = b->rows[z]; b->rows[z] = sw;
        }
    for(i = 0; i < b->n; i++)
        for(a = 0, z = b->rows[i].n - 1; a < z; a++, z--){
            sdp_interval sw = b->rows[i].iv[a]; b->rows[i].iv[a] = b->rows[i].iv[z]; b->rows[i].iv[z] = sw;
            }
    }

static void sdp_boundary_free(sdp_boundary *b){
    int32_t i;
    if(!b) return;
    for(i = 0; i < b->n; i++)
        free(b->rows[i].iv);
    free(b->rows); free(b);
    }

/* ---- span seeds ---- */
/* Scheduler_SpanData_get_curr, scheduler.c:567-613 */
static void sdp_span_get_curr(sdp_sched *S, int sp, int32_t query_pos, int32_t target_pos){
    const sdp_span *span = &S->spans[sp];
    sdp_span_seed *stored;
    if(S->curr[sp] >= 0){
        const sdp_span_seed *c = &S->cache[sp][S->curr[sp]];
        if((c->query_entry > query_pos) || ((c->query_entry + span->max_query) < query_pos)
        || ((c->target_entry + span->max_target) < target_pos))
            S->curr[sp] = -1;
        }
    stored = &S->cache[sp][query_pos];
    if(stored->present){
        if((stored->target_entry + span->max_target) >= target_pos){
            if(S->curr[sp] >= 0){
                if(S->cache[sp][S->curr[sp]].score < stored->score)
                    S->curr[sp] = query_pos;
            } else {
                S->curr[sp] = query_pos;
                }
        } else {
            stored->present = 0;
            }
        }
    }

/* Scheduler_SpanData_submit, scheduler.c:619-643 */
static void sdp_span_submit(sdp_sched *S, int sp, const sdp_span_seed *seed){
    sdp_span_seed *stored;
    if(!S->spans[sp].max_target)
        return;
    stored = &S->cache[sp][seed->query_entry];
    if(stored->present){
        if(stored->score <= seed->score)
            *stored = *seed;
    } else {
        *stored = *seed;
        }
    stored->present = 1;
    }

/* ---- the call-backs of sdp.c:110-153,271-289 ---- */
static void sdp_start_func(sdp_sched *S, int32_t seed_id, c4gpu_score score, int32_t q, int32_t t, int32_t cell){
    sdp_seed *seed = &S->seeds[seed_id];
    if(seed->max_start.score < score){
        seed->max_start.score = score; seed->max_start.query_pos = q; seed->max_start.target_pos = t;
        seed->max_start.cell = cell;
        }
    }

static void sdp_end_func(sdp_sched *S, int32_t seed_id, c4gpu_score score, int32_t q, int32_t t, int32_t cell){
    sdp_seed *seed = &S->seeds[seed_id];
    if(seed->max_end.score < score){
        seed->max_end.score = score; seed->max_end.query_pos = q; seed->max_end.target_pos = t;
        seed->max_end.cell = cell;
        }
    }

/* Scheduler_Cell_assign, scheduler.c:763-837 */
static void sdp_cell_assign(sdp_sched *S, sdp_cell *src, int input_pos, sdp_cell *dst, int output_pos,
                            c4gpu_score dst_score, c4gpu_score max_score, int transition, int32_t seed_id,
                            int32_t dst_q, int32_t dst_t){
    const c4gpu_model *m = S->m;
    const c4gpu_transition *tr = &m->transitions[transition];
    int l;
    dst->score[output_pos][0] = dst_score;
    dst->score[output_pos][2] = seed_id;
    if(S->has_traceback)
        dst->tb[output_pos] = sdp_tb_add(S->tb, transition, 1, src->tb[input_pos]);
    if(S->is_forward){
        for(l = 0; l < m->n_shadows; l++)                                  /* Scheduler_Cell_shadow_start :733 */
            if(m->shadows[l].src_state_mask & (1u << tr->input))
                src->score[input_pos][SDP_SH + m->shadows[l].designation] = m->shadows[l].on_target
                    ? (dst_t - tr->advance_target) : (dst_q - tr->advance_query);
        for(l = 0; l < m->total_shadow_designations; l++)
            dst->score[output_pos][SDP_SH + l] = src->score[input_pos][SDP_SH + l];
        }
    if(dst_score < max_score){
        dst->score[output_pos][1] = max_score;
    } else {
        dst->score[output_pos][1] = dst_score;
        if(S->start_func && (tr->input == m->start_state))
            sdp_start_func(S, seed_id, dst_score, dst_q, dst_t, S->has_traceback ? dst->tb[output_pos] : -1);
        if(S->end_func && (tr->output == m->end_state))
            sdp_end_func(S, seed_id, dst_score, dst_q, dst_t, dst->tb[output_pos]);
        }
    }

/* Scheduler_Cell_process, scheduler.c:859-1065 */
static void sdp_cell_process(sdp_sched *S, sdp_cell *cell, sdp_row *row){
    const c4gpu_model *m = S->m;
    int i, l, input_pos, output_pos;
    int32_t src_q, src_t, dst_q, dst_t, rel_q, rel_t, seed_id;
    c4gpu_score src_score, dst_score, tscore, max_score;
    if(S->is_forward){ src_q = cell->query_pos; src_t = row->target_pos; }
    else { src_q = -cell->query_pos; src_t = -row->target_pos; }
    for(i = m->n_transitions - 1; i >= 0; i--){
        const c4gpu_transition *tr = &m->transitions[i];
        sdp_row *dst_row;
        sdp_cell **slot, *dst_cell;
        if(sdp_is_span(tr)){
            if(S->is_forward && S->use_boundary){                          /* freeze, :890-921 */
                const int sp = S->span_map[tr->output];
                if(sp >= 0){
                    input_pos = tr->input;
                    if(cell->score[input_pos][0] >= 0){
                        sdp_span_seed seed;
                        memset(&seed, 0, sizeof(seed));
                        seed.score = cell->score[input_pos][0];
                        seed.max = cell->score[input_pos][1];
                        seed.seed_id = cell->score[input_pos][2];
                        seed.cell = cell->tb[input_pos];
                        seed.query_entry = src_q; seed.target_entry = src_t;
                        for(l = 0; l < m->total_shadow_designations; l++)
                            seed.shadow[l] = cell->score[input_pos][SDP_SH + l];
                        sdp_span_submit(S, sp, &seed);
                        }
                    }
                }
            continue;
            }
        if(S->is_forward){
            dst_q = src_q + tr->advance_query; dst_t = src_t + tr->advance_target;
            if((dst_q > S->qlen) || (dst_t > S->tlen))
                continue;
            input_pos = tr->input; output_pos = tr->output;
            rel_q = dst_q; rel_t = dst_t;
            if(cell->permit_span_thaw){                                    /* thaw, :940-986 */
                const int sp = S->span_map[tr->input];
                if(sp >= 0){
                    sdp_span_get_curr(S, sp, cell->query_pos, row->target_pos);
                    if((S->curr[sp] >= 0) && (cell->score[input_pos][0] < S->cache[sp][S->curr[sp]].score)){
                        const sdp_span_seed *c = &S->cache[sp][S->curr[sp]];
                        int32_t prev = c->cell;
                        cell->score[input_pos][0] = c->score;
                        cell->score[input_pos][1] = c->max;
                        cell->score[input_pos][2] = c->seed_id;
                        if(src_q - c->query_entry)                         /* Scheduler_Cell_add_span :839 */
                            prev = sdp_tb_add(S->tb, S->spans[sp].query_loop, src_q - c->query_entry, prev);
                        if(src_t - c->target_entry)
                            prev = sdp_tb_add(S->tb, S->spans[sp].target_loop, src_t - c->target_entry, prev);
                        cell->tb[input_pos] = prev;
                        for(l = 0; l < m->total_shadow_designations; l++)
                            cell->score[input_pos][SDP_SH + l] = c->shadow[l];
                        }
                    }
                }
            for(l = 0; l < m->n_shadows; l++)                              /* Scheduler_Cell_shadow_end :749 */
                if(tr->dst_shadow_mask & (1u << l))
                    S->od->curr_intron_start = cell->score[input_pos][SDP_SH + m->shadows[l].designation];
            tscore = calc_score(S->od, tr->calc, src_q, src_t);
        } else {
            dst_q = src_q - tr->advance_query; dst_t = src_t - tr->advance_target;
            if((dst_q < 0) || (dst_t < 0))
                continue;
            rel_q = -dst_q; rel_t = -dst_t;
            input_pos = tr->output; output_pos = tr->input;
            if(tr->dst_shadow_mask)                                        /* :1004-1006 */
                tscore = 0;
            else
                tscore = calc_score(S->od, tr->calc, dst_q, dst_t);
            }
        src_score = cell->score[input_pos][0];
        max_score = cell->score[input_pos][1];
        seed_id = cell->score[input_pos][2];
        dst_score = src_score + tscore;
        if(S->is_forward && (dst_score < 0))                               /* :1020-1022 */
            continue;
        if((max_score - dst_score) > S->dropoff)                           /* :1023-1024 */
            continue;
        if((tr->label == C4GPU_LABEL_MATCH) && S->soi && osoi_is_blocked(S->soi, src_q))   /* :1026-1032 */
            continue;
        dst_row = sdp_row_get(S, rel_t, 1);
        slot = sdp_cell_slot(S, dst_row, rel_q);
        dst_cell = *slot;
        if(dst_cell){
            if(dst_score <= dst_cell->score[output_pos][0])                /* :1047-1051 */
                continue;
        } else {
            dst_cell = *slot = sdp_cell_create(S, rel_q, 0);
            }
        sdp_cell_assign(S, cell, input_pos, dst_cell, output_pos, dst_score, max_score, i, seed_id, dst_q, dst_t);
        }
    }

/* Scheduler_Row_destroy + Scheduler_Row_traverse_cell_destroy, scheduler.c:1168-1247 */
static void sdp_row_destroy(sdp_sched *S, sdp_row *row){
    const c4gpu_model *m = S->m;
    sdp_brow *brow = NULL;
    int32_t k;
    int sp;
    if((!S->is_forward) && S->boundary)
        brow = sdp_boundary_add_row(S->boundary, -row->target_pos);
    for(k = 0; k <= S->qlen; k++){
        sdp_cell *cell = row->cells[k];
        if(!cell)
            continue;
        if(brow){
            if(cell->score[m->start_state][0] >= 0){
                sdp_brow_prepend(brow, -cell->query_pos, cell->score[m->start_state][2]);
            } else {
                for(sp = 0; sp < S->n_spans; sp++)
                    if(cell->score[S->spans[sp].state][0] > 0){
                        sdp_brow_prepend(brow, -cell->query_pos, cell->score[S->spans[sp].state][2]);
                        break;
                        }
                }
            }
        free(cell);
        }
    if(brow && (!brow->n)){                                                /* Boundary_remove_empty_last_row :457 */
        free(brow->iv);
        S->boundary->n--;
        }
    S->rows[row->target_pos + (S->is_forward ? 0 : S->tlen)] = NULL;
    free(row->cells);
    free(row);
    }

/* Scheduler_Row_add_seed + Scheduler_Cell_seed, scheduler.c:1068-1082,1249-1281 */
static void sdp_add_seed(sdp_sched *S, const sdp_sseed *seed){
    sdp_row *row = sdp_row_get(S, seed->target_pos, 1);
    sdp_cell **slot = sdp_cell_slot(S, row, seed->query_pos), *cell;
    const int st = S->is_forward ? S->m->start_state : S->m->end_state;
    if(*slot){
        fprintf(stderr, "oracle sdp: seed on an existing cell (%d,%d)\n", seed->query_pos, seed->target_pos);
        abort();
        }
    cell = *slot = sdp_cell_create(S, seed->query_pos, S->is_forward && S->use_boundary);
    cell->score[st][0] = seed->start_score;
    cell->score[st][1] = seed->start_score;
    cell->score[st][2] = seed->seed_id;
    cell->tb[st] = -1;
    }

/* Scheduler_Pair_calculate, scheduler.c:1445-1500 */
static void sdp_calculate(sdp_sched *S, const sdp_sseed *seeds, int32_t n_seeds){
    const int mta = S->m->max_target_advance;
    const int32_t lo = S->is_forward ? 0 : -S->tlen, hi = S->is_forward ? S->tlen : 0;
    int32_t si = 0, cur = 0, k, q;
    int have = 0, sp;
    S->rows = calloc((size_t)S->tlen + 1, sizeof(sdp_row*));
    for(sp = 0; sp < S->n_spans; sp++){
        S->cache[sp] = (S->is_forward && S->use_boundary) ? calloc((size_t)S->qlen + 1, sizeof(sdp_span_seed)) : NULL;
        S->curr[sp] = -1;
        }
    for(;;){
        sdp_row *row;
        if(!have){
            if(si >= n_seeds)
                break;
            cur = seeds[si].target_pos;
            sdp_add_seed(S, &seeds[si++]);
            have = 1;
            }
        while((si < n_seeds) && ((seeds[si].target_pos - cur) <= mta))
            sdp_add_seed(S, &seeds[si++]);
        row = sdp_row_get(S, cur, 0);
        osoi_set_row(S->soi, S->is_forward ? row->target_pos : -row->target_pos);      /* Scheduler_Row_process :1119 */
        for(q = 0; q <= S->qlen; q++)
            if(row->cells[q])
                sdp_cell_process(S, row->cells[q], row);
        sdp_row_destroy(S, row);
        have = 0;                                                          /* Lookahead_next: next occupied row in reach */
        for(k = cur + 1; (k <= cur + mta) && (k <= hi); k++)
            if((k >= lo) && sdp_row_get(S, k, 0)){
                cur = k; have = 1;
                break;
                }
        }
    for(sp = 0; sp < S->n_spans; sp++)
        free(S->cache[sp]);
    free(S->rows);
    S->rows = NULL;
    }

/* ---- SDP_Pair (sdp.c) ---- */
typedef struct {
    sdp_sched S;
    odata od;
    sdp_seed *seeds; int32_t n_seeds;
    sdp_seed **by_score; int32_t single_pass_pos;
    sdp_boundary *boundary;
    sdp_tb fwd_tb, rev_tb;
    int32_t alignment_count;
    int singlepass;
    const oracle_subopt *subopt;
} sdp_pair;

static int sdp_hsp_cmp(const void *a, const void *b){                      /* sdp.c:425-436 */
    const sdp_seed *x = *(sdp_seed * const *)a, *y = *(sdp_seed * const *)b;
    const int td = x->tcobs - y->tcobs;
    return td ? td : (x->qcobs - y->qcobs);
    }

static int sdp_score_cmp(const void *a, const void *b){                    /* sdp.c:736-741 */
    const sdp_seed *x = *(sdp_seed * const *)a, *y = *(sdp_seed * const *)b;
    return y->max_end.score - x->max_end.score;
    }

static void sdp_terminal_init(sdp_terminal *t){
    t->query_pos = 0; t->target_pos = 0; t->score = LOW; t->cell = -1;
    }

/* SDP_Pair_find_start_points, sdp.c:538-560 */
static void sdp_find_start_points(sdp_pair *P){
    sdp_sched *S = &P->S;
    sdp_sseed *ss = malloc(sizeof(sdp_sseed) * P->n_seeds);
    int32_t k;
    for(k = 0; k < P->n_seeds; k++){                                       /* Scheduler_Seed_List_get_reverse :95 */
        const sdp_seed *seed = &P->seeds[P->n_seeds - 1 - k];
        ss[k].query_pos = -seed->qcobs; ss[k].target_pos = -seed->tcobs;
        ss[k].seed_id = seed->seed_id; ss[k].start_score = seed->hsp->score >> 1;
        }
    S->is_forward = 0;
    S->has_traceback = !S->use_boundary;                                   /* sdp.c:344-366 */
    S->start_func = !S->use_boundary; S->end_func = 0;
    S->tb = &P->rev_tb;
    sdp_boundary_free(P->boundary);
    P->boundary = S->use_boundary ? calloc(1, sizeof(sdp_boundary)) : NULL;
    S->boundary = P->boundary;
    S->soi = osoi_create(P->subopt, &(c4gpu_region){0, 0, S->qlen, S->tlen});
    sdp_calculate(S, ss, P->n_seeds);
    osoi_destroy(S->soi);
    S->soi = NULL;
    if(P->boundary)
        sdp_boundary_reverse(P->boundary);
    free(ss);
    }

/* SDP_Pair_find_end_points, sdp.c:562-600 */
static void sdp_find_end_points(sdp_pair *P){
    sdp_sched *S = &P->S;
    sdp_sseed *ss;
    int32_t n = 0, k, r, i, p;
    if(P->boundary){                                                       /* Scheduler_Seed_Boundary_*, sdp.c:188-268 */
        for(r = 0; r < P->boundary->n; r++)
            for(i = 0; i < P->boundary->rows[r].n; i++)
                n += P->boundary->rows[r].iv[i].length;
        ss = malloc(sizeof(sdp_sseed) * (n ? n : 1));
        n = 0;
        for(r = 0; r < P->boundary->n; r++)
            for(i = 0; i < P->boundary->rows[r].n; i++)
                for(p = 0; p < P->boundary->rows[r].iv[i].length; p++){
                    ss[n].query_pos = P->boundary->rows[r].iv[i].query_pos + p;
                    ss[n].target_pos = P->boundary->rows[r].target_pos;
                    ss[n].seed_id = P->boundary->rows[r].iv[i].seed_id;
                    ss[n].start_score = 0;
                    n++;
                    }
    } else {
        n = P->n_seeds;
        ss = malloc(sizeof(sdp_sseed) * n);
        for(k = 0; k < n; k++){                                            /* Scheduler_Seed_List_get_forward :79 */
            const sdp_seed *seed = &P->seeds[k];
            ss[k].query_pos = seed->qcobs; ss[k].target_pos = seed->tcobs;
            ss[k].seed_id = seed->seed_id;
            ss[k].start_score = seed->max_start.score - (seed->hsp->score >> 1);
            }
        }
    S->is_forward = 1;
    S->has_traceback = 1;
    S->start_func = 0; S->end_func = 1;
    S->tb = &P->fwd_tb;
    S->boundary = NULL;
    S->soi = osoi_create(P->subopt, &(c4gpu_region){0, 0, S->qlen, S->tlen});
    sdp_calculate(S, ss, n);
    osoi_destroy(S->soi);
    S->soi = NULL;
    free(ss);
    }

/* SubOpt_overlaps_alignment, subopt.c:177-203 (RangeTree_find: start <= point < start + length, rangetree.c:70-79) */
static int sdp_overlaps(const oracle_subopt *so, const c4gpu_model *m, const c4gpu_alignment *a){
    int32_t qp = a->region.query_start, tp = a->region.target_start, k, j, x;
    for(k = 0; k < a->n_ops; k++){
        const c4gpu_transition *tr = &m->transitions[a->op_transition[k]];
        if(tr->label == C4GPU_LABEL_MATCH){
            for(j = 0; j < a->op_length[k]; j++){
                for(x = 0; x < so->n; x++)
                    if((so->q[x] >= qp) && (so->q[x] < qp + tr->advance_query)
                    && (so->t[x] >= tp) && (so->t[x] < tp + tr->advance_target))
                        return 1;
                qp += tr->advance_query; tp += tr->advance_target;
                }
        } else {
            qp += tr->advance_query * a->op_length[k];
            tp += tr->advance_target * a->op_length[k];
            }
        }
    return 0;
    }

/* SDP_Pair_find_path + SDP_Pair_add_traceback + SDP_Seed_find_start, sdp.c:640-734 */
static void sdp_find_path(sdp_pair *P, sdp_seed *best, c4gpu_alignment *a){
    const c4gpu_model *m = P->S.m;
    int cap = 0;
    int32_t c, n, *chain;
    memset(a, 0, sizeof(*a));
    if(P->S.use_boundary){
        best->max_start.query_pos = best->max_end.query_pos;
        best->max_start.target_pos = best->max_end.target_pos;
        c = best->max_end.cell;
        do {                                                               /* SDP_Seed_find_start, sdp.c:640-659 */
            const c4gpu_transition *tr = &m->transitions[P->fwd_tb.c[c].transition];
            best->max_start.query_pos -= tr->advance_query * P->fwd_tb.c[c].length;
            best->max_start.target_pos -= tr->advance_target * P->fwd_tb.c[c].length;
            c = P->fwd_tb.c[c].prev;
        } while(m->transitions[P->fwd_tb.c[c].transition].input != m->start_state);
        }
    a->score = best->max_end.score;
    a->region.query_start = best->max_start.query_pos;
    a->region.target_start = best->max_start.target_pos;
    a->region.query_length = best->max_end.query_pos - best->max_start.query_pos;
    a->region.target_length = best->max_end.target_pos - best->max_start.target_pos;
    a->valid = 1;
    if(!P->S.use_boundary){
        /* reverse traceback: from the cell that leaves START towards the seed, without the last one (into END) */
        for(c = best->max_start.cell; (c >= 0) && (P->rev_tb.c[c].prev >= 0); c = P->rev_tb.c[c].prev)
            alignment_add(a, &cap, P->rev_tb.c[c].transition, P->rev_tb.c[c].length);
        }
    n = 0;
    for(c = best->max_end.cell; c >= 0; c = P->fwd_tb.c[c].prev)
        n++;
    chain = malloc(sizeof(int32_t) * (n ? n : 1));
    n = 0;
    for(c = best->max_end.cell; c >= 0; c = P->fwd_tb.c[c].prev)
        chain[n++] = c;
    for(c = n - 1 - (P->S.use_boundary ? 0 : 1); c >= 0; c--)              /* 1st operation only with a boundary */
        alignment_add(a, &cap, P->fwd_tb.c[chain[c]].transition, P->fwd_tb.c[chain[c]].length);
    free(chain);
    }

/* SDP_Pair_next_path, sdp.c:743-815 */
static int sdp_next_path(sdp_pair *P, c4gpu_score threshold, c4gpu_alignment *a){
    sdp_seed *best = NULL;
    int32_t i;
    if(P->alignment_count){
        if(!P->singlepass){                                                /* SDP_Pair_update_starts/_ends :602-636 */
            for(i = 0; i < P->n_seeds; i++){
                P->seeds[i].max_start.score = LOW;
                if(!P->S.use_boundary)
                    P->seeds[i].max_start.cell = -1;
                }
            sdp_find_start_points(P);
            for(i = 0; i < P->n_seeds; i++){
                P->seeds[i].max_end.score = LOW;
                P->seeds[i].max_end.cell = -1;
                }
            sdp_find_end_points(P);
            }
    } else {
        sdp_find_start_points(P);
        sdp_find_end_points(P);
        if(P->singlepass){
            P->by_score = malloc(sizeof(sdp_seed*) * P->n_seeds);
            for(i = 0; i < P->n_seeds; i++)
                P->by_score[i] = &P->seeds[i];
            qsort(P->by_score, P->n_seeds, sizeof(sdp_seed*), sdp_score_cmp);
            P->single_pass_pos = 0;
            }
        }
    if(P->singlepass){
        while(P->single_pass_pos < P->n_seeds){
            best = P->by_score[P->single_pass_pos++];
            if(best->max_end.score < threshold)
                return 0;
            sdp_find_path(P, best, a);
            if(sdp_overlaps(P->subopt, P->S.m, a)){
                oracle_alignment_clear(a);
                best = NULL;
            } else {
                break;
                }
            }
        if(!best)
            return 0;
    } else {
        best = &P->seeds[0];
        for(i = 1; i < P->n_seeds; i++)
            if(best->max_end.score < P->seeds[i].max_end.score)
                best = &P->seeds[i];
        if(best->max_end.score < threshold)
            return 0;
        sdp_find_path(P, best, a);
        }
    P->alignment_count++;
    best->max_end.score = LOW;
    return 1;
    }

/* GAM_Result_SDP_create's loop (gam.c:868-881) on the HSPs of one pair: up to max_alignments alignments into out[].
 * hsps: the combined HSP list in the order SDP_Pair_create_seed_list meets them (dna, protein, codon; sdp.c:447-463);
 * query_advance / target_advance: the HSPset's match advances (HSP_query_cobs / HSP_target_cobs, hspset.h:93-99). */
int32_t oracle_sdp(const c4gpu_model *model, const c4gpu_params *params,
                   const uint8_t *query, int32_t qlen, const uint8_t *target, int32_t tlen,
                   const c4gpu_hsp *hsps, int32_t n_hsps, int32_t query_advance, int32_t target_advance,
                   int32_t dropoff, int32_t singlepass, c4gpu_score threshold, int32_t max_alignments,
                   c4gpu_alignment *out, int32_t *use_boundary){
    sdp_pair P;
    sdp_seed *all, **sorted;
    oracle_subopt *so;
    int32_t k, n = 0;
    memset(&P, 0, sizeof(P));
    if(n_hsps <= 0)
        return 0;
    odata_init(&P.od, model, params, query, qlen, target, tlen);
    P.S.m = model; P.S.od = &P.od;
    P.S.qlen = qlen; P.S.tlen = tlen;
    P.S.dropoff = dropoff;
    sdp_find_spans(&P.S, params);
    P.S.use_boundary = sdp_use_boundary(&P.S);
    if(use_boundary)
        *use_boundary = P.S.use_boundary;
    /* SDP_Pair_create_seed_list, sdp.c:438-477: HSPs sorted on their cobs point in DP order, one seed per point */
    all = malloc(sizeof(sdp_seed) * n_hsps);
    sorted = malloc(sizeof(sdp_seed*) * n_hsps);
    for(k = 0; k < n_hsps; k++){
        all[k].hsp = &hsps[k];
        all[k].qcobs = hsps[k].query_start + hsps[k].cobs * query_advance;
        all[k].tcobs = hsps[k].target_start + hsps[k].cobs * target_advance;
        sorted[k] = &all[k];
        }
    qsort(sorted, n_hsps, sizeof(sdp_seed*), sdp_hsp_cmp);
    P.seeds = malloc(sizeof(sdp_seed) * n_hsps);
    for(k = 0; k < n_hsps; k++)
        if((!k) || (sorted[k]->qcobs != sorted[k-1]->qcobs) || (sorted[k]->tcobs != sorted[k-1]->tcobs)){
            P.seeds[P.n_seeds] = *sorted[k];
            P.seeds[P.n_seeds].seed_id = P.n_seeds;
            sdp_terminal_init(&P.seeds[P.n_seeds].max_start);
            sdp_terminal_init(&P.seeds[P.n_seeds].max_end);
            P.n_seeds++;
            }
    free(sorted); free(all);
    P.S.seeds = P.seeds;
    P.singlepass = singlepass;
    so = oracle_subopt_create(qlen, tlen);
    P.subopt = so;
    while(n < max_alignments){
        if(!sdp_next_path(&P, threshold, &out[n]))
            break;
        oracle_subopt_add_alignment(so, model, &out[n]);                   /* GAM_Result_add_alignment, gam.c:673 */
        n++;
        }
    oracle_subopt_destroy(so);
    sdp_boundary_free(P.boundary);
    free(P.by_score); free(P.seeds);
    free(P.fwd_tb.c); free(P.rev_tb.c);
    odata_clear(&P.od);
    return n;
    }
