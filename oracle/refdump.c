/* refdump — golden-vector and table dumper that links against the *reference's own objects*.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/refdump from the reference
 * sources where they lie under /root/reference (nothing of the reference is copied into this repo;
 * this file only *calls* the reference's public API).  It is used
 *   (a) to dump the closed C4 model tables (transition id order, shadows, scopes) that
 *       exonerate_amd/csrc/c4_models.c must reproduce           -> tests/golden/tables_*.json
 *   (b) to dump the scoring data tables (submat, translate, splice PSSM)  -> tests/golden/data.json
 *   (c) to produce golden score / region / operation-list / vulgar / cigar vectors for seeded inputs
 *       through the reference's Optimal_find_score (src/c4/optimal.c:123) and Optimal_find_path
 *       (src/c4/optimal.c:368) with the *interpreted* Viterbi (src/c4/viterbi.c:655)
 *                                                              -> tests/golden/*.golden.jsonl
 * main() comes from the reference's general/argument.c:319; we supply Argument_main().
 */
#undef _XOPEN_SOURCE
#define _GNU_SOURCE 1   /* open_memstream */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "argument.h"
#include "c4.h"
#include "optimal.h"
#include "alignment.h"
#include "modeltype.h"
#include "match.h"
#include "affine.h"
#include "intron.h"
#include "frameshift.h"
#include "splice.h"
#include "translate.h"
#include "submat.h"
#include "viterbi.h"
#include "codegen.h"
#include "sequence.h"
#include "alphabet.h"
#include "est2genome.h"
#include "phase.h"
#include "heuristic.h"
#include "sar.h"
#include "gam.h"
#include "hspset.h"
#include "comparison.h"
#include "sdp.h"

static gint state_index(C4_Model *m, C4_State *s){
    register guint i;
    for(i = 0; i < m->state_list->len; i++)
        if(m->state_list->pdata[i] == s)
            return i;
    return -1;
    }

static gint calc_index(C4_Model *m, C4_Calc *c){
    register guint i;
    if(!c) return -1;
    for(i = 0; i < m->calc_list->len; i++)
        if(m->calc_list->pdata[i] == c)
            return i;
    return -2;
    }

static void dump_model(C4_Model *m, const char *key){
    register guint i, j;
    gchar *escaped_name = g_strescape(m->name, NULL);     /* derived models: Segment("a"->"b"):[..] */
    printf("{\"key\":\"%s\",\"name\":\"%s\",\"start_scope\":%d,\"end_scope\":%d,"
           "\"max_query_advance\":%d,\"max_target_advance\":%d,\"shadow_designations\":%d,\n",
           key, escaped_name, m->start_state->scope, m->end_state->scope,
           m->max_query_advance, m->max_target_advance, m->total_shadow_designations);
    printf(" \"start_state\":%d,\"end_state\":%d,\n", state_index(m, m->start_state->state),
           state_index(m, m->end_state->state));
    printf(" \"states\":[");
    for(i = 0; i < m->state_list->len; i++){
        C4_State *s = m->state_list->pdata[i];
        printf("%s\"%s\"", i?",":"", s->name);
        }
    printf("],\n \"calcs\":[");
    for(i = 0; i < m->calc_list->len; i++){
        C4_Calc *c = m->calc_list->pdata[i];
        printf("%s{\"name\":\"%s\",\"max_score\":%d,\"protect\":%d,\"has_func\":%d}",
               i?",":"", c->name, c->max_score, c->protect, c->calc_func?1:0);
        }
    printf("],\n \"transitions\":[\n");
    for(i = 0; i < m->transition_list->len; i++){
        C4_Transition *t = m->transition_list->pdata[i];
        printf("  %s{\"id\":%d,\"name\":\"%s\",\"in\":%d,\"out\":%d,\"aq\":%d,\"at\":%d,"
               "\"calc\":%d,\"label\":%d,\"dst_shadows\":[",
               i?",":"", t->id, t->name, state_index(m, t->input), state_index(m, t->output),
               t->advance_query, t->advance_target, calc_index(m, t->calc), t->label);
        for(j = 0; j < t->dst_shadow_list->len; j++){
            C4_Shadow *sh = t->dst_shadow_list->pdata[j];
            printf("%s%d", j?",":"", sh->id);
            }
        printf("]}\n");
        }
    printf(" ],\n \"shadows\":[");
    for(i = 0; i < m->shadow_list->len; i++){
        C4_Shadow *sh = m->shadow_list->pdata[i];
        printf("%s{\"name\":\"%s\",\"designation\":%d,\"src_states\":[", i?",":"",
               sh->name, sh->designation);
        for(j = 0; j < sh->src_state_list->len; j++)
            printf("%s%d", j?",":"", state_index(m, sh->src_state_list->pdata[j]));
        printf("],\"dst_transitions\":[");
        for(j = 0; j < sh->dst_transition_list->len; j++){
            C4_Transition *t = sh->dst_transition_list->pdata[j];
            printf("%s%d", j?",":"", t->id);
            }
        printf("]}");
        }
    printf("],\n \"spans\":[");
    for(i = 0; i < m->span_list->len; i++){
        C4_Span *sp = m->span_list->pdata[i];
        printf("%s{\"state\":%d,\"min_q\":%d,\"max_q\":%d,\"min_t\":%d,\"max_t\":%d}", i?",":"",
               state_index(m, sp->span_state), sp->min_query, sp->max_query,
               sp->min_target, sp->max_target);
        }
    printf("],\n \"portals\":[");
    for(i = 0; i < m->portal_list->len; i++){
        C4_Portal *p = m->portal_list->pdata[i];
        printf("%s{\"name\":\"%s\",\"aq\":%d,\"at\":%d,\"calc\":%d}", i?",":"",
               p->name, p->advance_query, p->advance_target, calc_index(m, p->calc));
        }
    printf("]}\n");
    }

static void dump_tables(void){
    struct { const char *key; Model_Type type; Alphabet_Type q, t; } list[] = {
        {"affine:global:protein", Model_Type_AFFINE_GLOBAL, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"affine:bestfit:protein", Model_Type_AFFINE_BESTFIT, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"affine:local:protein", Model_Type_AFFINE_LOCAL, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"affine:overlap:protein", Model_Type_AFFINE_OVERLAP, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"affine:global:dna", Model_Type_AFFINE_GLOBAL, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"affine:bestfit:dna", Model_Type_AFFINE_BESTFIT, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"affine:local:dna", Model_Type_AFFINE_LOCAL, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"affine:overlap:dna", Model_Type_AFFINE_OVERLAP, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"ungapped:dna", Model_Type_UNGAPPED, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"ungapped:protein", Model_Type_UNGAPPED, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"est2genome", Model_Type_EST2GENOME, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"protein2dna", Model_Type_PROTEIN2DNA, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        {"protein2dna:bestfit", Model_Type_PROTEIN2DNA_BESTFIT, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        {"protein2genome", Model_Type_PROTEIN2GENOME, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        {"protein2genome:bestfit", Model_Type_PROTEIN2GENOME_BESTFIT, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        };
    register guint i;
    printf("[\n");
    for(i = 0; i < sizeof(list)/sizeof(list[0]); i++){
        C4_Model *m = Model_Type_get_model(list[i].type, list[i].q, list[i].t);
        if(i) printf(",\n");
        dump_model(m, list[i].key);
        C4_Model_destroy(m);
        }
    printf("]\n");
    }

/* BSDP's derived models (heuristic.c:242-330): for every state a MATCH-labelled transition enters, the start
 * terminal START -> state, the end terminal state -> END, and the joins state -> state'. */
static void dump_derived_one(C4_Model *m, const char *key, C4_State *src, C4_State *dst,
                             C4_Scope start_scope, C4_Scope end_scope, gboolean *first){
    register C4_DerivedModel *dm;
    register guint i;
    gchar *dkey;
    if(!C4_Model_path_is_possible(m, src, dst))
        return;
    dm = C4_DerivedModel_create(m, src, dst, start_scope, NULL, NULL, end_scope, NULL, NULL);
    dkey = g_strdup_printf("%s|%d|%d|%d|%d", key, state_index(m, src), state_index(m, dst),
                           start_scope, end_scope);
    if(!*first) printf(",\n");
    *first = FALSE;
    printf("{\"derived_key\":\"%s\",\"transition_map\":[", dkey);
    for(i = 0; i < dm->derived->transition_list->len; i++)
        printf("%s%d", i?",":"", dm->transition_map[i]->id);
    printf("],\"table\":\n");
    dump_model(dm->derived, dkey);
    printf("}");
    g_free(dkey);
    C4_DerivedModel_destroy(dm);
    }

static void dump_derived(void){
    struct { const char *key; Model_Type type; Alphabet_Type q, t; } list[] = {
        {"affine:local:dna", Model_Type_AFFINE_LOCAL, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"affine:global:dna", Model_Type_AFFINE_GLOBAL, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"affine:local:protein", Model_Type_AFFINE_LOCAL, Alphabet_Type_PROTEIN, Alphabet_Type_PROTEIN},
        {"est2genome", Model_Type_EST2GENOME, Alphabet_Type_DNA, Alphabet_Type_DNA},
        {"protein2dna", Model_Type_PROTEIN2DNA, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        {"protein2genome", Model_Type_PROTEIN2GENOME, Alphabet_Type_PROTEIN, Alphabet_Type_DNA},
        };
    register guint i, j, k;
    gboolean first = TRUE;
    printf("[\n");
    for(i = 0; i < sizeof(list)/sizeof(list[0]); i++){
        C4_Model *m = Model_Type_get_model(list[i].type, list[i].q, list[i].t);
        GPtrArray *match_states = g_ptr_array_new();
        for(j = 0; j < m->transition_list->len; j++){
            C4_Transition *t = m->transition_list->pdata[j];
            gboolean seen = FALSE;
            if(t->label != C4_Label_MATCH)
                continue;
            for(k = 0; k < match_states->len; k++)
                if(match_states->pdata[k] == t->output)
                    seen = TRUE;
            if(!seen)
                g_ptr_array_add(match_states, t->output);
            }
        for(j = 0; j < match_states->len; j++){
            C4_State *s = match_states->pdata[j];
            dump_derived_one(m, list[i].key, m->start_state->state, s, m->start_state->scope,
                             C4_Scope_CORNER, &first);
            dump_derived_one(m, list[i].key, s, m->end_state->state, C4_Scope_CORNER,
                             m->end_state->scope, &first);
            for(k = 0; k < match_states->len; k++)
                dump_derived_one(m, list[i].key, s, match_states->pdata[k], C4_Scope_CORNER,
                                 C4_Scope_CORNER, &first);
            }
        /* span models (heuristic.c:461-472): match state -> span state, span state -> match state */
        for(j = 0; j < m->span_list->len; j++){
            C4_Span *span = m->span_list->pdata[j];
            for(k = 0; k < match_states->len; k++){
                dump_derived_one(m, list[i].key, match_states->pdata[k], span->span_state, C4_Scope_CORNER,
                                 C4_Scope_ANYWHERE, &first);
                dump_derived_one(m, list[i].key, span->span_state, match_states->pdata[k], C4_Scope_ANYWHERE,
                                 C4_Scope_CORNER, &first);
                dump_derived_one(m, list[i].key, match_states->pdata[k], span->span_state, C4_Scope_CORNER,
                                 C4_Scope_CORNER, &first);
                }
            }
        g_ptr_array_free(match_states, TRUE);
        C4_Model_destroy(m);
        }
    printf("]\n");
    }

/* ---- span models: the cell_end_func / cell_start_func seam of the Viterbi (viterbi.c:728-741,793-799) ----- */
/* BSDP runs a span as two DPs that exchange an integration matrix (heuristic.c:385-443, sar.c:898-921).  Here
 * the exchange is the identity: every END cell the src DP reports is offered to the dst DP as the START cell
 * of the same position.  That pins exactly what the Viterbi does with the two callbacks. */
static C4_Score *span_matrix = NULL;      /* [Q+1][T+1][cell_size], score LOW where nothing was reported */
static gint span_q0, span_t0, span_tlen, span_cs;
static C4_Score span_dummy[8];

static void span_report_end(C4_Score *cell, gint cell_size, gint query_pos, gint target_pos, gpointer user_data){
    register gint l;
    register C4_Score *dst = span_matrix + (((query_pos - span_q0) * (span_tlen + 1)) + (target_pos - span_t0)) * span_cs;
    for(l = 0; l < cell_size; l++)
        dst[l] = cell[l];
    return;
    }

static C4_Score *span_init_start(gint query_pos, gint target_pos, gpointer user_data){
    register C4_Score *src = span_matrix + (((query_pos - span_q0) * (span_tlen + 1)) + (target_pos - span_t0)) * span_cs;
    if(src[0] == C4_IMPOSSIBLY_LOW_SCORE)
        return span_dummy;
    return src;
    }

static void run_span(gchar *model_name, gchar *input_path, gint match_state, gint span_state){
    register Model_Type type = Model_Type_from_string(model_name);
    register FILE *fp = fopen(input_path, "r");
    register Alphabet *dna = Alphabet_create(Alphabet_Type_DNA, FALSE);
    register gboolean protein_query = !strncmp(model_name, "protein2", 8);
    register Alphabet *qalpha = protein_query ? Alphabet_create(Alphabet_Type_PROTEIN, FALSE) : dna;
    register C4_Model *model = Model_Type_get_model(type, protein_query ? Alphabet_Type_PROTEIN : Alphabet_Type_DNA,
                                                    Alphabet_Type_DNA);
    register C4_DerivedModel *src_dm = C4_DerivedModel_create(model, model->state_list->pdata[match_state],
            model->state_list->pdata[span_state], C4_Scope_CORNER, NULL, NULL,
            C4_Scope_ANYWHERE, span_report_end, NULL);
    register C4_DerivedModel *dst_dm = C4_DerivedModel_create(model, model->state_list->pdata[span_state],
            model->state_list->pdata[match_state], C4_Scope_ANYWHERE, span_init_start, NULL,
            C4_Scope_CORNER, NULL, NULL);
    register Optimal *src_optimal = Optimal_create(src_dm->derived, NULL, Optimal_Type_SCORE|Optimal_Type_PATH, FALSE);
    register Optimal *dst_optimal = Optimal_create(dst_dm->derived, NULL, Optimal_Type_SCORE|Optimal_Type_PATH, FALSE);
    register gchar *line = g_malloc(1<<20);
    register gint l;
    span_cs = 1 + src_dm->derived->total_shadow_designations;
    span_dummy[0] = C4_IMPOSSIBLY_LOW_SCORE;
    for(l = 1; l < 8; l++)
        span_dummy[l] = 0;
    while(fgets(line, 1<<20, fp)){
        gchar **f;
        Sequence *query, *target;
        gpointer user_data;
        Region *region;
        C4_Score src_score, dst_score;
        Alignment *alignment;
        register gint i, j, first = 1;
        g_strchomp(line);
        if((!line[0]) || (line[0] == '#'))
            continue;
        f = g_strsplit(line, "\t", 3);
        query = Sequence_create(f[0], NULL, f[1], 0, Sequence_Strand_FORWARD, qalpha);
        target = Sequence_create("tg", NULL, f[2], 0, Sequence_Strand_FORWARD, dna);
        user_data = Model_Type_create_data(type, query, target);
        region = Region_create(0, 0, query->len, target->len);
        span_q0 = 0; span_t0 = 0; span_tlen = target->len;
        span_matrix = g_new(C4_Score, (query->len + 1) * (target->len + 1) * span_cs);
        for(i = 0; i < (query->len + 1) * (target->len + 1); i++){
            span_matrix[i*span_cs] = C4_IMPOSSIBLY_LOW_SCORE;
            for(l = 1; l < span_cs; l++)
                span_matrix[i*span_cs+l] = 0;
            }
        src_score = Optimal_find_score(src_optimal, region, user_data, NULL);
        printf("{\"id\":\"%s\",\"qlen\":%d,\"tlen\":%d,\"cell_size\":%d,\"src_score\":%d,\"end_cells\":[",
               f[0], query->len, target->len, span_cs, src_score);
        for(i = 0; i <= query->len; i++)
            for(j = 0; j <= target->len; j++){
                C4_Score *c = span_matrix + ((i * (target->len + 1)) + j) * span_cs;
                if(c[0] == C4_IMPOSSIBLY_LOW_SCORE)
                    continue;
                printf("%s[%d,%d", first?"":",", i, j);
                for(l = 0; l < span_cs; l++)
                    printf(",%d", c[l]);
                printf("]");
                first = 0;
                }
        dst_score = Optimal_find_score(dst_optimal, region, user_data, NULL);
        printf("],\"dst_score\":%d", dst_score);
        alignment = Optimal_find_path(dst_optimal, region, user_data, C4_IMPOSSIBLY_LOW_SCORE, NULL);
        if(alignment){
            register guint k;
            printf(",\"path_score\":%d,\"region\":[%d,%d,%d,%d],\"ops\":[", alignment->score,
                   alignment->region->query_start, alignment->region->target_start,
                   alignment->region->query_length, alignment->region->target_length);
            for(k = 0; k < alignment->operation_list->len; k++){
                AlignmentOperation *ao = alignment->operation_list->pdata[k];
                printf("%s[%d,%d]", k?",":"", ao->transition->id, ao->length);
                }
            printf("]");
            Alignment_destroy(alignment);
            }
        printf("}\n");
        fflush(stdout);
        g_free(span_matrix);
        Model_Type_destroy_data(type, user_data);
        Region_destroy(region);
        Sequence_destroy(query);
        Sequence_destroy(target);
        g_strfreev(f);
        }
    fclose(fp);
    return;
    }

static void dump_submat(const char *name, Submat *s, gboolean last){
    register gint i, j;
    printf(" \"%s\":[", name);
    for(i = 0; i < SUBMAT_ALPHABETSIZE; i++){
        printf("%s[", i?",":"");
        for(j = 0; j < SUBMAT_ALPHABETSIZE; j++)
            printf("%s%d", j?",":"", s->matrix[i][j]);
        printf("]");
        }
    printf("]%s\n", last?"":",");
    }

static void dump_splice(const char *name, SplicePredictor *sp){
    register gint i, j;
    union { gfloat f; guint32 u; } cv;
    printf(" \"%s\":{\"model_length\":%d,\"splice_after\":%d,\"max_score_bits\":", name,
           sp->model_length, sp->model_splice_after);
    cv.f = SplicePredictor_get_max_score(sp);
    printf("%u,\"index_ACGTN\":[%d,%d,%d,%d,%d],\"data_bits\":[", cv.u,
           sp->index['A'], sp->index['C'], sp->index['G'], sp->index['T'], sp->index['N']);
    for(i = 0; i < sp->model_length; i++){
        printf("%s[", i?",":"");
        for(j = 0; j < 5; j++){
            cv.f = sp->model_data[i][j];
            printf("%s%u", j?",":"", cv.u);
            }
        printf("]");
        }
    printf("]},\n");
    }

static void dump_data(void){
    register Match_ArgumentSet *mas = Match_ArgumentSet_create(NULL);
    register Intron_ArgumentSet *ias = Intron_ArgumentSet_create(NULL);
    register Affine_ArgumentSet *aas = Affine_ArgumentSet_create(NULL);
    register Frameshift_ArgumentSet *fas = Frameshift_ArgumentSet_create(NULL);
    register gint i;
    printf("{\n \"submat_index\":[");
    for(i = 0; i < 256; i++)
        printf("%s%d", i?",":"", mas->dna_submat->index[i]);
    printf("],\n");
    dump_submat("nucleic", mas->dna_submat, FALSE);
    dump_submat("blosum62", mas->protein_submat, FALSE);
    printf(" \"translate_nt2d\":[");
    for(i = 0; i < 256; i++)
        printf("%s%d", i?",":"", mas->translate->nt2d[i]);
    printf("],\n \"translate_trans\":[");
    for(i = 0; i < Translate_TRANSLATION_SIZE; i++)
        printf("%s%d", i?",":"", mas->translate->trans[i]);
    printf("],\n \"translate_aa\":\"");
    for(i = 0; i < Translate_AA_SET_SIZE && mas->translate->aa[i]; i++)
        printf("%c", mas->translate->aa[i]);
    printf("\",\n");
    dump_splice("ss5_forward", ias->sps->ss5_forward);
    dump_splice("ss5_reverse", ias->sps->ss5_reverse);
    dump_splice("ss3_forward", ias->sps->ss3_forward);
    dump_splice("ss3_reverse", ias->sps->ss3_reverse);
    printf(" \"gap_open\":%d,\"gap_extend\":%d,\"codon_gap_open\":%d,\"codon_gap_extend\":%d,\n",
           aas->gap_open, aas->gap_extend, aas->codon_gap_open, aas->codon_gap_extend);
    printf(" \"min_intron\":%d,\"max_intron\":%d,\"intron_open_penalty\":%d,\"frameshift_penalty\":%d\n}\n",
           ias->min_intron, ias->max_intron, ias->intron_open_penalty, fas->frameshift_penalty);
    }

/**/

static gchar *capture_display(Alignment *alignment, Sequence *query, Sequence *target,
                              void (*func)(Alignment*, Sequence*, Sequence*, FILE*)){
    char *buf = NULL;
    size_t len = 0;
    FILE *fp = open_memstream(&buf, &len);
    register gchar *result;
    func(alignment, query, target, fp);
    fclose(fp);
    while(len && (buf[len-1] == '\n'))
        buf[--len] = '\0';
    result = g_strdup(buf);
    free(buf);
    return result;
    }

static void dump_splice_array(const char *name, SplicePredictor *sp, Sequence *s){
    register gint *pred = g_new(gint, s->len);
    register gchar *seq = Sequence_get_str(s);
    register guint i;
    SplicePredictor_predict_array_int(sp, seq, s->len, 0, s->len, pred);
    printf(",\"%s\":[", name);
    for(i = 0; i < s->len; i++)
        printf("%s%d", i?",":"", pred[i]);
    printf("]");
    g_free(seq);
    g_free(pred);
    }

/* SubOpt points of the whole pair, sorted by target then query (the order SubOpt_Index uses) */
static gboolean collect_point(gint query_pos, gint target_pos, gint path_id, gpointer user_data){
    register GArray *a = user_data;
    gint v[2];
    v[0] = target_pos; v[1] = query_pos;
    g_array_append_val(a, v[0]);
    g_array_append_val(a, v[1]);
    return FALSE;
    }

static int compare_point(const void *a, const void *b){
    const gint *x = a, *y = b;
    if(x[0] != y[0]) return x[0] - y[0];
    return x[1] - y[1];
    }

static void dump_subopt_points(SubOpt *subopt, Sequence *query, Sequence *target){
    register GArray *a = g_array_new(FALSE, FALSE, sizeof(gint));
    register guint i;
    Region all;
    all.query_start = 0; all.target_start = 0;
    all.query_length = query->len + 1; all.target_length = target->len + 1;
    SubOpt_find(subopt, &all, collect_point, a);
    qsort(a->data, a->len/2, 2*sizeof(gint), compare_point);
    printf(",\"points\":[");
    for(i = 0; i < a->len; i += 2)
        printf("%s[%d,%d]", i?",":"", g_array_index(a, gint, i+1), g_array_index(a, gint, i));
    printf("]");
    g_array_free(a, TRUE);
    return;
    }

static void dump_alignment_fields(Alignment *alignment, Sequence *query, Sequence *target){
    register guint i;
    gchar *s;
    printf("\"path_score\":%d,\"region\":[%d,%d,%d,%d],\"ops\":[",
           alignment->score, alignment->region->query_start,
           alignment->region->target_start, alignment->region->query_length,
           alignment->region->target_length);
    for(i = 0; i < alignment->operation_list->len; i++){
        AlignmentOperation *ao = alignment->operation_list->pdata[i];
        printf("%s[%d,%d]", i?",":"", ao->transition->id, ao->length);
        }
    printf("]");
    s = capture_display(alignment, query, target, Alignment_display_vulgar);
    printf(",\"vulgar\":\"%s\"", s); g_free(s);
    return;
    }

/* the loop of GAM_Result_exhaustive_create (gam.c:1139-1180): next best path with everything found so
 * far blocked, until the score drops below the threshold */
static void run_subopt(Optimal *optimal, Region *region, gpointer user_data,
                       Sequence *query, Sequence *target, gint subopt_max, gint threshold){
    register SubOpt *subopt = SubOpt_create(query->len, target->len);
    register Alignment *alignment;
    register gint k;
    printf(",\"threshold\":%d,\"subopt\":[", threshold);
    for(k = 0; k < subopt_max; k++){
        alignment = Optimal_find_path(optimal, region, user_data, threshold, subopt);
        if(!alignment)
            break;
        printf("%s{", k?",":"");
        dump_alignment_fields(alignment, query, target);
        SubOpt_add_alignment(subopt, alignment);
        dump_subopt_points(subopt, query, target);
        printf("}");
        Alignment_destroy(alignment);
        }
    printf("]");
    SubOpt_destroy(subopt);
    return;
    }

static void run_golden(gchar *model_name, gchar *input_path, gboolean with_splice,
                       gboolean revcomp_target, gint subopt_max, gint subopt_threshold, gchar *derived){
    register Model_Type type;
    register FILE *fp = fopen(input_path, "r");
    register Alphabet *dna = Alphabet_create(Alphabet_Type_DNA, FALSE),
                      *protein = Alphabet_create(Alphabet_Type_PROTEIN, FALSE);
    register Alphabet *qa, *ta;
    register C4_Model *model;
    register Optimal *optimal;
    register gboolean query_is_protein = FALSE, target_is_protein = FALSE;
    register size_t cap = 1<<26;
    register gchar *line = g_malloc(cap);
    if(!fp)
        g_error("cannot open [%s]", input_path);
    if(strstr(model_name, ":protein")){ /* affine:local:protein etc (refdump-only suffix) */
        query_is_protein = target_is_protein = TRUE;
        model_name[strlen(model_name)-strlen(":protein")] = '\0';
        }
    type = Model_Type_from_string(model_name);
    switch(type){
        case Model_Type_PROTEIN2DNA: case Model_Type_PROTEIN2DNA_BESTFIT:
        case Model_Type_PROTEIN2GENOME: case Model_Type_PROTEIN2GENOME_BESTFIT:
            query_is_protein = TRUE;
            break;
        default:
            break;
        }
    qa = query_is_protein?protein:dna;
    ta = target_is_protein?protein:dna;
    model = Model_Type_get_model(type, qa->type, ta->type);
    if(derived && strcmp(derived, "none")){
        /* "src,dst,start_scope,end_scope": one of BSDP's derived models (heuristic.c:242-330) */
        gchar **d = g_strsplit(derived, ",", 4);
        register C4_DerivedModel *dm = C4_DerivedModel_create(model,
                model->state_list->pdata[atoi(d[0])], model->state_list->pdata[atoi(d[1])],
                atoi(d[2]), NULL, NULL, atoi(d[3]), NULL, NULL);
        model = C4_Model_share(dm->derived);      /* the original stays referenced by dm (leaked: one-shot tool) */
        g_strfreev(d);
        }
    optimal = Optimal_create(model, NULL,
                  Optimal_Type_SCORE|Optimal_Type_PATH|Optimal_Type_REDUCED_SPACE, FALSE);
    while(fgets(line, cap, fp)){
        gchar **f;
        Sequence *query, *target, *tfwd;
        gpointer user_data;
        Region *region;
        C4_Score score;
        Alignment *alignment;
        g_strchomp(line);
        if((!line[0]) || (line[0] == '#'))
            continue;
        f = g_strsplit(line, "\t", 4);                 /* id, query, target[, "cds_start:cds_length"] */
        g_assert(f[0] && f[1] && f[2]);
        query = Sequence_create(f[0], NULL, f[1], 0, Sequence_Strand_FORWARD, qa);
        if(f[3] && strchr(f[3], ':')){
            /* --annotation: what Sequence_create attaches to a sequence whose id has an entry in the annotation file
             * (sequence.c:51-88,176-178: a pointer into the file's tree, not owned by the sequence; here a record of its own,
             * never freed: a dump tool) */
            query->annotation = g_new0(Sequence_Annotation, 1);
            query->annotation->id = g_strdup(f[0]);
            query->annotation->strand = Sequence_Strand_FORWARD;
            query->annotation->cds_start = atoi(f[3]);
            query->annotation->cds_length = atoi(strchr(f[3], ':') + 1);
            }
        tfwd = Sequence_create("tg", NULL, f[2], 0, Sequence_Strand_FORWARD, ta);
        if(revcomp_target){
            target = Sequence_revcomp(tfwd);
        } else {
            target = Sequence_share(tfwd);
            }
        user_data = Model_Type_create_data(type, query, target);
        region = Region_create(0, 0, query->len, target->len);
        score = Optimal_find_score(optimal, region, user_data, NULL);
        {
            gchar *escaped_name = g_strescape(model->name, NULL);
            printf("{\"id\":\"%s\",\"model\":\"%s\",\"qlen\":%d,\"tlen\":%d,\"score\":%d",
                   f[0], escaped_name, query->len, target->len, score);
            g_free(escaped_name);
        }
        alignment = Optimal_find_path(optimal, region, user_data,
                                      C4_IMPOSSIBLY_LOW_SCORE, NULL);
        if(alignment){
            register guint i;
            gchar *s;
            printf(",\"path_score\":%d,\"region\":[%d,%d,%d,%d],\"ops\":[",
                   alignment->score, alignment->region->query_start,
                   alignment->region->target_start, alignment->region->query_length,
                   alignment->region->target_length);
            for(i = 0; i < alignment->operation_list->len; i++){
                AlignmentOperation *ao = alignment->operation_list->pdata[i];
                printf("%s[%d,%d]", i?",":"", ao->transition->id, ao->length);
                }
            printf("]");
            s = capture_display(alignment, query, target, Alignment_display_sugar);
            printf(",\"sugar\":\"%s\"", s); g_free(s);
            s = capture_display(alignment, query, target, Alignment_display_cigar);
            printf(",\"cigar\":\"%s\"", s); g_free(s);
            s = capture_display(alignment, query, target, Alignment_display_vulgar);
            printf(",\"vulgar\":\"%s\"", s); g_free(s);
            Alignment_destroy(alignment);
            }
        if(subopt_max > 0)
            run_subopt(optimal, region, user_data, query, target, subopt_max, subopt_threshold);
        if(with_splice){
            register Intron_ArgumentSet *ias = Intron_ArgumentSet_create(NULL);
            dump_splice_array("ss5_forward", ias->sps->ss5_forward, target);
            dump_splice_array("ss3_forward", ias->sps->ss3_forward, target);
            dump_splice_array("ss5_reverse", ias->sps->ss5_reverse, target);
            dump_splice_array("ss3_reverse", ias->sps->ss3_reverse, target);
            }
        printf("}\n");
        fflush(stdout);
        Model_Type_destroy_data(type, user_data);
        Region_destroy(region);
        Sequence_destroy(query);
        Sequence_destroy(target);
        Sequence_destroy(tfwd);
        g_strfreev(f);
        }
    fclose(fp);
    g_free(line);
    Optimal_destroy(optimal);
    C4_Model_destroy(model);
    Alphabet_destroy(dna);
    Alphabet_destroy(protein);
    return;
    }

/* ---- HSP seeding (src/comparison/hspset.c:933): per seed the HSP a fresh HSPset grows from it (no horizon in the way),
 * and the HSP list of ONE HSPset fed all seeds in order (horizon filter, threshold, store order, cobs) -------------- */
static void run_hsp(gchar *match_name, gchar *input_path){
    register FILE *fp = fopen(input_path, "r");
    register Match_Type type = (!strcmp(match_name, "protein2dna")) ? Match_Type_PROTEIN2DNA
                             : (!strcmp(match_name, "protein2protein")) ? Match_Type_PROTEIN2PROTEIN : Match_Type_DNA2DNA;
    register Match *match = Match_find(type);
    register HSP_Param *hsp_param = HSP_Param_create(match, TRUE);
    gchar *line = g_malloc(1<<22);
    if(!fp)
        g_error("cannot open [%s]", input_path);
    printf("{\"params\":{\"match\":\"%s\",\"seedlen\":%d,\"wordlen\":%d,\"dropoff\":%d,\"threshold\":%d,"
           "\"query_advance\":%d,\"target_advance\":%d,\"seed_repeat\":%d}}\n", match_name, hsp_param->seedlen,
           hsp_param->wordlen, hsp_param->dropoff, hsp_param->threshold, match->query->advance, match->target->advance,
           hsp_param->seed_repeat);
    while(fgets(line, 1<<22, fp)){
        gchar **f, **seeds;
        Sequence *query, *target;
        register gint k, first = 1;
        register HSPset *all;
        g_strchomp(line);
        if((!line[0]) || (line[0] == '#'))
            continue;
        f = g_strsplit(line, "\t", 4);                   /* id, query, target, "q:t,q:t,..." */
        query = Sequence_create(f[0], NULL, f[1], 0, Sequence_Strand_UNKNOWN, NULL);
        target = Sequence_create("tg", NULL, f[2], 0, Sequence_Strand_UNKNOWN, NULL);
        seeds = g_strsplit(f[3], ",", -1);
        all = HSPset_create(query, target, hsp_param);
        printf("{\"id\":\"%s\",\"single\":[", f[0]);
        for(k = 0; seeds[k] && seeds[k][0]; k++){
            register gint q = atoi(seeds[k]), t = atoi(strchr(seeds[k], ':') + 1);
            register HSPset *one = HSPset_create(query, target, hsp_param);
            HSPset_seed_hsp(one, q, t);
            HSPset_finalise(one);
            if(one->hsp_list->len){
                register HSP *h = one->hsp_list->pdata[0];
                printf("%s[%d,%d,%d,%d,%d]", first?"":",", h->query_start, h->target_start, h->length, h->score, h->cobs);
            } else {
                printf("%snull", first?"":",");
                }
            first = 0;
            HSPset_destroy(one);
            HSPset_seed_hsp(all, q, t);
            }
        HSPset_finalise(all);
        printf("],\"set\":[");
        for(k = 0; k < (gint)all->hsp_list->len; k++){
            register HSP *h = all->hsp_list->pdata[k];
            printf("%s[%d,%d,%d,%d,%d]", k?",":"", h->query_start, h->target_start, h->length, h->score, h->cobs);
            }
        printf("]}\n");
        HSPset_destroy(all);
        Sequence_destroy(query);
        Sequence_destroy(target);
        g_strfreev(f);
        g_strfreev(seeds);
        }
    fclose(fp);
    g_free(line);
    return;
    }

/* ---- SDP (src/sdp/sdp.c:743 SDP_Pair_next_path over src/sdp/scheduler.c:1445 Scheduler_Pair_calculate, interpreted
 * Scheduler_Cell_process scheduler.c:859): the loop of GAM_Result_SDP_create (gam.c:852-890) on a Comparison whose
 * HSPset was grown from the given word hits; dumps the HSPs (the oracle's input) and every alignment ------------------ */
static void run_sdp(gchar *model_name, gchar *input_path, gint subopt_max, gint threshold){
    register Model_Type type;
    register FILE *fp = fopen(input_path, "r");
    register Alphabet *dna = Alphabet_create(Alphabet_Type_DNA, FALSE),
                      *protein = Alphabet_create(Alphabet_Type_PROTEIN, FALSE);
    register Alphabet *qa, *ta;
    register C4_Model *model;
    register gboolean query_is_protein = FALSE, target_is_protein = FALSE;
    register Match *match;
    register HSP_Param *hsp_param;
    register Comparison_Param *cparam;
    register SDP *sdp;
    register SDP_ArgumentSet *sas = SDP_ArgumentSet_create(NULL);
    gchar *line = g_malloc(1<<24);
    if(!fp)
        g_error("cannot open [%s]", input_path);
    if(strstr(model_name, ":protein")){
        query_is_protein = target_is_protein = TRUE;
        model_name[strlen(model_name)-strlen(":protein")] = '\0';
        }
    type = Model_Type_from_string(model_name);
    switch(type){
        case Model_Type_PROTEIN2DNA: case Model_Type_PROTEIN2DNA_BESTFIT:
        case Model_Type_PROTEIN2GENOME: case Model_Type_PROTEIN2GENOME_BESTFIT:
            query_is_protein = TRUE;
            break;
        default:
            break;
        }
    qa = query_is_protein?protein:dna;
    ta = target_is_protein?protein:dna;
    match = Match_find(query_is_protein ? (target_is_protein ? Match_Type_PROTEIN2PROTEIN : Match_Type_PROTEIN2DNA)
                                        : Match_Type_DNA2DNA);
    hsp_param = HSP_Param_create(match, TRUE);
    cparam = Comparison_Param_create(qa->type, ta->type,
                 (!query_is_protein) ? hsp_param : NULL,
                 (query_is_protein && target_is_protein) ? hsp_param : NULL,
                 (query_is_protein && !target_is_protein) ? hsp_param : NULL);
    model = Model_Type_get_model(type, qa->type, ta->type);
    sdp = SDP_create(model);
    printf("{\"params\":{\"model\":\"%s\",\"use_boundary\":%d,\"dropoff\":%d,\"singlepass\":%d,\"threshold\":%d,"
           "\"hsp_threshold\":%d,\"n_spans\":%d,\"spans\":[", model->name, sdp->use_boundary, sas->dropoff,
           sas->single_pass_subopt, threshold, hsp_param->threshold, model->span_list->len);
    {
        register guint k;
        for(k = 0; k < model->span_list->len; k++){
            register C4_Span *span = model->span_list->pdata[k];
            printf("%s[%d,%d,%d,%d,%d,%d,%d]", k?",":"", span->span_state->id, span->min_query, span->max_query,
                   span->min_target, span->max_target, span->query_loop ? span->query_loop->id : -1,
                   span->target_loop ? span->target_loop->id : -1);
            }
    }
    printf("]}}\n");
    while(fgets(line, 1<<24, fp)){
        gchar **f, **seeds;
        Sequence *query, *target;
        Comparison *comparison;
        HSPset *hspset;
        gpointer user_data;
        register gint k;
        g_strchomp(line);
        if((!line[0]) || (line[0] == '#'))
            continue;
        f = g_strsplit(line, "\t", 4);                  /* id, query, target, "q:t,q:t,..." word hits */
        query = Sequence_create(f[0], NULL, f[1], 0, Sequence_Strand_FORWARD, qa);
        target = Sequence_create("tg", NULL, f[2], 0, Sequence_Strand_FORWARD, ta);
        seeds = g_strsplit(f[3], ",", -1);
        comparison = Comparison_create(cparam, query, target);
        hspset = comparison->dna_hspset ? comparison->dna_hspset
               : comparison->protein_hspset ? comparison->protein_hspset : comparison->codon_hspset;
        for(k = 0; seeds[k] && seeds[k][0]; k++)
            HSPset_seed_hsp(hspset, atoi(seeds[k]), atoi(strchr(seeds[k], ':') + 1));
        Comparison_finalise(comparison);
        printf("{\"id\":\"%s\",\"qlen\":%d,\"tlen\":%d,\"hsps\":[", f[0], query->len, target->len);
        for(k = 0; k < (gint)hspset->hsp_list->len; k++){
            register HSP *h = hspset->hsp_list->pdata[k];
            printf("%s[%d,%d,%d,%d,%d]", k?",":"", h->query_start, h->target_start, h->length, h->score, h->cobs);
            }
        printf("],\"alignments\":[");
        if(Comparison_has_hsps(comparison)){
            register SubOpt *subopt = SubOpt_create(query->len, target->len);
            register SDP_Pair *sdp_pair;
            register Alignment *alignment;
            user_data = Model_Type_create_data(type, query, target);
            sdp_pair = SDP_Pair_create(sdp, subopt, comparison, user_data);
            for(k = 0; k < subopt_max; k++){
                alignment = SDP_Pair_next_path(sdp_pair, threshold);
                if(!alignment)
                    break;
                printf("%s{", k?",":"");
                dump_alignment_fields(alignment, query, target);
                printf("}");
                SubOpt_add_alignment(subopt, alignment);
                Alignment_destroy(alignment);
                }
            SDP_Pair_destroy(sdp_pair);
            SubOpt_destroy(subopt);
            Model_Type_destroy_data(type, user_data);
            }
        printf("]}\n");
        fflush(stdout);
        Comparison_destroy(comparison);
        Sequence_destroy(query);
        Sequence_destroy(target);
        g_strfreev(f);
        g_strfreev(seeds);
        }
    fclose(fp);
    g_free(line);
    return;
    }

/* ---- the seeder's automaton walk (src/comparison/seeder.c:852-915 -> fsm.c:186-198 / seeder.c:698-720 -> :649-695):
 * per record a fresh Seeder over the record's queries; dumps (1) the words the queries put into the automaton, read off the
 * trie before it is compiled (or off the VFSM's leaf table) with each word's OWN seeds in list order and its neighbour words
 * in list order -- the structure Seeder_FSM_traverse_func walks --, (2) the target as the columns of the automaton after the
 * reference's own masking, (3) every (query, query position, target position) the reference's walk hands to HSPset_seed_hsp,
 * in order (the call is intercepted at link time: -Wl,--wrap=HSPset_seed_hsp in oracle/Makefile).  Untranslated matches. */
#include "seeder.h"
#include "fsm.h"
#include "vfsm.h"
typedef struct { Sequence *query; gint query_pos, target_pos; } SeedCall;
static GArray *seed_calls = NULL;
extern void __real_HSPset_seed_hsp(HSPset *hsp_set, guint query_start, guint target_start);
void __wrap_HSPset_seed_hsp(HSPset *hsp_set, guint query_start, guint target_start){
    if(seed_calls){
        SeedCall c;
        c.query = hsp_set->query; c.query_pos = query_start; c.target_pos = target_start;
        g_array_append_val(seed_calls, c);
        }
    __real_HSPset_seed_hsp(hsp_set, query_start, target_start);
    return;
    }
static void seeds_report(Comparison *comparison, gpointer user_data){ return; }

typedef struct { guint64 code; Seeder_WordInfo *info; } SeedWord;
typedef struct { FSM_Node *node; guint64 code; } SeedTrieItem;

static void run_seeds(gchar *match_name, gchar *input_path){
    register FILE *fp = fopen(input_path, "r");
    register gboolean is_protein = !strcmp(match_name, "protein2protein");
    register Alphabet *alpha = Alphabet_create(is_protein ? Alphabet_Type_PROTEIN : Alphabet_Type_DNA, FALSE);
    register Match *match = Match_find(is_protein ? Match_Type_PROTEIN2PROTEIN : Match_Type_DNA2DNA);
    register HSP_Param *hsp_param = HSP_Param_create(match, TRUE);
    register Comparison_Param *cparam = Comparison_Param_create(alpha->type, alpha->type, is_protein ? NULL : hsp_param,
                                                                is_protein ? hsp_param : NULL, NULL);
    gchar *line = g_malloc(1<<24);
    if(!fp)
        g_error("cannot open [%s]", input_path);
    while(fgets(line, 1<<24, fp)){
        gchar **f, **qs;
        register Seeder *seeder;
        register GPtrArray *queries = g_ptr_array_new();
        register GArray *words = g_array_new(FALSE, FALSE, sizeof(SeedWord));
        register GHashTable *word_of = g_hash_table_new(g_direct_hash, g_direct_equal);
        register Sequence *target, *masked;
        register gchar *seq;
        register gint i, k, width, wordlen;
        register guint w;
        guchar column[256];
        g_strchomp(line);
        if((!line[0]) || (line[0] == '#'))
            continue;
        f = g_strsplit(line, "\t", 3);                  /* id, "q0,q1,...", target */
        qs = g_strsplit(f[1], ",", -1);
        seeder = Seeder_create(0, cparam, 0, seeds_report, NULL);
        for(k = 0; qs[k]; k++){
            gchar *name = g_strdup_printf("q%d", k);
            register Sequence *q = Sequence_create(name, NULL, qs[k], 0, Sequence_Strand_FORWARD, alpha);
            g_ptr_array_add(queries, q);
            Seeder_add_query(seeder, q);
            g_free(name);
            }
        wordlen = seeder->any_hsp_param->wordlen;
        memset(column, 0, sizeof(column));
        if(seeder->seeder_fsm){                          /* the trie before FSM_compile: `next` is NULL where there is no child */
            register FSM *fsm = seeder->seeder_fsm->fsm;
            register GArray *level = g_array_new(FALSE, FALSE, sizeof(SeedTrieItem)), *next_level;
            register gint depth, c;
            SeedTrieItem it;
            width = fsm->width;
            for(i = 1; i < 256; i++)
                column[i] = fsm->traversal_filter[i];
            it.node = fsm->root; it.code = 0;
            g_array_append_val(level, it);
            for(depth = 0; depth + 1 < wordlen; depth++){
                next_level = g_array_new(FALSE, FALSE, sizeof(SeedTrieItem));
                for(w = 0; w < level->len; w++){
                    register SeedTrieItem *p = &g_array_index(level, SeedTrieItem, w);
                    for(c = 1; c < fsm->width; c++)
                        if(p->node[c].next){
                            it.node = p->node[c].next; it.code = p->code * fsm->width + c;
                            g_array_append_val(next_level, it);
                            }
                    }
                g_array_free(level, TRUE);
                level = next_level;
                }
            for(w = 0; w < level->len; w++){
                register SeedTrieItem *p = &g_array_index(level, SeedTrieItem, w);
                for(c = 1; c < fsm->width; c++)
                    if(p->node[c].data){
                        SeedWord sw;
                        sw.code = p->code * fsm->width + c; sw.info = p->node[c].data;
                        g_hash_table_insert(word_of, sw.info, GINT_TO_POINTER(words->len + 1));
                        g_array_append_val(words, sw);
                        }
                }
            g_array_free(level, TRUE);
        } else {
            register VFSM *vfsm = seeder->seeder_vfsm->vfsm;
            register VFSM_Int leaf;
            register gchar *word = g_new0(gchar, vfsm->depth + 1);
            width = vfsm->alphabet_size + 1;
            for(i = 1; i < 256; i++)
                column[i] = (guchar)vfsm->index[toupper(i)];
            for(leaf = 0; leaf < vfsm->lrw; leaf++){
                register Seeder_WordInfo *info = seeder->seeder_vfsm->leaf[leaf];
                SeedWord sw;
                if(!info)
                    continue;
                VFSM_state2word(vfsm, VFSM_leaf2state(vfsm, leaf), word);
                sw.code = 0;
                for(i = 0; i < wordlen; i++)
                    sw.code = sw.code * width + (guchar)vfsm->index[(guchar)word[i]];
                sw.info = info;
                g_hash_table_insert(word_of, sw.info, GINT_TO_POINTER(words->len + 1));
                g_array_append_val(words, sw);
                }
            g_free(word);
            }
        printf("{\"id\":\"%s\",\"match\":\"%s\",\"automaton\":\"%s\",\"width\":%d,\"wordlen\":%d,\"tpos_modifier\":%d,\"n_queries\":%d,\"words\":[",
               f[0], match_name, seeder->seeder_fsm ? "fsm" : "vfsm", width, wordlen, wordlen - 1, queries->len);
        for(w = 0; w < words->len; w++){
            register SeedWord *sw = &g_array_index(words, SeedWord, w);
            register Seeder_Seed *seed;
            register Seeder_Neighbour *nb;
            register gboolean first = TRUE;
            printf("%s[%lu,[", w ? "," : "", (gulong)sw->code);
            for(seed = sw->info->seed_list; seed; seed = seed->next){
                register gint qi = -1;
                for(k = 0; k < (gint)queries->len; k++)
                    if(queries->pdata[k] == seed->context->query_info->query)
                        qi = k;
                printf("%s[%d,%d]", first ? "" : ",", qi, seed->query_pos);
                first = FALSE;
                }
            printf("],[");
            first = TRUE;
            for(nb = sw->info->neighbour_list; nb; nb = nb->next){
                printf("%s%d", first ? "" : ",", GPOINTER_TO_INT(g_hash_table_lookup(word_of, nb->word_info)) - 1);
                first = FALSE;
                }
            printf("]]");
            }
        /* the string the reference walks (Seeder_add_target: Sequence_mask, Sequence_get_str), as automaton columns */
        target = Sequence_create("tg", NULL, f[2], 0, Sequence_Strand_FORWARD, alpha);
        masked = Sequence_mask(target);
        seq = Sequence_get_str(masked);
        Sequence_destroy(masked);
        printf("],\"symbols\":[");
        for(i = 0; seq[i]; i++)
            printf("%s%d", i ? "," : "", column[(guchar)seq[i]]);
        g_free(seq);
        seed_calls = g_array_new(FALSE, FALSE, sizeof(SeedCall));
        Seeder_add_target(seeder, target);
        printf("],\"expected\":[");
        for(w = 0; w < seed_calls->len; w++){
            register SeedCall *c = &g_array_index(seed_calls, SeedCall, w);
            register gint qi = -1;
            for(k = 0; k < (gint)queries->len; k++)
                if(queries->pdata[k] == c->query)
                    qi = k;
            printf("%s[%d,%d,%d]", w ? "," : "", qi, c->query_pos, c->target_pos);
            }
        printf("]}\n");
        g_array_free(seed_calls, TRUE);
        seed_calls = NULL;
        g_array_free(words, TRUE);
        g_hash_table_destroy(word_of);
        Sequence_destroy(target);
        /* (the seeder and its queries are left to the end of the process: Seeder_destroy frees the query infos it shares) */
        g_strfreev(f); g_strfreev(qs);
        }
    fclose(fp);
    g_free(line);
    return;
    }

int Argument_main(Argument *arg){
    register ArgumentSet *as = ArgumentSet_create("refdump options");
    gchar *cmd, *model_name, *input_path;
    gboolean with_splice, revcomp_target;
    gint subopt_max, subopt_threshold;
    gchar *derived;
    ArgumentSet_add_option(as, '\0', "cmd", "name", "tables|data|golden", "tables",
                           Argument_parse_string, &cmd);
    ArgumentSet_add_option(as, 'm', "model", "name", "model name", "affine:local",
                           Argument_parse_string, &model_name);
    ArgumentSet_add_option(as, '\0', "input", "path", "tsv of id,query,target", "none",
                           Argument_parse_string, &input_path);
    ArgumentSet_add_option(as, '\0', "withsplice", NULL, "dump splice arrays", "FALSE",
                           Argument_parse_boolean, &with_splice);
    ArgumentSet_add_option(as, '\0', "revcomptarget", NULL, "align to revcomp of target", "FALSE",
                           Argument_parse_boolean, &revcomp_target);
    ArgumentSet_add_option(as, '\0', "suboptmax", "n", "also dump up to n successive sub-optimal paths", "0",
                           Argument_parse_int, &subopt_max);
    ArgumentSet_add_option(as, '\0', "derived", "src,dst,ss,es", "run the golden set on a derived model", "none",
                           Argument_parse_string, &derived);
    ArgumentSet_add_option(as, '\0', "suboptthreshold", "score", "threshold of the sub-optimal loop", "30",
                           Argument_parse_int, &subopt_threshold);
    Argument_absorb_ArgumentSet(arg, as);
    Translate_ArgumentSet_create(arg);
    Viterbi_ArgumentSet_create(arg);
    Codegen_ArgumentSet_create(arg);
    Sequence_ArgumentSet_create(arg);
    Match_ArgumentSet_create(arg);
    Affine_ArgumentSet_create(arg);
    Intron_ArgumentSet_create(arg);
    Frameshift_ArgumentSet_create(arg);
    Alphabet_ArgumentSet_create(arg);
    Alignment_ArgumentSet_create(arg);
    Splice_ArgumentSet_create(arg);
    HSPset_ArgumentSet_create(arg);
    SDP_ArgumentSet_create(arg);
    Seeder_ArgumentSet_create(arg);
    Argument_process(arg, "refdump", "reference table/golden dumper", "");
    if(!strcmp(cmd, "tables"))
        dump_tables();
    else if(!strcmp(cmd, "data"))
        dump_data();
    else if(!strcmp(cmd, "derived"))
        dump_derived();
    else if(!strcmp(cmd, "span")){
        gchar **d = g_strsplit(derived, ",", 2);   /* --derived "<match state>,<span state>" */
        run_span(g_strdup(model_name), input_path, atoi(d[0]), atoi(d[1]));
        }
    else if(!strcmp(cmd, "hsp"))
        run_hsp(g_strdup(model_name), input_path);
    else if(!strcmp(cmd, "seeds"))
        run_seeds(g_strdup(model_name), input_path);
    else if(!strcmp(cmd, "sdp"))
        run_sdp(g_strdup(model_name), input_path, subopt_max > 0 ? subopt_max : 1, subopt_threshold);
    else if(!strcmp(cmd, "golden"))
        run_golden(g_strdup(model_name), input_path, with_splice, revcomp_target,
                   subopt_max, subopt_threshold, derived);
    else
        g_error("unknown cmd [%s]", cmd);
    return 0;
    }
