/* c4gpu_shim.c — the ONE file a maintainer adds to exonerate to run C4 Viterbi calls on libc4gpu.so.
 *
 * It provides `Bootstrapper_lookup`, the name-keyed plug-in table Viterbi_create consults
 * (src/c4/viterbi.c:81-90).  For names of accelerated (model x mode) functions it returns a
 * Viterbi_DP_Func (src/c4/viterbi.h:95-98) backed by the GPU engine; for everything else, and for calls
 * (sub-optimal blocking, soi != NULL, IS accelerated: the index is mirrored into a c4gpu_subopt) it hands over to the reference's own generated CPU function
 * (`Bootstrapper_lookup_cpu` = the generated lookup of the compiled-model archive, renamed at link time by
 * integration/Makefile).  Host code, model builders, FASTA I/O, GAM, printers: all untouched reference C.
 *
 * Build: integration/Makefile (needs the reference tree; compiled here against its own headers).
 */
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <unistd.h>
extern int on_exit(void (*function)(int, void *), void *arg);      /* glibc; hidden by _XOPEN_SOURCE */

#include "viterbi.h"
#include "ungapped.h"
#include "affine.h"
#include "intron.h"
#include "frameshift.h"
#include "splice.h"
#include "translate.h"
#include "submat.h"
#include "match.h"
#include "codegen.h"

#include "c4gpu.h"
#include "c4gpu_shim.h"

extern gpointer Bootstrapper_lookup_cpu(gchar *name);

static c4gpu_ctx * volatile shim_ctx = NULL;
static gboolean shim_tried = FALSE, shim_verbose = FALSE;

/* ---- command line: --gpu / --gpudevice / --gpubatch beside -C/--compiled (codegen.c:25-37) ---------------- */
/* exonerate.c:85 calls Codegen_ArgumentSet_create(arg) once while it assembles its option sets; the archive's
 * definition is renamed Codegen_ArgumentSet_create_cpu by the Makefile and this one adds ours after it. */
extern Codegen_ArgumentSet *Codegen_ArgumentSet_create_cpu(Argument *arg);
static struct { gboolean use_gpu; gint device; gint batch; } shim_args = {TRUE, 0, 2048};

static void shim_start_ctx(void);

Codegen_ArgumentSet *Codegen_ArgumentSet_create(Argument *arg){
    register Codegen_ArgumentSet *cas = Codegen_ArgumentSet_create_cpu(arg);
    register ArgumentSet *as;
    if(arg){
        as = ArgumentSet_create("GPU options (libc4gpu, MI355X)");
        ArgumentSet_add_option(as, '\0', "gpu", NULL,
                "Run the optimal (exhaustive / refinement) Viterbi passes on the GPU", "TRUE",
                Argument_parse_boolean, &shim_args.use_gpu);
        ArgumentSet_add_option(as, '\0', "gpudevice", "ordinal",
                "HIP device to use", "0", Argument_parse_int, &shim_args.device);
        ArgumentSet_add_option(as, '\0', "gpubatch", "pairs",
                "Pairs of an exhaustive run collected per GPU batch (0 = one Viterbi call at a time)", "2048",
                Argument_parse_int, &shim_args.batch);
        Argument_absorb_ArgumentSet(arg, as);
    } else {
        shim_start_ctx();                 /* options parsed: open the device beside the rest of the start-up */
        }
    return cas;
    }

/* The context is opened on a thread of its own as soon as the options are known (the first Codegen_ArgumentSet_create(NULL)
 * of Viterbi_create, while Analysis_create still builds its models): HIP start-up (~120 ms) and the first code-object loads
 * then run beside the host's own start-up work (reading the queries, building the word automaton) instead of in front of
 * the first batch.  shim_get_ctx waits for the context only; the code-object loads go on in the background. */
static gint64 shim_t0 = 0, shim_t_open = 0, shim_t_ready = 0, shim_t_warm = 0, shim_waited = 0;
/* C4GPU_SEGV_TRACE=1: a SIGSEGV / SIGABRT writes the faulting thread's stack to stderr before the default action (the way out of
 * short runs has ended in the runtime's own threads before: see shim_exit) */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void shim_fault(int sig){
    void *frames[48];
    static const char head[] = "c4gpu: fatal signal, stack of the faulting thread:\n";
    register int n = backtrace(frames, 48);
    if(write(2, head, sizeof(head) - 1) < 0)
        n = n + 0;
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
    }
/* The process's C4GPU_* variables, copied once before main (one thread, nothing else running) and read from the copy ever after.
 * getenv walks `environ` without a lock, and the HIP / HSA start-up on the device thread ADDS variables (setenv moves the array):
 * the main thread's look-ups in the seams' hot paths (one per seeded HSP) then ran into freed memory -- SIGSEGV inside getenv in
 * 3 % of the 0.3 s heuristic runs that overlap the device's start-up (tools/gpu_small_work_stress.sh: 9 of 300 before, 0 after). */
extern char **environ;
static struct { gchar *name, *value; } *shim_env_copy = NULL;
static gint shim_env_n = 0;
const gchar *shim_env(const gchar *name){
    register gint i;
    for(i = 0; i < shim_env_n; i++)
        if(!strcmp(shim_env_copy[i].name, name))
            return shim_env_copy[i].value;
    return NULL;
    }
static void shim_env_snapshot(void){
    register char **e;
    register gint n = 0;
    for(e = environ; e && *e; e++)
        if(!strncmp(*e, "C4GPU_", 6))
            n++;
    shim_env_copy = calloc(n + 1, sizeof(*shim_env_copy));
    for(e = environ; e && *e; e++){
        register const char *eq = strchr(*e, '=');
        if(strncmp(*e, "C4GPU_", 6) || !eq)
            continue;
        shim_env_copy[shim_env_n].name = strndup(*e, eq - *e);
        shim_env_copy[shim_env_n].value = strdup(eq + 1);
        shim_env_n++;
        }
    return;
    }
static void __attribute__((constructor)) shim_clock_start(void){
    shim_t0 = g_get_monotonic_time();
    shim_env_snapshot();
    if(shim_env("C4GPU_SEGV_TRACE")){
        signal(SIGSEGV, shim_fault);
        signal(SIGABRT, shim_fault);
        signal(SIGBUS, shim_fault);
        }
    }
void shim_mark(const gchar *what){          /* C4GPU_TRACE: where the wall time of a run goes */
    static gint on = -1;
    if(on < 0)
        on = shim_env("C4GPU_TRACE") ? 1 : 0;
    if(on)
        g_printerr("c4gpu mark: %8.1f ms  %s\n", (g_get_monotonic_time() - shim_t0) / 1e3, what);
    return;
    }
static GThread *shim_ctx_thread = NULL;
static GMutex shim_ctx_lock;
static GCond shim_ctx_cond;
static gboolean shim_ctx_ready = FALSE;
static gchar *shim_ctx_error = NULL;

/* The way out.  The device thread may still be loading code objects (c4gpu_ctx_warm) when a short run is over; the HIP
 * runtime registers its exit handlers while that thread loads it, so an ordinary exit would run them -- tearing the runtime
 * down -- under a thread that is still inside it (SIGSEGV at the end of a 0.2 s run, seen once in ~10 in round 3, then
 * papered over with _exit).  Now every way out first QUIESCES the device thread: the warm-up is told to stop before its next
 * load (c4gpu_ctx_warm_cancel), the thread is joined, and only then do the exit handlers run, with no thread left inside
 * the runtime.  Both ways out come through here: main()'s return (the wrapper below) and every exit() call of the
 * reference's own objects (general/argument.c's error handler among them), which the Makefile points at shim_exit with
 * objcopy --redefine-sym.  Then the ordinary exit(), handlers and all.  Rounds 3-4 left with _exit here (stdio flushed first)
 * because one short run in about two hundred ended with SIGSEGV after its complete output; round 5 found the cause elsewhere --
 * getenv on the main thread racing with the setenv calls of the HIP start-up on the device thread (shim_env above) -- and with
 * that gone the ordinary exit ended 1 500 of 1 500 short runs cleanly (profiles/r05_small_work_stress_exit.log), so _exit is now
 * the exception: C4GPU_FAST_EXIT=1 asks for it, and a way out that is taken while a flush thread is still on the device (an
 * error exit in the middle of a run) takes it too. */
static gboolean shim_joined = FALSE;
static gboolean shim_fast_exit(void){
    register const gchar *e = shim_env("C4GPU_FAST_EXIT");
    return shim_ctx_thread && e && (e[0] == '1');
    }
static gboolean shim_flush_threads_alive(void);
static void shim_quiesce(void){
    register GThread *t;
    g_mutex_lock(&shim_ctx_lock);
    t = shim_joined ? NULL : shim_ctx_thread;
    shim_joined = TRUE;
    g_mutex_unlock(&shim_ctx_lock);
    if(!t)
        return;
    c4gpu_ctx_warm_cancel();
    g_thread_join(t);
    return;
    }
void shim_exit(int status){
    shim_quiesce();
    if(shim_fast_exit() || shim_flush_threads_alive()){
        fflush(NULL);
        _exit(status);
        }
    exit(status);
    }

static gpointer shim_ctx_open(gpointer data){
    register c4gpu_ctx *ctx = NULL;
    register gchar *why = NULL;
    shim_t_open = g_get_monotonic_time();
    /* the first call into libc4gpu.so: the lazy front (integration/Makefile) loads it here, on this thread */
    if(c4gpu_abi_version() != C4GPU_ABI_VERSION)
        why = g_strdup_printf("libc4gpu.so has ABI version %d, this binary was built for %d", c4gpu_abi_version(), C4GPU_ABI_VERSION);
    else
        ctx = c4gpu_ctx_create(shim_device_ordinal());
    /* C4GPU_OWN_STREAM=1: a stream of its own instead of the default one (tried while looking for what held the word scans up
     * beside an SDP flight, profiles/r05_c5_cold.md: it was not the default stream, and the default stays) */
    if(ctx && shim_env("C4GPU_OWN_STREAM") && (c4gpu_ctx_own_stream(ctx) != 0))
        g_warning("c4gpu: %s -- launching in the default stream", c4gpu_last_error());
    shim_t_ready = g_get_monotonic_time();
    g_mutex_lock(&shim_ctx_lock);
    shim_ctx = ctx;
    if(why)
        shim_ctx_error = why;
    else if(!ctx)
        shim_ctx_error = g_strdup(c4gpu_last_error());
    shim_ctx_ready = TRUE;
    g_cond_broadcast(&shim_ctx_cond);
    g_mutex_unlock(&shim_ctx_lock);
    if(ctx && !shim_env("C4GPU_NO_WARM"))
        c4gpu_ctx_warm(ctx);
    shim_t_warm = g_get_monotonic_time();
    return NULL;
    }

static void shim_start_ctx(void){
    if(shim_tried)
        return;
    shim_tried = TRUE;
    shim_verbose = (shim_env("C4GPU_VERBOSE") != NULL);
    if((!shim_args.use_gpu) || shim_env("C4GPU_DISABLE"))
        return;
    g_mutex_lock(&shim_ctx_lock);
    shim_ctx_thread = g_thread_new("c4gpu-ctx", shim_ctx_open, NULL);
    g_mutex_unlock(&shim_ctx_lock);
    return;
    }

/* The program's main() (general/argument.c:319, renamed by the Makefile): its return is a way out like any other. */
extern int exonerate_main_cpu(int argc, char **argv);
int main(int argc, char **argv){
    register int rc = exonerate_main_cpu(argc, argv);
    shim_quiesce();
    if(shim_fast_exit()){
        fflush(NULL);
        _exit(rc);
        }
    return rc;
    }

gint shim_device_ordinal(void){
    return shim_env("C4GPU_DEVICE") ? atoi(shim_env("C4GPU_DEVICE")) : shim_args.device;
    }

c4gpu_ctx *shim_get_ctx(void){
    register gchar *why = NULL;
    register c4gpu_ctx *ctx;
    shim_start_ctx();
    /* ready, error and the context itself are the device thread's to write: read under its lock */
    g_mutex_lock(&shim_ctx_lock);
    if(shim_ctx_thread && !shim_ctx_ready){
        gint64 w0 = g_get_monotonic_time();
        while(!shim_ctx_ready)
            g_cond_wait(&shim_ctx_cond, &shim_ctx_lock);
        shim_waited += g_get_monotonic_time() - w0;
        }
    why = shim_ctx_error;
    shim_ctx_error = NULL;
    ctx = shim_ctx;
    g_mutex_unlock(&shim_ctx_lock);
    if(why){
        g_warning("c4gpu: %s -- using the CPU Viterbi", why);
        g_free(why);
        }
    return ctx;
    }

/* The seams in front of SMALL pieces of work (word scan, HSP extension, BSDP's sub-DPs, SDP) ask this way: the context if
 * the device is open, NULL while it is still being opened -- such a piece then keeps the reference's own function (the
 * results are the same either way), and a run that is over before the device is ready never waits for it: the whole
 * est2genome BSDP run of 32 x 32 takes the reference 0.23 s, the device needs 0.15-0.2 s to open.  The seams in front of
 * large dynamic programmes (the exhaustive batching seam, Viterbi calls of C4GPU_MIN_CELLS and more) do wait
 * (shim_get_ctx).  C4GPU_WAIT=1: every seam waits (the tests: they count what the device served). */
c4gpu_ctx *shim_ctx_nowait(void){
    register gboolean ready;
    shim_start_ctx();
    if((!shim_ctx_thread) || shim_env("C4GPU_WAIT"))
        return shim_get_ctx();
    g_mutex_lock(&shim_ctx_lock);
    ready = shim_ctx_ready;
    g_mutex_unlock(&shim_ctx_lock);
    return ready ? shim_get_ctx() : NULL;
    }

/* ---- C4_Model (closed) -> c4gpu_model ------------------------------------------------------------- */

/* --annotation (match.c:276-281: no 1:1 DNA match inside a DNA query's CDS): the Optimal seams -- the per-call shim and the
 * exhaustive batches -- hand the annotation to the library (c4gpu_batch_set_annotation) and set this around their flattening;
 * the other seams (BSDP's sub-DPs, SDP, seeding: their kernels read their own tables) still leave such runs to the
 * reference's functions. */
static __thread gboolean shim_annotation_ok = FALSE;

gboolean shim_flatten_any(C4_Model *m, Ungapped_Data *ud, c4gpu_model *out, gboolean allow_span){
    register guint i, j;
    memset(out, 0, sizeof(*out));
    if((m->state_list->len > C4GPU_MAX_STATES) || (m->transition_list->len > C4GPU_MAX_TRANSITIONS)
    || (m->calc_list->len > C4GPU_MAX_CALCS) || (m->shadow_list->len > C4GPU_MAX_SHADOWS))
        return FALSE;
    g_strlcpy(out->name, m->name, C4GPU_NAME_LEN);
    out->n_states = m->state_list->len;
    out->n_transitions = m->transition_list->len;
    out->n_calcs = m->calc_list->len;
    out->n_shadows = m->shadow_list->len;
    out->start_state = m->start_state->state->id;
    out->end_state = m->end_state->state->id;
    out->start_scope = m->start_state->scope;       /* C4_Scope values are the C4GPU_SCOPE_* values */
    out->end_scope = m->end_state->scope;
    out->max_query_advance = m->max_query_advance;
    out->max_target_advance = m->max_target_advance;
    out->total_shadow_designations = m->total_shadow_designations;
    out->query_alphabet = (ud->query->alphabet->type == Alphabet_Type_PROTEIN);
    out->target_alphabet = (ud->target->alphabet->type == Alphabet_Type_PROTEIN);
    if((m->start_state->cell_start_func || m->end_state->cell_end_func) && !allow_span)
        return FALSE;                                /* BSDP span models: only with their matrices (c4gpu_bsdp.c) */
    for(i = 0; i < m->state_list->len; i++){
        C4_State *s = m->state_list->pdata[i];
        g_strlcpy(out->state_names[i], s->name, C4GPU_NAME_LEN);
        }
    for(i = 0; i < m->calc_list->len; i++){
        C4_Calc *c = m->calc_list->pdata[i];
        c4gpu_calc *o = &out->calcs[i];
        g_strlcpy(o->name, c->name, C4GPU_NAME_LEN);
        o->max_score = c->max_score;
        o->protect = c->protect;
        /* calc functions are file-static in the reference: recognised by the names the model builders give */
        if(!strcmp(c->name, "match")){
            if(out->query_alphabet && !out->target_alphabet) o->kind = C4GPU_CALC_MATCH_P2D;
            else if(out->query_alphabet && out->target_alphabet) o->kind = C4GPU_CALC_MATCH_PROTEIN;
            else if(!out->query_alphabet && !out->target_alphabet) o->kind = C4GPU_CALC_MATCH_DNA;
            else return FALSE;
            if(ud->query->annotation && (!shim_annotation_ok)) return FALSE; /* cds veto, match.c:276-281 */
        } else if((!strcmp(c->name, "gap open")) || (!strcmp(c->name, "gap extend"))
               || (!strcmp(c->name, "frameshift"))){
            o->kind = C4GPU_CALC_CONST;
            o->value = c->calc_func ? c->calc_func(0, 0, ud) : c->max_score;
        } else if(!strncmp(c->name, "5'ss forward", 12)){
            o->kind = C4GPU_CALC_SPLICE_PRE; o->param = C4GPU_SS5_FORWARD;
            o->value = Intron_ArgumentSet_create(NULL)->intron_open_penalty;
        } else if(!strncmp(c->name, "3'ss forward", 12)){
            o->kind = C4GPU_CALC_SPLICE_POST; o->param = C4GPU_SS3_FORWARD;
        } else if(!strncmp(c->name, "3'ss reverse", 12)){
            o->kind = C4GPU_CALC_SPLICE_PRE; o->param = C4GPU_SS3_REVERSE;
            o->value = Intron_ArgumentSet_create(NULL)->intron_open_penalty;
        } else if(!strncmp(c->name, "5'ss reverse", 12)){
            o->kind = C4GPU_CALC_SPLICE_POST; o->param = C4GPU_SS5_REVERSE;
        } else if(!strncmp(c->name, "phase1post to dst", 17)){
            o->kind = C4GPU_CALC_PHASE_POST; o->param = 1;
        } else if(!strncmp(c->name, "phase2post to dst", 17)){
            o->kind = C4GPU_CALC_PHASE_POST; o->param = 2;
        } else {
            return FALSE;
            }
        }
    for(i = 0; i < m->transition_list->len; i++){
        C4_Transition *t = m->transition_list->pdata[i];
        c4gpu_transition *o = &out->transitions[i];
        g_strlcpy(o->name, t->name, C4GPU_NAME_LEN);
        o->input = t->input->id;
        o->output = t->output->id;
        o->advance_query = t->advance_query;
        o->advance_target = t->advance_target;
        o->calc = t->calc ? t->calc->id : -1;
        o->label = t->label;
        for(j = 0; j < t->dst_shadow_list->len; j++)
            o->dst_shadow_mask |= 1u << ((C4_Shadow*)t->dst_shadow_list->pdata[j])->id;
        }
    for(i = 0; i < m->shadow_list->len; i++){
        C4_Shadow *s = m->shadow_list->pdata[i];
        c4gpu_shadow *o = &out->shadows[i];
        g_strlcpy(o->name, s->name, C4GPU_NAME_LEN);
        o->designation = s->designation;
        o->on_target = !strncmp(s->name, "target intron", 13);
        if(!o->on_target)
            return FALSE;
        for(j = 0; j < s->src_state_list->len; j++)
            o->src_state_mask |= 1u << ((C4_State*)s->src_state_list->pdata[j])->id;
        for(j = 0; j < s->dst_transition_list->len; j++)
            o->dst_transition_mask |= ((uint64_t)1) << ((C4_Transition*)s->dst_transition_list->pdata[j])->id;
        }
    return c4gpu_model_is_accelerated(out);
    }

static gboolean shim_flatten(C4_Model *m, Ungapped_Data *ud, c4gpu_model *out){
    return shim_flatten_any(m, ud, out, FALSE);
    }
/* ... for a seam that passes a query's annotation on */
static gboolean shim_flatten_annotated(C4_Model *m, Ungapped_Data *ud, c4gpu_model *out){
    register gboolean ok;
    shim_annotation_ok = TRUE;
    ok = shim_flatten_any(m, ud, out, FALSE);
    shim_annotation_ok = FALSE;
    return ok;
    }
/* the annotation as the library takes it (only a DNA query's counts: match.c:277) */
static void shim_cds(Sequence *query, int32_t *start, int32_t *length){
    *start = 0; *length = 0;
    if(query->annotation && (query->alphabet->type == Alphabet_Type_DNA)){
        *start = query->annotation->cds_start;
        *length = query->annotation->cds_length;
        }
    return;
    }

/* the static ArgumentSets + Match tables -> c4gpu_params */
void shim_params(Ungapped_Data *ud, c4gpu_params *p){
    register Affine_ArgumentSet *aas = Affine_ArgumentSet_create(NULL);
    register Intron_ArgumentSet *ias = Intron_ArgumentSet_create(NULL);
    register Frameshift_ArgumentSet *fas = Frameshift_ArgumentSet_create(NULL);
    register gint i, j, k;
    SplicePredictor *sp[4];
    c4gpu_params_default(p);
    for(i = 0; i < SUBMAT_ALPHABETSIZE; i++)
        for(j = 0; j < SUBMAT_ALPHABETSIZE; j++){
            p->dna_submat[i][j] = ud->mas->dna_submat->matrix[i][j];
            p->protein_submat[i][j] = ud->mas->protein_submat->matrix[i][j];
            }
    memcpy(p->submat_index, ud->mas->dna_submat->index, 256);
    if(ud->mas->translate){
        memcpy(p->nt2d, ud->mas->translate->nt2d, 256);
        memcpy(p->trans, ud->mas->translate->trans, 4096);
        memcpy(p->aa, ud->mas->translate->aa, 40);
        }
    p->gap_open = aas->gap_open; p->gap_extend = aas->gap_extend;
    p->codon_gap_open = aas->codon_gap_open; p->codon_gap_extend = aas->codon_gap_extend;
    p->min_intron = ias->min_intron; p->max_intron = ias->max_intron;
    p->intron_open_penalty = ias->intron_open_penalty;
    p->frameshift_penalty = fas->frameshift_penalty;
    if(ias->sps){
        sp[C4GPU_SS5_FORWARD] = ias->sps->ss5_forward; sp[C4GPU_SS3_FORWARD] = ias->sps->ss3_forward;
        sp[C4GPU_SS3_REVERSE] = ias->sps->ss3_reverse; sp[C4GPU_SS5_REVERSE] = ias->sps->ss5_reverse;
        for(k = 0; k < 4; k++){
            if(sp[k]->model_length > C4GPU_SPLICE_MAX_LEN)
                continue;
            p->splice[k].gtag_only = sp[k]->gtag_only ? 1 : 0;   /* --forcegtag, splice.c:290-293 */
            if(sp[k]->gtag_only){
                p->splice[k].expect_one = sp[k]->gtag_only->expect_one;
                p->splice[k].expect_two = sp[k]->gtag_only->expect_two;
                }
            p->splice[k].model_length = sp[k]->model_length;
            p->splice[k].splice_after = sp[k]->model_splice_after;
            memcpy(p->splice[k].index, sp[k]->index, 256);
            for(i = 0; i < sp[k]->model_length; i++)
                for(j = 0; j < 5; j++)
                    p->splice[k].data[i][j] = sp[k]->model_data[i][j];
            }
        }
    }

void shim_hsp_params(c4gpu_params *p){
    register Match_ArgumentSet *mas = Match_ArgumentSet_create(NULL);
    register gint i, j;
    memset(p, 0, sizeof(*p));
    c4gpu_params_default(p);
    for(i = 0; i < SUBMAT_ALPHABETSIZE; i++)
        for(j = 0; j < SUBMAT_ALPHABETSIZE; j++){
            p->dna_submat[i][j] = mas->dna_submat->matrix[i][j];
            p->protein_submat[i][j] = mas->protein_submat->matrix[i][j];
            }
    memcpy(p->submat_index, mas->dna_submat->index, 256);
    if(mas->translate){
        memcpy(p->nt2d, mas->translate->nt2d, 256);
        memcpy(p->trans, mas->translate->trans, 4096);
        memcpy(p->aa, mas->translate->aa, 40);
        }
    return;
    }

/* ---- one Viterbi call ----------------------------------------------------------------------------------- */

static GMutex shim_device_lock[2];       /* one batch on a context at a time (the batching seam below; slot 1: C4GPU_SLOTS=2) */

static C4_Score shim_dp(C4_Model *model, Region *region, Viterbi_Data *vd, SubOpt_Index *soi,
                        gpointer user_data, int mode, Viterbi_DP_Func cpu_func){
    Ungapped_Data *ud = user_data;
    c4gpu_model fm;
    c4gpu_params params;
    c4gpu_pair pair;
    c4gpu_viterbi_job job;
    c4gpu_viterbi_result r;
    register gchar *qstr, *tstr;
    register gint i, j, k, l, cs = vd->vr->cell_size;
    register C4_Score score;
    c4gpu_subopt *blocked = NULL;
    static gdouble min_cells = -1.0;
    if(min_cells < 0.0)
        min_cells = shim_env("C4GPU_MIN_CELLS") ? atof(shim_env("C4GPU_MIN_CELLS")) : 1e5;
    /* a single small rectangle is faster on the host than one launch + copies (a batch is another matter) */
    if(cpu_func && (((gdouble)region->query_length + 1.0) * ((gdouble)region->target_length + 1.0) < min_cells))
        return cpu_func(model, region, vd, soi, user_data);
    if((!shim_get_ctx()) || (!shim_flatten_annotated(model, ud, &fm))){
        if(!cpu_func)
            g_error("c4gpu shim: no CPU implementation to fall back to");
        return cpu_func(model, region, vd, soi, user_data);
        }
    qstr = Sequence_get_str(ud->query);
    tstr = Sequence_get_str(ud->target);
    memset(&job, 0, sizeof(job));
    job.pair = 0;
    job.region.query_start = region->query_start;   job.region.target_start = region->target_start;
    job.region.query_length = region->query_length; job.region.target_length = region->target_length;
    if(vd->continuation){                            /* viterbi.c:705-714 */
        job.use_continuation = 1;
        job.continuation.first_state = vd->continuation->first_state->id;
        job.continuation.final_state = vd->continuation->final_state->id;
        for(l = 0; l < cs; l++)
            job.continuation.first_cell[l] = vd->continuation->first_cell[l];
        }
    if(mode == C4GPU_MODE_FIND_CHECKPOINTS)
        job.checkpoint_count = vd->checkpoint->checkpoint_list->len;
    if(soi){
        /* the index holds the blocked cells of this region in region coordinates, one row per target
         * position plus a trailing blank row (subopt.c:270-318): hand them over as sequence coordinates */
        blocked = c4gpu_subopt_create(ud->query->len, ud->target->len);
        for(k = 0; k < (gint)soi->row_list->len - 1; k++){
            SubOpt_Index_Row *row = soi->row_list->pdata[k];
            for(l = 0; l < row->total; l++)
                c4gpu_subopt_add_point(blocked, region->query_start + row->query_pos[l],
                                                region->target_start + row->target_pos);
            }
        job.subopt = blocked;
        }
    /* The reference makes many Viterbi calls on one pair (region, checkpoints, every sub-alignment): the
     * pair stays resident on the device between them.  Identity = content: the residues (a freed Sequence's
     * address may be reused), the whole flattened model (tables, calc values, scopes -- derived models share long
     * name prefixes) and the scoring parameters in force. */
    /* The exhaustive seam may have batch N + 1 on the device (a `c4gpu-flush` thread inside c4gpu_batch_run on this very
     * context) while the main thread replays batch N; a replay that leaves its recorded rounds lands here.  A context serves
     * one batch at a time: this call takes the device lock of the seam's slot 0 (shim_ctx) for its device part, i.e. it waits
     * for a flush in flight (rare: -S yes past the recorded rounds, a batch that failed, a region the dry run did
     * not see). */
    g_mutex_lock(&shim_device_lock[0]);
    {
        static c4gpu_batch *resident = NULL;
        static guint64 resident_key = 0;
        static gboolean resident_failed = FALSE;     /* creation failed for resident_key: do not retry per call */
        static gint resident_scope[2];
        register guint64 key = 1469598103934665603ULL;
        register const guchar *c;
        register gsize n;
        for(c = (const guchar*)qstr; *c; c++) key = (key ^ *c) * 1099511628211ULL;
        key = (key ^ 0xff) * 1099511628211ULL;
        for(c = (const guchar*)tstr; *c; c++) key = (key ^ *c) * 1099511628211ULL;
        key ^= ((guint64)ud->query->len << 32) ^ (guint64)ud->target->len;
        {   /* (the resident copy of an annotated query carries the annotation in its residue codes) */
            int32_t cs, cl;
            shim_cds(ud->query, &cs, &cl);
            key = (key ^ (guint64)(guint32)cs) * 1099511628211ULL;
            key = (key ^ (guint64)(guint32)cl) * 1099511628211ULL;
        }
        memset(&params, 0, sizeof(params));
        shim_params(ud, &params);
        /* a continuation copy differs from its model in the scopes only (viterbi.c:68-76) and the device picks the
         * CORNER kernels by the call, not by the batch: scopes are compared for the other calls only */
        {
            static c4gpu_model keyed;
            keyed = fm;
            keyed.start_scope = keyed.end_scope = 0;
            for(c = (const guchar*)&keyed, n = sizeof(keyed); n; n--, c++) key = (key ^ *c) * 1099511628211ULL;
        }
        for(c = (const guchar*)&params, n = sizeof(params); n; n--, c++) key = (key ^ *c) * 1099511628211ULL;
        if((!resident && !(resident_failed && (key == resident_key))) || (key != resident_key)
        || ((!vd->continuation) && ((resident_scope[0] != fm.start_scope) || (resident_scope[1] != fm.end_scope)))){
            if(resident)
                c4gpu_batch_destroy(resident);
            pair.query = (const uint8_t*)qstr;  pair.query_len = ud->query->len;
            pair.target = (const uint8_t*)tstr; pair.target_len = ud->target->len;
            resident = c4gpu_batch_create(shim_ctx, &fm, &params, &pair, 1);
            if(resident){
                int32_t cs, cl;
                shim_cds(ud->query, &cs, &cl);
                if((cl > 0) && (c4gpu_batch_set_annotation(resident, &cs, &cl) != 0)){
                    c4gpu_batch_destroy(resident);
                    resident = NULL;
                    }
                }
            resident_key = key;
            resident_failed = (resident == NULL);
            resident_scope[0] = fm.start_scope; resident_scope[1] = fm.end_scope;
            }
        i = resident ? c4gpu_batch_viterbi(resident, mode, &job, 1, &r) : -1;
    }
    g_mutex_unlock(&shim_device_lock[0]);
    if(blocked)
        c4gpu_subopt_destroy(blocked);
    if(i != 0){
        g_free(qstr); g_free(tstr);
        if(!cpu_func)
            g_error("c4gpu shim: %s, and no CPU implementation to fall back to", c4gpu_last_error());
        g_warning("c4gpu: %s -- using the CPU Viterbi for this call", c4gpu_last_error());
        return cpu_func(model, region, vd, soi, user_data);
        }
    if(shim_verbose)
        g_message("c4gpu: %s mode %d region %d %d %d %d%s -> %d", model->name, mode, region->query_start,
                  region->target_start, region->query_length, region->target_length,
                  soi?" with blocked cells":"", r.score);
    score = r.score;
    /* out-parameters by mode (viterbi.c:464-478,633-653,813-832) */
    vd->curr_query_end = r.query_end;
    vd->curr_target_end = r.target_end;
    if(mode == C4GPU_MODE_FIND_REGION){
        if(vd->vr->region_start_query_id != -1)
            vd->curr_query_start = r.query_start;
        if(vd->vr->region_start_target_id != -1)
            vd->curr_target_start = r.target_start;
        if(vd->alignment_region){                    /* Viterbi_Data_finalise */
            if(vd->vr->region_start_query_id != -1)
                vd->alignment_region->query_start = vd->curr_query_start + region->query_start;
            if(vd->vr->region_start_target_id != -1)
                vd->alignment_region->target_start = vd->curr_target_start + region->target_start;
            vd->alignment_region->query_length = vd->curr_query_end - vd->curr_query_start;
            vd->alignment_region->target_length = vd->curr_target_end - vd->curr_target_start;
            }
        }
    if(mode == C4GPU_MODE_FIND_PATH){
        /* lay the transition pointers along the path so that Viterbi_Data_create_Alignment
         * (viterbi.c:342-392) walks exactly it; everything else stays NULL (g_malloc0) */
        i = r.query_start; j = r.target_start;
        for(k = 0; k < r.n_ops; k++){
            C4_Transition *t = model->transition_list->pdata[r.ops[k]];
            i += t->advance_query; j += t->advance_target;
            vd->traceback[i][j][t->output->id] = t;
            }
        }
    if(mode == C4GPU_MODE_FIND_CHECKPOINTS){
        register gint nstates = model->state_list->len, mta = model->max_target_advance,
                      ql = region->query_length;
        for(k = 0; k < (gint)vd->checkpoint->checkpoint_list->len; k++){
            C4_Score ****cp = vd->checkpoint->checkpoint_list->pdata[k];
            register gint row, s;
            for(row = 0; row < mta; row++)
                for(i = 0; i <= ql; i++)
                    for(s = 0; s < nstates; s++)
                        for(l = 0; l < cs; l++)
                            cp[row][i][s][l] = r.checkpoints[(((( (size_t)k * mta + row) * (ql+1) + i)
                                                               * nstates + s) * cs) + l];
            }
        vd->checkpoint->last_srp = r.last_srp;
        vd->checkpoint->counter = vd->checkpoint->checkpoint_list->len;
        }
    if(vd->continuation)
        for(l = 0; l < cs; l++)
            vd->continuation->final_cell[l] = r.final_cell[l];
    c4gpu_viterbi_result_clear(&r);
    g_free(qstr);
    g_free(tstr);
    return score;
    }

/* ---- the plug-in table --------------------------------------------------------------------------------- */
/* One Viterbi_DP_Func per looked-up name: each remembers its mode and the CPU function of the same name. */

#define SHIM_SLOTS 64
static struct { gchar *name; int mode; Viterbi_DP_Func cpu; } shim_slot[SHIM_SLOTS];
static gint shim_slot_count = 0;

#define SHIM_FUNC(n) \
    static C4_Score shim_func_##n(C4_Model *model, Region *region, Viterbi_Data *vd, SubOpt_Index *soi, \
                                  gpointer user_data){ \
        return shim_dp(model, region, vd, soi, user_data, shim_slot[n].mode, shim_slot[n].cpu); }
#define S4(a) SHIM_FUNC(a##0) SHIM_FUNC(a##1) SHIM_FUNC(a##2) SHIM_FUNC(a##3)
SHIM_FUNC(0) SHIM_FUNC(1) SHIM_FUNC(2) SHIM_FUNC(3) SHIM_FUNC(4) SHIM_FUNC(5) SHIM_FUNC(6) SHIM_FUNC(7)
SHIM_FUNC(8) SHIM_FUNC(9) SHIM_FUNC(10) SHIM_FUNC(11) SHIM_FUNC(12) SHIM_FUNC(13) SHIM_FUNC(14) SHIM_FUNC(15)
SHIM_FUNC(16) SHIM_FUNC(17) SHIM_FUNC(18) SHIM_FUNC(19) SHIM_FUNC(20) SHIM_FUNC(21) SHIM_FUNC(22) SHIM_FUNC(23)
SHIM_FUNC(24) SHIM_FUNC(25) SHIM_FUNC(26) SHIM_FUNC(27) SHIM_FUNC(28) SHIM_FUNC(29) SHIM_FUNC(30) SHIM_FUNC(31)
static Viterbi_DP_Func shim_funcs[] = {
    shim_func_0, shim_func_1, shim_func_2, shim_func_3, shim_func_4, shim_func_5, shim_func_6, shim_func_7,
    shim_func_8, shim_func_9, shim_func_10, shim_func_11, shim_func_12, shim_func_13, shim_func_14, shim_func_15,
    shim_func_16, shim_func_17, shim_func_18, shim_func_19, shim_func_20, shim_func_21, shim_func_22, shim_func_23,
    shim_func_24, shim_func_25, shim_func_26, shim_func_27, shim_func_28, shim_func_29, shim_func_30, shim_func_31};

/* "…_find_32_score" / "_path" / "_region" / "_checkpoint" / "_path_32_continuation" (optimal.c:31-67) */
static int shim_mode_of(const gchar *name){
    if(g_str_has_suffix(name, "_find_32_score")) return C4GPU_MODE_FIND_SCORE;
    if(g_str_has_suffix(name, "_find_32_path")) return C4GPU_MODE_FIND_PATH;
    if(g_str_has_suffix(name, "_find_32_path_32_continuation")) return C4GPU_MODE_FIND_PATH;
    if(g_str_has_suffix(name, "_find_32_region")) return C4GPU_MODE_FIND_REGION;
    if(g_str_has_suffix(name, "_find_32_checkpoint")) return C4GPU_MODE_FIND_CHECKPOINTS;
    return -1;
    }

gpointer Bootstrapper_lookup(gchar *name){
    register gint i;
    register int mode = shim_mode_of(name);
    register Viterbi_DP_Func cpu = (Viterbi_DP_Func)Bootstrapper_lookup_cpu(name);
    /* only the Optimal (full Viterbi) functions are ours: "optimal_58_<model>_32_find_32_<mode>" */
    if((mode < 0) || strncmp(name, "optimal_58_", 11) || shim_env("C4GPU_DISABLE") || (!shim_args.use_gpu))
        return (gpointer)cpu;
    /* no compiled CPU function of this name (the archive was built without the model) and no device either: leave
     * the slot empty, Viterbi_calculate then runs the reference's interpreted loop (viterbi.c:855-859) */
    if((!cpu) && (!shim_get_ctx()))
        return NULL;
    for(i = 0; i < shim_slot_count; i++)
        if(!strcmp(shim_slot[i].name, name))
            return (gpointer)shim_funcs[i];
    if(shim_slot_count >= 32)
        return (gpointer)cpu;
    shim_slot[shim_slot_count].name = g_strdup(name);
    shim_slot[shim_slot_count].mode = mode;
    shim_slot[shim_slot_count].cpu = cpu;
    return (gpointer)shim_funcs[shim_slot_count++];
    }

/* ---- the batching seam ------------------------------------------------------------------------------------ */
/* One pair per Viterbi call leaves a 256-CU device idle.  The exhaustive front end hands every pair to
 * GAM_Result_exhaustive_create (analysis.c:218-230 -> gam.c:1139) and prints what comes back at once; this
 * replacement (the archive's own definition is renamed GAM_Result_exhaustive_create_cpu by the Makefile)
 * only COLLECTS (gam, query, target) and returns NULL.  Every C4GPU_BATCH pairs, and before the final
 * GAM_report, the collected pairs go through c4gpu_batch_run / c4gpu_batch_next_paths in one piece; then
 * each pair is replayed IN SUBMISSION ORDER through the reference's own GAM_Result_exhaustive_create_cpu,
 * whose Optimal_find_path calls (also fronted here) are answered from the batch results.  Thresholds that
 * move while results are submitted (--bestn, gam.c:683-692) are honoured in the replay: the k-th best path
 * of a pair does not depend on the threshold, only whether it is reported does. */
#include "gam.h"
#include "optimal.h"
#include "alignment.h"
#include "modeltype.h"

extern GAM_Result *GAM_Result_exhaustive_create_cpu(GAM *gam, Sequence *query, Sequence *target);
extern void GAM_report_cpu(GAM *gam);
extern Alignment *Optimal_find_path_cpu(Optimal *optimal, Region *region, gpointer user_data,
                                        C4_Score threshold, SubOpt *subopt);

typedef struct {
    GAM *gam;
    Sequence *query, *target;
    c4gpu_alignment *round;        /* round[k] = k-th alignment of the pair; valid == 0 ends the loop */
    gint round_total;
    gboolean done;
} ShimPending;

static GPtrArray *shim_pending = NULL;
static gdouble shim_pending_bytes = 0.0;   /* device footprint of the collected pairs (no sharing assumed) */
static ShimPending *shim_replay_pair = NULL;
static gint shim_replay_call = 0;

gint shim_batch_size(void){
    static gint size = -1;
    if(size < 0)
        size = shim_env("C4GPU_BATCH") ? atoi(shim_env("C4GPU_BATCH")) : shim_args.batch;
    return size;
    }

static gboolean shim_can_batch(GAM *gam, Sequence *query, Sequence *target){
    register gpointer ud;
    register gboolean ok;
    c4gpu_model fm;
    if((shim_batch_size() <= 0) || (!gam->optimal) || (!shim_get_ctx()))
        return FALSE;
    if(gam->gas->refinement != GAM_Refinement_NONE)
        return FALSE;  /* refinement re-enters Optimal_find_path with other regions */
    if((gam->optimal->type & (Optimal_Type_SCORE|Optimal_Type_PATH|Optimal_Type_REDUCED_SPACE))
       != (Optimal_Type_SCORE|Optimal_Type_PATH|Optimal_Type_REDUCED_SPACE))
        return FALSE;
    {   /* whether the model is one of the accelerated families does not depend on the pair: asked once per model (the
         * per-pair user data the flattening reads -- Match tables, splice predictors -- are the process's static argument
         * sets); at 4 096 pairs of a run the repeated check was 0.1 ms per pair */
        static GAM *seen_gam = NULL;                 /* (pinned while pairs of it are pending: GAM_share below) */
        static C4_Model *seen_model = NULL;
        static gboolean seen_ok = FALSE;
        static Alphabet_Type seen_q, seen_t;
        if((seen_gam == gam) && (seen_model == gam->optimal->find_path->model) && (seen_q == query->alphabet->type)
        && (seen_t == target->alphabet->type))
            return seen_ok;
        seen_gam = gam;
        ud = Model_Type_create_data(gam->gas->type, query, target);
        ok = shim_flatten_annotated(gam->optimal->find_path->model, ud, &fm);
        Model_Type_destroy_data(gam->gas->type, ud);
        seen_model = gam->optimal->find_path->model; seen_q = query->alphabet->type; seen_t = target->alphabet->type;
        seen_ok = ok;
        }
    return ok;
    }

/* GAM_QueryInfo_create (gam.c:466-487): the --percent threshold of a query = the best self-comparison score
 * over the model's match types x percent / 100, never below --score.  The function is file-static in the
 * reference; the structures it walks are public. */
static C4_Score shim_query_threshold(GAM *gam, Sequence *query){
    register C4_Score threshold = 0;
    register guint i;
    register gint j;
    if(!gam->gas->percent_threshold)
        return gam->gas->threshold;
    for(i = 0; i < gam->match_list->len; i++){
        register Match *match = gam->match_list->pdata[i];
        register Match_Score th = 0;
        for(j = 0; j < (gint)query->len; j += match->query->advance)
            th += match->query->self_func(match->query, query, j);
        if(threshold < th)
            threshold = th;
        }
    threshold *= gam->gas->percent_threshold;
    threshold /= 100;
    if(threshold < gam->gas->threshold)
        threshold = gam->gas->threshold;
    return threshold;
    }

/* A flush has three parts: PREPARE on the main thread (the reference's Sequence / Model_Type objects are read here: flattened
 * residues, flattened model and parameters, per-query thresholds), DEVICE (library calls only: stage the batch, run the rounds,
 * keep every pair's alignments) and REPLAY on the main thread (each pair through the reference's own
 * GAM_Result_exhaustive_create, in submission order).  The device part of a batch runs on a thread of its own while the main
 * thread goes on reading sequences and collecting the next batch, and while it replays the batch before: at most one batch on
 * the device and one behind it, replayed strictly in submission order (C4GPU_ASYNC=0: one after the other on the main thread,
 * the form of rounds 1-4). */
typedef struct {
    GPtrArray *todo;
    guint n;
    GAM *gam;
    C4_Score threshold;
    gint dpmemory, rounds_max;
    gboolean flattened;
    c4gpu_model fm;
    c4gpu_params params;
    c4gpu_pair *pair;
    gchar **str;
    c4gpu_score *per_pair;            /* --percent thresholds, or NULL */
    int32_t *cds_start, *cds_length;  /* --annotation: per pair, or NULL where no query of the batch is annotated */
    GThread *thread;
    gint rounds_done;
    gint slot;                        /* which of the two resident batches carries it */
    gchar *error;
    gint64 t_start, t_prepared, t_dev0, t_dev1;
} ShimFlushJob;

/* One batch on the device at a time.  C4GPU_SLOTS=2 (an experiment that stays opt-in): TWO resident batches (slots), taken in turn
 * by the flushes, so that the device part of a flush starts beside the batch before it; slot 1 has a context and a stream of its
 * own.  Measured on config 4 through the command line (64 x 64, two flushes of 2 048): the two device parts take 841 + 941 ms side
 * by side against 709 + 627 ms one after the other -- end of the run at 1 675 / 1 957 / 3 067 ms against 1 751 / 1 833 / 1 907 ms:
 * 0.08-0.15 s gained at best, and one run in three lost a second in a stalled allocation beside the other batch's kernels. */
static c4gpu_ctx *shim_ctx2 = NULL;
static guint shim_flush_seq = 0;
static ShimFlushJob *shim_in_flight = NULL;
static volatile gint shim_device_busy = 0;        /* a flush thread is between taking and releasing the device */

static gboolean shim_async(void){
    static gint on = -1;
    if(on < 0)
        on = (shim_env("C4GPU_ASYNC") && (atoi(shim_env("C4GPU_ASYNC")) == 0)) ? 0 : 1;
    return on;
    }

static ShimFlushJob *shim_flush_prepare(void){
    register guint i, n;
    register ShimPending *sp;
    register gpointer ud;
    register ShimFlushJob *job;
    GPtrArray *todo = shim_pending;
    if((!todo) || (!todo->len))
        return NULL;
    shim_mark("flush: start");
    shim_pending = NULL;          /* pairs submitted from here on start a new collection */
    shim_pending_bytes = 0.0;
    job = g_new0(ShimFlushJob, 1);
    job->slot = (shim_env("C4GPU_SLOTS") && (atoi(shim_env("C4GPU_SLOTS")) == 2)) ? (gint)(shim_flush_seq++ & 1) : 0;
    job->t_start = g_get_monotonic_time();
    job->todo = todo;
    n = job->n = todo->len;
    sp = todo->pdata[0];
    job->gam = sp->gam;
    job->threshold = job->gam->gas->threshold;
    job->dpmemory = job->gam->optimal->find_path->vas->traceback_memory_limit;
    job->rounds_max = shim_env("C4GPU_BATCH_ROUNDS") ? atoi(shim_env("C4GPU_BATCH_ROUNDS")) : 4;
    if(!job->gam->gas->use_subopt)
        job->rounds_max = 1;
    job->pair = g_new0(c4gpu_pair, n);
    job->str = g_new0(gchar*, 2*n);
    {   /* one flattened copy per Sequence: pairs that share a Sequence share the buffer, and the library
         * keeps one device copy (and one set of splice arrays) per buffer */
        register GHashTable *flat = g_hash_table_new(g_direct_hash, g_direct_equal);
        for(i = 0; i < n; i++){
            register gchar *qs, *ts;
            sp = todo->pdata[i];
            if(!(qs = g_hash_table_lookup(flat, sp->query))){
                qs = job->str[2*i] = Sequence_get_str(sp->query);
                g_hash_table_insert(flat, sp->query, qs);
                }
            if(!(ts = g_hash_table_lookup(flat, sp->target))){
                ts = job->str[2*i+1] = Sequence_get_str(sp->target);
                g_hash_table_insert(flat, sp->target, ts);
                }
            job->pair[i].query = (const uint8_t*)qs;  job->pair[i].query_len = sp->query->len;
            job->pair[i].target = (const uint8_t*)ts; job->pair[i].target_len = sp->target->len;
            }
        g_hash_table_destroy(flat);
        }
    sp = todo->pdata[0];
    ud = Model_Type_create_data(job->gam->gas->type, sp->query, sp->target);
    memset(&job->fm, 0, sizeof(job->fm));
    if((job->flattened = shim_flatten_annotated(job->gam->optimal->find_path->model, ud, &job->fm)))
        shim_params(ud, &job->params);
    /* --annotation: the CDS of every annotated DNA query of the batch (c4gpu_batch_set_annotation) */
    job->cds_start = job->cds_length = NULL;
    for(i = 0; i < n; i++){
        int32_t cs, cl;
        sp = todo->pdata[i];
        shim_cds(sp->query, &cs, &cl);
        if(cl > 0){
            if(!job->cds_start){
                job->cds_start = g_new0(int32_t, n);
                job->cds_length = g_new0(int32_t, n);
                }
            job->cds_start[i] = cs; job->cds_length[i] = cl;
            }
        }
    Model_Type_destroy_data(job->gam->gas->type, ud);
    if(job->flattened && job->gam->gas->percent_threshold){
        /* one threshold per query (cached by Sequence): pairs below it stop after the score pass */
        register GHashTable *seen = g_hash_table_new(g_direct_hash, g_direct_equal);
        job->per_pair = g_new(c4gpu_score, n);
        for(i = 0; i < n; i++){
            gpointer v;
            sp = todo->pdata[i];
            if(!g_hash_table_lookup_extended(seen, sp->query, NULL, &v)){
                v = GINT_TO_POINTER(shim_query_threshold(job->gam, sp->query));
                g_hash_table_insert(seen, sp->query, v);
                }
            job->per_pair[i] = GPOINTER_TO_INT(v);
            }
        g_hash_table_destroy(seen);
        }
    job->t_prepared = g_get_monotonic_time();
    return job;
    }

/* library calls and plain memory only: runs on the flush thread */
static gpointer shim_flush_device(gpointer data){
    register ShimFlushJob *job = data;
    register guint i, n = job->n;
    register gint k = 0;
    register ShimPending *sp;
    register c4gpu_batch *batch = NULL;
    /* ONE batch object and one stage for the whole run (per model and parameters): a flush stages its pairs through the stage's
     * page-locked buffers and swaps them into the batch, which keeps its engine, launch lanes and launch buffers -- a batch
     * created and destroyed per flush allocated and freed ~15 GB of device memory each time (the second 2 048-pair flush of a
     * run took 1 630 ms on the device where the passes themselves take 520) */
    static c4gpu_batch *res_batches[2] = {NULL, NULL};
    static c4gpu_stage *res_stages[2] = {NULL, NULL};
    static c4gpu_model res_fms[2];
    static c4gpu_params res_paramss[2];
    register gint slot = job->slot;
    register c4gpu_ctx *ctx;
#define res_batch  res_batches[slot]
#define res_stage  res_stages[slot]
#define res_fm     res_fms[slot]
#define res_params res_paramss[slot]
    g_mutex_lock(&shim_device_lock[slot]);
    g_atomic_int_inc(&shim_device_busy);
    job->t_dev0 = g_get_monotonic_time();
    ctx = shim_ctx;
    if(slot == 1){
        if(!shim_ctx2){
            shim_ctx2 = c4gpu_ctx_create(shim_device_ordinal());
            if(shim_ctx2 && (c4gpu_ctx_own_stream(shim_ctx2) != 0)){
                c4gpu_ctx_destroy(shim_ctx2);
                shim_ctx2 = NULL;
                }
            }
        ctx = shim_ctx2;
        }
    if(job->flattened && ctx){
        if(res_stage && (memcmp(&res_fm, &job->fm, sizeof(res_fm)) || memcmp(&res_params, &job->params, sizeof(res_params)))){
            if(res_batch)
                c4gpu_batch_destroy(res_batch);
            c4gpu_stage_destroy(res_stage);
            res_batch = NULL;
            res_stage = NULL;
            }
        if(!res_stage){
            res_fm = job->fm;
            res_params = job->params;
            res_stage = c4gpu_stage_create(ctx, &res_fm, &res_params);
            }
        if(res_stage && (c4gpu_stage_load(res_stage, job->pair, n) == 0)){
            if(!res_batch)
                res_batch = c4gpu_batch_create(ctx, &res_fm, &res_params, job->pair, 1);      /* (swapped out at once) */
            if(res_batch && (c4gpu_batch_swap_stage(res_batch, res_stage) == 0)){
                batch = res_batch;
                c4gpu_batch_set_thresholds(batch, NULL);
                /* (NULL, NULL where no query of this batch is annotated: the batch leaves its annotated form) */
                if(c4gpu_batch_set_annotation(batch, job->cds_start, job->cds_length) != 0)
                    batch = NULL;
                }
            }
        }
#undef res_batch
#undef res_stage
#undef res_fm
#undef res_params
    if(batch && job->per_pair)
        c4gpu_batch_set_thresholds(batch, job->per_pair);
    if(batch && (c4gpu_batch_run(batch, 2, job->dpmemory, job->threshold) == 0)){
        for(k = 0; k < job->rounds_max; k++){
            register gint found = 1;
            if(k && ((found = c4gpu_batch_next_paths(batch, job->dpmemory, job->threshold)) < 0))
                break;
            for(i = 0; i < n; i++){
                sp = job->todo->pdata[i];
                if(sp->done)
                    continue;                          /* this pair left the loop in an earlier round */
                sp->round = g_renew(c4gpu_alignment, sp->round, k+1);
                if(c4gpu_batch_alignment(batch, i, &sp->round[k]) != 0)
                    memset(&sp->round[k], 0, sizeof(c4gpu_alignment));
                sp->round_total = k+1;
                sp->done = !sp->round[k].valid;        /* the entry that ends the pair's loop is kept */
                }
            if(!found){
                k++;
                break;
                }
            }
        job->rounds_done = k;
    } else {
        job->error = g_strdup(c4gpu_last_error());
        }
    job->t_dev1 = g_get_monotonic_time();
    (void)g_atomic_int_dec_and_test(&shim_device_busy);
    g_mutex_unlock(&shim_device_lock[slot]);
    return NULL;
    }

static void shim_flush_replay(ShimFlushJob *job){
    register guint i, n = job->n;
    register gint k;
    register ShimPending *sp;
    gint64 t_rep0, t_rep_create = 0, t_rep_submit = 0, t_rep_destroy = 0, t_wait0 = g_get_monotonic_time();
    if(job->thread){
        g_thread_join(job->thread);
        job->thread = NULL;
        }
    t_rep0 = g_get_monotonic_time();
    if(job->error)
        g_warning("c4gpu: %s -- batch falls back to per-call", job->error);
    else if(shim_verbose)
        g_message("c4gpu: batch of %d pairs, %d round(s) on the device", n, job->rounds_done);
    for(i = 0; i < 2*n; i++)
        g_free(job->str[i]);
    g_free(job->str);
    g_free(job->pair);
    g_free(job->per_pair);
    g_free(job->cds_start);
    g_free(job->cds_length);
    /* replay in submission order through the reference's own code */
    for(i = 0; i < n; i++){
        register GAM_Result *gam_result;
        gint64 r0 = shim_verbose ? g_get_monotonic_time() : 0, r1 = 0, r2 = 0;
        sp = job->todo->pdata[i];
        shim_replay_pair = sp;
        shim_replay_call = 0;
        gam_result = GAM_Result_exhaustive_create_cpu(sp->gam, sp->query, sp->target);
        shim_replay_pair = NULL;
        if(shim_verbose)
            r1 = g_get_monotonic_time();
        if(gam_result){
            GAM_Result_submit(gam_result);
            if(shim_verbose)
                r2 = g_get_monotonic_time();
            GAM_Result_destroy(gam_result);
            }
        if(shim_verbose){
            t_rep_create += r1 - r0;
            if(gam_result){
                t_rep_submit += r2 - r1;
                t_rep_destroy += g_get_monotonic_time() - r2;
                }
            }
        for(k = 0; k < sp->round_total; k++)
            c4gpu_alignment_clear(&sp->round[k]);
        g_free(sp->round);
        GAM_destroy(sp->gam);
        Sequence_destroy(sp->query);
        Sequence_destroy(sp->target);
        g_free(sp);
        }
    g_ptr_array_free(job->todo, TRUE);
    if(shim_verbose)
        g_message("c4gpu: flush of %d pairs: %.0f ms flattening, %.0f ms on the device (upload, passes, read-back; slot %d, started %.0f ms after "
                  "the flush, the main thread waited %.0f ms for it), %.0f ms replaying through the reference "
                  "(GAM_Result_exhaustive_create %.0f, GAM_Result_submit %.0f, GAM_Result_destroy %.0f ms)", n,
                  (job->t_prepared - job->t_start) / 1e3, (job->t_dev1 - job->t_dev0) / 1e3, job->slot, (job->t_dev0 - job->t_prepared) / 1e3,
                  (t_rep0 - t_wait0) / 1e3, (g_get_monotonic_time() - t_rep0) / 1e3, t_rep_create / 1e3, t_rep_submit / 1e3,
                  t_rep_destroy / 1e3);
    g_free(job->error);
    g_free(job);
    return;
    }

/* everything collected so far is printed when this returns (what every caller that must keep the output order needs) */
static void shim_flush(void){
    register ShimFlushJob *job = shim_flush_prepare();
    if(job){
        if(shim_in_flight && shim_async())       /* the last batch goes to the device while the one before it is replayed */
            job->thread = g_thread_new("c4gpu-flush", shim_flush_device, job);
        else
            shim_flush_device(job);
        }
    if(shim_in_flight){
        shim_flush_replay(shim_in_flight);
        shim_in_flight = NULL;
        }
    if(job)
        shim_flush_replay(job);
    return;
    }

/* a full batch while more pairs are coming: its device part starts on a thread of its own; the batch before it (whose device
 * part has had the whole collection time of this one) is replayed now, beside it */
static gboolean shim_flush_threads_alive(void){
    return (shim_in_flight != NULL) || shim_sdp_busy();
    }

static void shim_flush_async(void){
    register ShimFlushJob *job, *before = shim_in_flight;
    if(!shim_async()){
        shim_flush();
        return;
        }
    if(!(job = shim_flush_prepare()))
        return;
    job->thread = g_thread_new("c4gpu-flush", shim_flush_device, job);
    shim_in_flight = job;
    if(before)
        shim_flush_replay(before);
    return;
    }

/* GAM_Result_add_alignment (gam.c:673) writes every match cell of an alignment into the pair's SubOpt range tree, whatever
 * --subopt says; with -S no nothing ever reads it (the tree only serves the next Optimal_find_path of the same pair).  While a
 * pair of a batch is replayed with use_subopt off, the call is dropped: ~1 000 tree insertions per chance alignment, 0.06 ms per
 * pair of the all-against-all run.  Every other caller (hpair.c:805, any run with -S yes) reaches the reference's function. */
extern void SubOpt_add_alignment_cpu(SubOpt *subopt, Alignment *alignment);
void SubOpt_add_alignment(SubOpt *subopt, Alignment *alignment){
    if(shim_replay_pair && (!shim_replay_pair->gam->gas->use_subopt) && !shim_env("C4GPU_KEEP_SUBOPT"))
        return;
    SubOpt_add_alignment_cpu(subopt, alignment);
    return;
    }

GAM_Result *GAM_Result_exhaustive_create(GAM *gam, Sequence *query, Sequence *target){
    register ShimPending *sp;
    if(shim_replay_pair || (!shim_can_batch(gam, query, target))){
        shim_flush();             /* keep the output order */
        return GAM_Result_exhaustive_create_cpu(gam, query, target);
        }
    if(shim_pending && shim_pending->len
    && (((ShimPending*)shim_pending->pdata[0])->gam != gam))
        shim_flush();
    if(!shim_pending){
        shim_pending = g_ptr_array_new();
        shim_mark("first pair of a batch collected");
        }
    sp = g_new0(ShimPending, 1);
    sp->gam = GAM_share(gam);
    sp->query = Sequence_share(query);
    sp->target = Sequence_share(target);
    g_ptr_array_add(shim_pending, sp);
    /* residues + codes + 4 int32 splice arrays + tn4 per target position; flush well inside the HBM */
    shim_pending_bytes += 22.0 * target->len + 2.0 * query->len;
    if(((gint)shim_pending->len >= shim_batch_size())
    || (shim_pending_bytes > (shim_env("C4GPU_BATCH_GB") ? atof(shim_env("C4GPU_BATCH_GB")) : 96.0) * 1e9))
        shim_flush_async();
    /* C4GPU_EAGER=1 (off by default): also whenever the device has nothing to do and an eighth of --gpubatch (at least 256 pairs)
     * has collected, so that the first pairs of a run go to the device while the front end is still reading the rest.  Measured
     * on the 64 x 64 run (profiles/r05_dropin_c4.log): 1.71 s at best against 1.73-1.87 s without, but 2.4-2.6 s in two runs of
     * three -- the small first batch sizes the launch buffers, and the full batches behind it grow them in the middle of a run */
    else if(shim_async() && (!g_atomic_int_get(&shim_device_busy))
         && ((gint)shim_pending->len >= MAX(256, shim_batch_size() / 8))
         && shim_env("C4GPU_EAGER") && (atoi(shim_env("C4GPU_EAGER")) != 0))
        shim_flush_async();
    return NULL;                  /* the result is submitted by the flush, in submission order */
    }

void GAM_report(GAM *gam){        /* analysis.c:1421: after the last pair */
    shim_mark("GAM_report: last flushes");
    shim_flush();
    shim_bsdp_flush();
    shim_sdp_flush();
    shim_bsdp_report();
    shim_sdp_report();
    shim_seed_report();
    shim_hsp_report();
    /* the start-up line of C4GPU_VERBOSE wants the device thread's last clock: wait for it (a run without the line leaves its
     * warm-up to shim_quiesce, which cuts it short) */
    if(shim_ctx_thread && shim_verbose){
        register gboolean mine;
        g_mutex_lock(&shim_ctx_lock);
        mine = !shim_joined;
        shim_joined = TRUE;
        g_mutex_unlock(&shim_ctx_lock);
        if(mine)
            g_thread_join(shim_ctx_thread);
        if(shim_verbose)
            g_message("c4gpu start-up: device opened on its own thread from %.0f to %.0f ms after process start, code objects loaded "
                      "by %.0f ms; the main thread waited %.0f ms for it; end of the run at %.0f ms", (shim_t_open - shim_t0) / 1e3,
                      (shim_t_ready - shim_t0) / 1e3, (shim_t_warm - shim_t0) / 1e3, shim_waited / 1e3,
                      (g_get_monotonic_time() - shim_t0) / 1e3);
        }
    GAM_report_cpu(gam);
    return;
    }

Alignment *Optimal_find_path(Optimal *optimal, Region *region, gpointer user_data,
                             C4_Score threshold, SubOpt *subopt){
    register ShimPending *sp = shim_replay_pair;
    register c4gpu_alignment *a;
    register Alignment *alignment;
    register Region *ar;
    register C4_Model *model = optimal->find_path->model;
    register gint k;
    if((!sp) && (alignment = shim_bsdp_find_path(optimal, region, subopt)))
        return alignment;                       /* a terminal / join path of the BSDP batch (c4gpu_bsdp.c) */
    if((!sp) || (shim_replay_call >= sp->round_total)
    || region->query_start || region->target_start
    || (region->query_length != (gint)sp->query->len) || (region->target_length != (gint)sp->target->len)){
        if(sp)
            shim_replay_call = G_MAXINT/2;     /* once off the recorded sequence, stay off it */
        return Optimal_find_path_cpu(optimal, region, user_data, threshold, subopt);
        }
    a = &sp->round[shim_replay_call++];
    if((!a->valid) || (a->score < threshold))
        return NULL;                            /* optimal.c:144-145,408-411 */
    ar = Region_create(a->region.query_start, a->region.target_start,
                       a->region.query_length, a->region.target_length);
    alignment = Alignment_create(model, ar, a->score);
    Region_destroy(ar);
    for(k = 0; k < a->n_ops; k++)
        Alignment_add(alignment, model->transition_list->pdata[a->op_transition[k]], a->op_length[k]);
    return alignment;
    }
