/* c4gpu_sdp.c — the SDP seam of the exonerate-gpu shim (fourth file).
 *
 * The reference's default heuristic (--gappedextension yes) hands every candidate pair to GAM_Result_SDP_create
 * (src/hub/gam.c:852-890), whose loop asks SDP_Pair_next_path (src/sdp/sdp.c:743) for one alignment after the other;
 * the first call runs the two Scheduler passes (sdp.c:538-600), one pair at a time, on one core.  For both flavours of
 * SDP_create (sdp.c:322-366: bidirectional from the seeds for affine / protein2dna, BASELINE config 1; boundary + spans
 * for est2genome / protein2genome) this file
 *   1. lets the front of GAM_Result_heuristic_create (c4gpu_bsdp.c) COLLECT (gam, comparison) here instead;
 *   2. at a flush gives the HSPs of all collected pairs to c4gpu_sdp_batch: both passes of every pair in two launches,
 *      then the reference's single-pass loop over the seeds (sdp.c:776-795) inside the library;
 *   3. REPLAYS every pair in submission order through the reference's own GAM_Result_heuristic_create, with the front
 *      of SDP_Pair_next_path handing out the batch's alignments in order — and NULL as soon as the next one is below the
 *      threshold of that call (which --bestn / --percent raise between calls: gam.c:870).
 * Thresholds, --bestn bookkeeping, ryo / vulgar printing stay the reference's.  Not taken (the reference's own function
 * runs): --refine (GAM_Result_add_alignment then blocks the REFINED alignment's cells, gam.c:663-673), --singlepass no,
 * --geneseed, pairs whose HSP sets differ in their advances, pairs that fill all 16 alignment slots of the batch and
 * pairs the library reports as not served (n_out = -1: their traceback did not fit the device).  There is no size limit:
 * the device passes are sparse wavefronts that visit what the reference's scheduler visits (c4_sdp_wave.h).
 * C4GPU_SDP_HOST=1 replaces step 2 by the reference's own SDP on the host (tests of the seam without a device);
 * C4GPU_SDP_OFF=1 switches the seam off.
 */
#include <string.h>
#include <stdlib.h>

#include "gam.h"
#include "sdp.h"
#include "comparison.h"
#include "alignment.h"
#include "modeltype.h"
#include "hspset.h"

#include "c4gpu.h"
#include "c4gpu_shim.h"

extern GAM_Result *GAM_Result_heuristic_create_cpu(GAM *gam, Comparison *comparison);
extern Alignment *SDP_Pair_next_path_cpu(SDP_Pair *sdp_pair, C4_Score threshold);

#define SHIM_SDP_MAX 16

typedef struct {
    GAM *gam;
    Comparison *comparison;
    gboolean have;                 /* the batch holds this pair's alignments */
    gint n, served;
    c4gpu_alignment alns[SHIM_SDP_MAX];
} ShimSdpPending;

static GPtrArray *sdp_pending = NULL;
static gdouble sdp_pending_bytes = 0.0;
static gdouble sdp_budget_bytes(void){
    static gdouble budget = 0.0;
    if(budget <= 0.0){
        int64_t mem = 0;
        register c4gpu_ctx *ctx = shim_get_ctx();
        if(shim_env("C4GPU_SDP_GB"))
            budget = atof(shim_env("C4GPU_SDP_GB")) * 1e9;
        else if(ctx && (c4gpu_ctx_device_info(ctx, NULL, 0, NULL, &mem) == 0) && (mem > 0))
            budget = MIN(0.8 * (gdouble)mem, 120e9);
        else
            budget = 64e9;
        }
    return budget;
    }
static ShimSdpPending *sdp_cur = NULL;
static struct { long pairs, served_pairs, alignments, flushes, async_flushes; double device_ms, replay_ms, waited_ms; } sst;

static gboolean sdp_eligible(GAM *gam, Comparison *comparison){
    if((shim_batch_size() <= 0) || shim_env("C4GPU_SDP_OFF") || sdp_cur)
        return FALSE;
    if((!gam->gas->use_gapped_extension) || (!gam->sdp))
        return FALSE;
    if(!gam->sdp->sas->single_pass_subopt)
        return FALSE;
    if(gam->gas->refinement != GAM_Refinement_NONE)
        return FALSE;
    if(Comparison_Param_get_HSPSet_Argument_Set(comparison->param)->geneseed_threshold)
        return FALSE;
    if((!shim_env("C4GPU_SDP_HOST")) && (!shim_ctx_nowait()))
        return FALSE;
    return TRUE;
    }

/* the HSPs of a comparison in the order SDP_Pair_create_seed_list meets them (sdp.c:447-463); FALSE when its sets do
 * not share one pair of advances */
static gboolean sdp_gather_hsps(Comparison *comparison, GArray *hsps, gint *qa, gint *ta){
    register gint s, k;
    HSPset *sets[3];
    sets[0] = comparison->dna_hspset; sets[1] = comparison->protein_hspset; sets[2] = comparison->codon_hspset;
    *qa = *ta = 0;
    for(s = 0; s < 3; s++){
        if((!sets[s]) || (!sets[s]->hsp_list->len))
            continue;
        for(k = 0; k < (gint)sets[s]->hsp_list->len; k++){
            register HSP *h = sets[s]->hsp_list->pdata[k];
            c4gpu_hsp c;
            if(!*qa){
                *qa = HSP_query_advance(h); *ta = HSP_target_advance(h);
            } else if((*qa != HSP_query_advance(h)) || (*ta != HSP_target_advance(h))){
                return FALSE;
                }
            c.query_start = h->query_start; c.target_start = h->target_start;
            c.length = h->length; c.score = h->score; c.cobs = h->cobs;
            g_array_append_val(hsps, c);
            }
        }
    return hsps->len > 0;
    }

/* C4GPU_SDP_HOST: the alignments the reference's own SDP finds, in a private SubOpt as GAM_Result_add_alignment keeps */
static void sdp_host_pair(ShimSdpPending *p){
    register GAM *gam = p->gam;
    register gpointer ud = Model_Type_create_data(gam->gas->type, p->comparison->query, p->comparison->target);
    register SubOpt *so = SubOpt_create(p->comparison->query->len, p->comparison->target->len);
    register SDP_Pair *sdp_pair = SDP_Pair_create(gam->sdp, so, p->comparison, ud);
    register Alignment *a;
    register guint k;
    p->n = 0;
    while((p->n < SHIM_SDP_MAX) && (a = SDP_Pair_next_path_cpu(sdp_pair, gam->gas->threshold))){
        register c4gpu_alignment *c = &p->alns[p->n++];
        memset(c, 0, sizeof(*c));
        c->score = a->score;
        c->region.query_start = a->region->query_start;   c->region.target_start = a->region->target_start;
        c->region.query_length = a->region->query_length; c->region.target_length = a->region->target_length;
        c->n_ops = a->operation_list->len;
        c->op_transition = malloc(sizeof(int32_t) * (c->n_ops ? c->n_ops : 1));
        c->op_length = malloc(sizeof(int32_t) * (c->n_ops ? c->n_ops : 1));
        for(k = 0; k < a->operation_list->len; k++){
            register AlignmentOperation *ao = a->operation_list->pdata[k];
            c->op_transition[k] = ao->transition->id;
            c->op_length[k] = ao->length;
            }
        c->valid = 1;
        SubOpt_add_alignment(so, a);
        Alignment_destroy(a);
        }
    SDP_Pair_destroy(sdp_pair);
    SubOpt_destroy(so);
    Model_Type_destroy_data(gam->gas->type, ud);
    p->have = (p->n < SHIM_SDP_MAX);
    return;
    }

/* A flush in three parts, like the exhaustive seam's (c4gpu_shim.c): PREPARE on the main thread (everything that reads the
 * reference's objects: sequences flattened, HSPs gathered, model and scoring data), the DEVICE part -- one library call on
 * plain arrays -- on a thread of its own with a context and a stream of its own, LAND on the main thread (join, results into
 * the pending pairs, replay in submission order).  A flush that is cut in the middle of a run (the memory estimate) goes to
 * the device while the main thread carries on with the comparisons behind it -- reading tries, word scans, HSP extensions,
 * collecting the next batch -- and lands in front of the next flush or at the end: config 5's heuristic leg has ~0.6 s of such
 * work behind its first flush of ~0.44 s (profiles/r05_c5_cold.md).  One flight at a time (two arenas of 65 GB do not fit
 * beside the staged sequences everywhere); replays keep the submission order, and nothing else reports in between: whatever
 * takes the reference's own path flushes -- and with that lands -- first (c4gpu_bsdp.c).  C4GPU_SDP_ASYNC=0: every flush
 * synchronous on the main context, as before. */
typedef struct {
    GPtrArray *todo, *strings;
    GArray *hsps;
    c4gpu_pair *pair;
    int32_t *first, *n_out;
    c4gpu_alignment *out;
    gboolean *usable, ready, async;
    gint qa, ta, dropoff, rc;
    C4_Score threshold;
    c4gpu_model fm;
    c4gpu_params params;
    gchar *err;
    GThread *thread;
    gint64 t0, t_host;
} SdpFlight;

static SdpFlight *sdp_in_flight = NULL;
/* The flight thread's context (own stream).  It is opened, and its record arena taken (c4gpu_ctx_sdp_reserve), by a thread that
 * starts as soon as the pending pairs add up to a batch worth a flight (8 GB of estimated records): the first allocation of
 * tens of GB in a process takes 0.3 ms or -- after another process has just given memory back -- 1.7-3 s, and the arena is then
 * there, allocated once and kept, when the first flush is cut a second later.  Whoever uses sdp_ctx2 takes sdp_ctx2_lock. */
static c4gpu_ctx *sdp_ctx2 = NULL;
static GMutex sdp_ctx2_lock;
static GThread *sdp_reserve_thread = NULL;
static gdouble sdp_reserve_ms = 0.0;

static void sdp_ctx2_open(void){               /* (under sdp_ctx2_lock) */
    int64_t mem = 0;
    if(sdp_ctx2)
        return;
    sdp_ctx2 = c4gpu_ctx_create(shim_device_ordinal());
    if(sdp_ctx2 && (c4gpu_ctx_own_stream(sdp_ctx2) != 0)){
        c4gpu_ctx_destroy(sdp_ctx2);
        sdp_ctx2 = NULL;
        }
    if(sdp_ctx2 && (!(shim_env("C4GPU_SDP_RESERVE") && (atof(shim_env("C4GPU_SDP_RESERVE")) <= 0.0)))
    && (c4gpu_ctx_device_info(sdp_ctx2, NULL, 0, NULL, &mem) == 0)){
        /* what a flush cut at the memory estimate asks for (config 5: 67.7 GB of a 288 GB device); C4GPU_SDP_RESERVE=<GB>, 0: none */
        register gdouble want = shim_env("C4GPU_SDP_RESERVE") ? atof(shim_env("C4GPU_SDP_RESERVE")) * 1e9 : MIN(0.25 * (gdouble)mem, 70e9);
        (void)c4gpu_ctx_sdp_reserve(sdp_ctx2, (int64_t)want);
        }
    return;
    }

static gpointer sdp_reserve_run(gpointer data){
    gint64 t0 = g_get_monotonic_time();
    g_mutex_lock(&sdp_ctx2_lock);
    sdp_ctx2_open();
    g_mutex_unlock(&sdp_ctx2_lock);
    sdp_reserve_ms = (g_get_monotonic_time() - t0) / 1e3;
    return NULL;
    }

static gboolean sdp_async(void){
    static gint on = -1;
    if(on < 0)
        on = (shim_env("C4GPU_SDP_ASYNC") && (atoi(shim_env("C4GPU_SDP_ASYNC")) == 0)) ? 0 : 1;
    return on;
    }

static SdpFlight *sdp_flight_prepare(GPtrArray *todo){
    register guint i, n = todo->len;
    register ShimSdpPending *p = todo->pdata[0];
    register GAM *gam = p->gam;
    register GHashTable *flat = g_hash_table_new(g_direct_hash, g_direct_equal);
    register SdpFlight *f = g_new0(SdpFlight, 1);
    gpointer ud;
    f->t0 = g_get_monotonic_time();
    f->todo = todo;
    f->strings = g_ptr_array_new();
    f->hsps = g_array_new(FALSE, FALSE, sizeof(c4gpu_hsp));
    f->pair = g_new0(c4gpu_pair, n);
    f->first = g_new0(int32_t, n + 1);
    f->n_out = g_new0(int32_t, n);
    f->out = g_new0(c4gpu_alignment, (gsize)n * SHIM_SDP_MAX);
    f->usable = g_new0(gboolean, n);
    f->rc = -1;
    for(i = 0; i < n; i++){
        register gchar *qs, *ts;
        gint pqa, pta;
        register guint before = f->hsps->len;
        p = todo->pdata[i];
        if(!(qs = g_hash_table_lookup(flat, p->comparison->query))){
            qs = Sequence_get_str(p->comparison->query);
            g_hash_table_insert(flat, p->comparison->query, qs);
            g_ptr_array_add(f->strings, qs);
            }
        if(!(ts = g_hash_table_lookup(flat, p->comparison->target))){
            ts = Sequence_get_str(p->comparison->target);
            g_hash_table_insert(flat, p->comparison->target, ts);
            g_ptr_array_add(f->strings, ts);
            }
        f->pair[i].query = (const uint8_t*)qs;  f->pair[i].query_len = p->comparison->query->len;
        f->pair[i].target = (const uint8_t*)ts; f->pair[i].target_len = p->comparison->target->len;
        f->first[i] = before;
        f->usable[i] = sdp_gather_hsps(p->comparison, f->hsps, &pqa, &pta);
        if(f->usable[i] && f->qa && ((pqa != f->qa) || (pta != f->ta)))
            f->usable[i] = FALSE;
        if(f->usable[i]){
            f->qa = pqa; f->ta = pta;
        } else {
            g_array_set_size(f->hsps, before);          /* this pair brings no HSPs to the batch: nothing comes back */
            }
        }
    f->first[n] = f->hsps->len;
    p = todo->pdata[0];
    ud = Model_Type_create_data(gam->gas->type, p->comparison->query, p->comparison->target);
    if(f->qa && shim_flatten_any(gam->sdp->model, ud, &f->fm, FALSE)){
        shim_params(ud, &f->params);
        f->ready = TRUE;
        }
    Model_Type_destroy_data(gam->gas->type, ud);
    f->dropoff = gam->sdp->sas->dropoff;
    f->threshold = gam->gas->threshold;
    g_hash_table_destroy(flat);
    f->t_host = g_get_monotonic_time() - f->t0;
    return f;
    }

/* nothing of the reference's is touched here: plain arrays in, plain arrays out */
static gpointer sdp_flight_device(gpointer data){
    register SdpFlight *f = data;
    register c4gpu_ctx *ctx = NULL;
    if(!f->ready)
        return NULL;
    if(f->async){
        g_mutex_lock(&sdp_ctx2_lock);               /* (waits for the thread that opens it, if that is still at it) */
        sdp_ctx2_open();
        ctx = sdp_ctx2;
    } else {
        ctx = shim_get_ctx();
        }
    if(!ctx){
        f->err = g_strdup(c4gpu_last_error());
    } else {
        f->rc = c4gpu_sdp_batch(ctx, &f->fm, &f->params, f->pair, f->todo->len, (const c4gpu_hsp*)f->hsps->data, f->first,
                                f->qa, f->ta, f->dropoff, f->threshold, SHIM_SDP_MAX, f->out, f->n_out);
        if(f->rc != 0)
            f->err = g_strdup(c4gpu_last_error());      /* the error string is the calling thread's */
        }
    if(f->async)
        g_mutex_unlock(&sdp_ctx2_lock);
    return NULL;
    }

static void sdp_replay(GPtrArray *todo){
    register guint i;
    register gint k;
    for(i = 0; i < todo->len; i++){                       /* replay in submission order */
        register ShimSdpPending *p = todo->pdata[i];
        register GAM_Result *gam_result;
        sdp_cur = p;
        gam_result = GAM_Result_heuristic_create_cpu(p->gam, p->comparison);
        sdp_cur = NULL;
        if(gam_result){
            GAM_Result_submit(gam_result);
            GAM_Result_destroy(gam_result);
            }
        sst.pairs++;
        if(p->have){
            sst.served_pairs++;
            sst.alignments += p->served;
            }
        for(k = 0; k < p->n; k++)
            c4gpu_alignment_clear(&p->alns[k]);
        Comparison_destroy(p->comparison);
        GAM_destroy(p->gam);
        g_free(p);
        }
    g_ptr_array_free(todo, TRUE);
    return;
    }

static void sdp_flight_land(SdpFlight *f){
    register guint i, n = f->todo->len;
    gint64 t0 = g_get_monotonic_time(), t1;
    if(f->thread){
        g_thread_join(f->thread);
        sst.waited_ms += (g_get_monotonic_time() - t0) / 1e3;
        }
    if(f->ready && (f->rc == 0)){
        for(i = 0; i < n; i++){
            register gint k;
            register ShimSdpPending *p = f->todo->pdata[i];
            p->n = (f->n_out[i] < 0) ? 0 : f->n_out[i];     /* -1: the device could not serve this pair (memory): CPU */
            for(k = 0; k < p->n; k++)
                p->alns[k] = f->out[(gsize)i * SHIM_SDP_MAX + k];
            p->have = f->usable[i] && (f->n_out[i] >= 0) && (f->n_out[i] < SHIM_SDP_MAX);
            }
    } else if(f->ready){
        g_warning("c4gpu: %s -- SDP stays on the CPU for this batch", f->err ? f->err : "SDP batch failed");
        }
    for(i = 0; i < f->strings->len; i++)
        g_free(f->strings->pdata[i]);
    g_ptr_array_free(f->strings, TRUE);
    g_array_free(f->hsps, TRUE);
    g_free(f->pair); g_free(f->first); g_free(f->n_out); g_free(f->out); g_free(f->usable); g_free(f->err);
    t1 = g_get_monotonic_time();
    shim_mark("sdp flush: replay");
    sdp_replay(f->todo);
    sst.device_ms += (f->async ? f->t_host + (t1 - t0) : (t1 - f->t0)) / 1e3;
    sst.replay_ms += (g_get_monotonic_time() - t1) / 1e3;
    g_free(f);
    return;
    }

/* lands what is in flight, then takes the pending pairs to the device: beside the main thread (`async`, left in flight) or
 * on it (landed before the return) */
static void sdp_flush_pending(gboolean async){
    register GPtrArray *todo = sdp_pending;
    register SdpFlight *f = NULL;
    if(todo && todo->len){
        sdp_pending = NULL;
        sdp_pending_bytes = 0.0;
        sst.flushes++;
        shim_mark("sdp flush: batch");
        if(shim_env("C4GPU_SDP_HOST")){
            register guint i;
            gint64 t0 = g_get_monotonic_time(), t1;
            if(sdp_in_flight){
                sdp_flight_land(sdp_in_flight);
                sdp_in_flight = NULL;
                }
            for(i = 0; i < todo->len; i++)
                sdp_host_pair(todo->pdata[i]);
            t1 = g_get_monotonic_time();
            shim_mark("sdp flush: replay");
            sdp_replay(todo);
            sst.device_ms += (t1 - t0) / 1e3;
            sst.replay_ms += (g_get_monotonic_time() - t1) / 1e3;
            return;
            }
        f = sdp_flight_prepare(todo);
        }
    /* (the last flush of a run that already has the side context goes there too: its arena is waiting) */
    if(f && sdp_async() && (async || sdp_in_flight || sdp_reserve_thread || sst.async_flushes)){
        /* its device part starts now; the flight before it (if any) is landed -- joined, replayed -- beside it.  The two
         * never share the device thread's context: the one before ran on sdp_ctx2 as well, and has to be JOINED first */
        if(sdp_in_flight && sdp_in_flight->thread){
            gint64 w0 = g_get_monotonic_time();
            g_thread_join(sdp_in_flight->thread);
            sdp_in_flight->thread = NULL;
            sst.waited_ms += (g_get_monotonic_time() - w0) / 1e3;
            }
        f->async = TRUE;
        f->thread = g_thread_new("c4gpu-sdp", sdp_flight_device, f);
        sst.async_flushes++;
        }
    if(sdp_in_flight){
        sdp_flight_land(sdp_in_flight);
        sdp_in_flight = NULL;
        }
    if(f){
        if(f->thread && async){
            sdp_in_flight = f;
        } else {
            if(!f->thread)
                sdp_flight_device(f);
            sdp_flight_land(f);
            }
        }
    return;
    }

void shim_sdp_flush(void){
    sdp_flush_pending(FALSE);
    if(sdp_reserve_thread){                    /* nothing of ours is left inside the runtime when the caller moves on */
        g_thread_join(sdp_reserve_thread);
        sdp_reserve_thread = NULL;
        }
    return;
    }

/* called by the front of GAM_Result_heuristic_create (c4gpu_bsdp.c): TRUE = the pair was taken, its result is submitted
 * by the flush */
gboolean shim_sdp_collect(GAM *gam, Comparison *comparison){
    register ShimSdpPending *p;
    if(!sdp_eligible(gam, comparison))
        return FALSE;
    if(!Comparison_has_hsps(comparison))                   /* gam.c:1122 */
        return TRUE;
    if(sdp_pending && sdp_pending->len && (((ShimSdpPending*)sdp_pending->pdata[0])->gam != gam))
        shim_sdp_flush();
    if(!sdp_pending)
        sdp_pending = g_ptr_array_new();
    p = g_new0(ShimSdpPending, 1);
    p->gam = GAM_share(gam);
    p->comparison = Comparison_share(comparison);
    g_ptr_array_add(sdp_pending, p);
    /* the device passes visit, like the reference's scheduler, only the cells inside the X-drop and keep their traceback
     * in an arena of 64 KB chunks: measured 7 chunks per seed + 0.25 per HSP position (both flavours, c4_sdp_dev.inc; config
     * 5's heuristic leg takes 0.99 of that).  A launch lasts as long as its longest pair -- a serial chain on one wave -- so
     * the pairs of a run belong in as FEW flushes as the device's memory holds (round 4: 512 pairs of config 5 in four
     * flushes of 64 GB were four times 350 ms of passes): a flush is cut when the estimate (x 1.15) plus the staged residues
     * (~20 bytes per residue with codes and splice arrays) pass C4GPU_SDP_GB.  Default: 0.8 of the device's memory, at most
     * 120 GB -- this estimate counts every HSP where the library's arena counts the seeds it runs (about half), and ONE
     * allocation of 113 GB takes hipMalloc 1.6 s where 65 GB take 0.3 ms (measured, profiles/r04_c5_breakdown.md): config 5's
     * 512 pairs go in two flushes of 65 + 48 GB (2 x 370 ms of passes) instead of one (385 ms of passes behind 1.6 s of hipMalloc) */
    {
        register GArray *hsps = g_array_new(FALSE, FALSE, sizeof(c4gpu_hsp));
        gint qa, ta;
        register guint k;
        register gdouble positions = 0.0;
        if(sdp_gather_hsps(comparison, hsps, &qa, &ta)){
            for(k = 0; k < hsps->len; k++)
                positions += g_array_index(hsps, c4gpu_hsp, k).length;
            sdp_pending_bytes += 1.15 * 65536.0 * (7.0 * hsps->len + 0.25 * positions);
            }
        g_array_free(hsps, TRUE);
    }
    sdp_pending_bytes += 20.0 * (comparison->query->len + comparison->target->len);
    if(sdp_async() && (!sdp_reserve_thread) && (!sdp_ctx2) && (sdp_pending_bytes > 8e9) && (!shim_env("C4GPU_SDP_HOST")))
        sdp_reserve_thread = g_thread_new("c4gpu-sdp-arena", sdp_reserve_run, NULL);
    if(((gint)sdp_pending->len >= shim_batch_size())
    || (sdp_pending_bytes > sdp_budget_bytes()))
        sdp_flush_pending(TRUE);          /* more comparisons are coming: this batch runs beside them */
    return TRUE;
    }

gboolean shim_sdp_busy(void){                 /* a thread of this seam may be inside the runtime */
    return (sdp_in_flight != NULL) || (sdp_reserve_thread != NULL);
    }

gboolean shim_sdp_replaying(void){
    return sdp_cur != NULL;
    }

/* sdp.c:743 — in a replay the batch's alignments, in order */
Alignment *SDP_Pair_next_path(SDP_Pair *sdp_pair, C4_Score threshold){
    register ShimSdpPending *p = sdp_cur;
    register c4gpu_alignment *c;
    register Alignment *alignment;
    register Region *region;
    register gint k;
    if((!p) || (!p->have))
        return SDP_Pair_next_path_cpu(sdp_pair, threshold);
    if(p->served >= p->n)
        return NULL;
    c = &p->alns[p->served];
    if(c->score < threshold)                               /* sdp.c:780-781: seeds come in score order */
        return NULL;
    p->served++;
    region = Region_create(c->region.query_start, c->region.target_start, c->region.query_length, c->region.target_length);
    alignment = Alignment_create(sdp_pair->sdp->model, region, c->score);
    for(k = 0; k < c->n_ops; k++)
        Alignment_add(alignment, sdp_pair->sdp->model->transition_list->pdata[c->op_transition[k]], c->op_length[k]);
    Region_destroy(region);
    return alignment;
    }

void shim_sdp_report(void){
    if(shim_env("C4GPU_VERBOSE") && sst.pairs)
        g_message("c4gpu sdp: %ld pairs in %ld flush(es): %ld served from device batches (%ld alignments); batches %.0f ms, "
                  "replay %.0f ms; %ld flush(es) beside the main thread, which waited %.0f ms for them (their context and arena were "
                  "ready after %.0f ms on a thread of their own)", sst.pairs, sst.flushes,
                  sst.served_pairs, sst.alignments, sst.device_ms, sst.replay_ms, sst.async_flushes, sst.waited_ms, sdp_reserve_ms);
    return;
    }
