/* c4gpu_hsp.c — the seeding seam of the drop-in: the ungapped X-drop extensions of HSPset_seed_hsp
 * (src/comparison/hspset.c:933-997) of ALL word hits of a target scan in one device launch.  Third file of the
 * exonerate-gpu shim (integration/Makefile); INTEGRATION.md section 3b.
 *
 * The reference's word-neighbourhood FSM scan (seeder.c) reports word hits one by one to HSPset_seed_hsp, which tests
 * the diagonal's horizon, grows the HSP (trim, initial score, X-drop extension left and right) and stores it.  The
 * extension depends on the seed alone; only the horizon test, the threshold and the store order need the seeds in
 * sequence.  So the front of HSPset_seed_hsp only writes the seed down; at the first HSPset_finalise after a scan
 * (comparison.c:206-210) every pending seed of every HSPset goes to c4gpu_hsp_extend_batch, and each set is then
 * replayed in seed order: horizon test on the set's own horizon array (hspset.c:952-958), HSPset_add_known_hsp
 * (hspset.c:999: HSP_init + HSP_store, i.e. score, threshold, filter queues exactly as the reference) and the horizon
 * update.  The HSP lists that reach BSDP / SDP are the reference's, HSP for HSP.
 * Not taken: soft-masked sequences (--softmaskquery / --softmasktarget: the two-stage extension differs), --seedrepeat > 1, HSPsets without a horizon, match types other
 * than DNA / protein / protein-vs-DNA: those sets keep the reference's own function.
 * C4GPU_HSP_HOST=1 (CPU test of the seam): the extensions come from the reference's own HSPset_seed_hsp on scratch sets. */
#include <string.h>
#include <stdlib.h>

#include "hspset.h"
#include "match.h"
#include "sequence.h"

#include "c4gpu.h"
#include "c4gpu_shim.h"

extern void HSPset_seed_hsp_cpu(HSPset *hsp_set, guint query_start, guint target_start);
extern HSPset *HSPset_finalise_cpu(HSPset *hsp_set);

typedef struct { HSPset *set; GArray *seeds; /* guint pairs (query_start, target_start) */ } ShimSeedSet;

static GHashTable *hsp_pending = NULL;          /* HSPset* -> ShimSeedSet* */
static GPtrArray *hsp_order = NULL;             /* ShimSeedSet*, in order of first seed */
static struct { long sets, seeds, stored, flushes; double device_ms; } hst;

static gint hsp_match_kind(HSPset *hsp_set){
    switch(hsp_set->param->match->type){
        case Match_Type_DNA2DNA: return C4GPU_MATCH_DNA2DNA;
        case Match_Type_PROTEIN2PROTEIN: return C4GPU_MATCH_PROTEIN2PROTEIN;
        case Match_Type_PROTEIN2DNA: return C4GPU_MATCH_PROTEIN2DNA;
        default: return -1;
        }
    }

static gboolean hsp_eligible(HSPset *hsp_set){
    static gint off = -1;
    if(off < 0)
        off = shim_env("C4GPU_HSP_OFF") ? 1 : 0;
    if(off || (!hsp_set->horizon) || (hsp_set->param->seed_repeat > 1) || (hsp_match_kind(hsp_set) < 0))
        return FALSE;
    /* the mask functions are always set (match.c:670,679) and ask the sequence's alphabet: without --softmaskquery /
     * --softmasktarget nothing is ever masked, the masked first extension (hspset.c:975-983) is then the plain one, and a
     * second extension from its ends finds what the first one found */
    if(hsp_set->query->alphabet->is_soft_masked || hsp_set->target->alphabet->is_soft_masked)
        return FALSE;
    if(shim_batch_size() <= 0)
        return FALSE;
    /* a set the reference's own function has already stored an HSP in (its first word hits came while the device was still
     * being opened, shim_ctx_nowait) stays with that function: the replay starts from an empty set */
    if(!hsp_set->is_empty)
        return FALSE;
    return shim_env("C4GPU_HSP_HOST") || (shim_ctx_nowait() != NULL);
    }

void HSPset_seed_hsp(HSPset *hsp_set, guint query_start, guint target_start){
    register ShimSeedSet *ss;
    guint seed[2];
    if(shim_seed_recording(hsp_set, query_start, target_start))        /* C4GPU_SEED_CHECK: the reference's own walk, written down */
        return;
    ss = hsp_pending ? g_hash_table_lookup(hsp_pending, hsp_set) : NULL;
    if(!ss){
        if(!hsp_eligible(hsp_set)){
            HSPset_seed_hsp_cpu(hsp_set, query_start, target_start);
            return;
            }
        if(!hsp_pending){
            hsp_pending = g_hash_table_new(g_direct_hash, g_direct_equal);
            hsp_order = g_ptr_array_new();
            }
        ss = g_new0(ShimSeedSet, 1);
        ss->set = hsp_set;
        ss->seeds = g_array_new(FALSE, FALSE, 2 * sizeof(guint));
        g_hash_table_insert(hsp_pending, hsp_set, ss);
        g_ptr_array_add(hsp_order, ss);
        /* the seeder asks Comparison_has_hsps BEFORE it finalises (seeder.c:903-909): a set with pending word hits must
         * not look empty there; the replay sets the flag to what the reference would hold, and both consumers of a
         * reported comparison ask again (gam.c:740,1122) */
        hsp_set->is_empty = FALSE;
        }
    seed[0] = query_start; seed[1] = target_start;
    g_array_append_val(ss->seeds, seed);
    return;
    }

/* the replay of one set: hspset.c:936-958 (seed_repeat == 1) around the precomputed HSPs */
static void hsp_replay(ShimSeedSet *ss, const c4gpu_hsp *hsp){
    register HSPset *hsp_set = ss->set;
    register gint aq = hsp_set->param->match->query->advance, at = hsp_set->param->match->target->advance;
    register guint k;
    hsp_set->is_empty = TRUE;                                            /* HSP_store clears it with the first HSP kept */
    for(k = 0; k < ss->seeds->len; k++){
        register guint *seed = &g_array_index(ss->seeds, guint, 2 * k);
        register gint diag_pos = (seed[1] * aq) - (seed[0] * at);
        register gint query_frame = seed[0] % aq, target_frame = seed[1] % at;
        register gint section_pos = (diag_pos + hsp_set->query->len) % hsp_set->query->len;
        /* a seed more than a query length above the main diagonal (query_start * target advance > target_start + query
         * length: protein2dna hits at the very start of a target) has a NEGATIVE section_pos in the reference
         * (hspset.c:943-944, its g_assert compiled out): an out-of-bounds horizon entry there, none here */
        register gint *horizon = (section_pos >= 0) ? &hsp_set->horizon[0][section_pos][query_frame][target_frame] : NULL;
        if(horizon && ((gint)seed[1] < *horizon))
            continue;
        HSPset_add_known_hsp(hsp_set, hsp[k].query_start, hsp[k].target_start, hsp[k].length);
        if(horizon)
            *horizon = hsp[k].target_start + hsp[k].length * at;         /* HSP_target_end */
        hst.stored++;
        }
    return;
    }

static void hsp_flush(void){
    register guint i, k, n_sets = hsp_order ? hsp_order->len : 0, total = 0;
    register GHashTable *seq_index;
    register GPtrArray *strs;
    c4gpu_pair *pairs;
    c4gpu_hsp_seed *seeds;
    c4gpu_hsp *hsps;
    gint *set_pair, *set_first;
    gint32 *chain_of = NULL, *horizon0 = NULL, *gc = NULL;
    gint n_chains = 0;
    gboolean ok = TRUE;
    gint64 t0 = g_get_monotonic_time();
    if(!n_sets)
        return;
    for(i = 0; i < n_sets; i++)
        total += ((ShimSeedSet*)hsp_order->pdata[i])->seeds->len;
    pairs = g_new0(c4gpu_pair, n_sets);
    seeds = g_new(c4gpu_hsp_seed, total + 1);
    hsps = g_new0(c4gpu_hsp, total + 1);
    set_pair = g_new(gint, n_sets);
    set_first = g_new(gint, n_sets + 1);
    seq_index = g_hash_table_new(g_direct_hash, g_direct_equal);
    strs = g_ptr_array_new();
    for(i = 0, total = 0; i < n_sets; i++){
        register ShimSeedSet *ss = hsp_order->pdata[i];
        register gchar *qs, *ts;
        if(!(qs = g_hash_table_lookup(seq_index, ss->set->query))){
            qs = Sequence_get_str(ss->set->query);
            g_hash_table_insert(seq_index, ss->set->query, qs);
            g_ptr_array_add(strs, qs);
            }
        if(!(ts = g_hash_table_lookup(seq_index, ss->set->target))){
            ts = Sequence_get_str(ss->set->target);
            g_hash_table_insert(seq_index, ss->set->target, ts);
            g_ptr_array_add(strs, ts);
            }
        pairs[i].query = (const uint8_t*)qs;  pairs[i].query_len = ss->set->query->len;
        pairs[i].target = (const uint8_t*)ts; pairs[i].target_len = ss->set->target->len;
        set_first[i] = total;
        for(k = 0; k < ss->seeds->len; k++){
            seeds[total].pair = i;
            seeds[total].query_start = g_array_index(ss->seeds, guint, 2 * k);
            seeds[total].target_start = g_array_index(ss->seeds, guint, 2 * k + 1);
            total++;
            }
        }
    set_first[n_sets] = total;
    /* the horizon entry every seed is tested against (hspset.c:939-958): one chain per (set, section, frames); on the
     * device a chain's seeds are taken in order and the ones below the running horizon are not extended at all
     * (c4gpu_hsp_extend_chains): a long identical diagonal costs one extension instead of one per word hit.
     * C4GPU_HSP_CHAIN=0: every seed is extended and the replay alone applies the horizon */
    if(!shim_env("C4GPU_HSP_HOST") && !(shim_env("C4GPU_HSP_CHAIN") && !atoi(shim_env("C4GPU_HSP_CHAIN")))){
        register GArray *h0 = g_array_new(FALSE, FALSE, sizeof(gint32));
        chain_of = g_new(gint32, total + 1);
        for(i = 0; i < n_sets; i++){
            register ShimSeedSet *ss = hsp_order->pdata[i];
            register HSPset *set = ss->set;
            register gint aq = set->param->match->query->advance, at = set->param->match->target->advance;
            register gint qlen = set->query->len, slots = qlen * aq * at, x;
            register gint32 *slot = g_new(gint32, slots > 0 ? slots : 1);
            for(x = 0; x < slots; x++)
                slot[x] = -1;
            for(k = 0; k < ss->seeds->len; k++){
                register guint *seed = &g_array_index(ss->seeds, guint, 2 * k);
                register gint diag_pos = (seed[1] * aq) - (seed[0] * at);
                register gint qf = seed[0] % aq, tf = seed[1] % at;
                register gint section_pos = (diag_pos + qlen) % qlen;
                gint32 h;
                if(section_pos < 0){                          /* no horizon entry (see hsp_replay): never skipped */
                    h = G_MININT32;
                    chain_of[set_first[i] + k] = h0->len;
                    g_array_append_val(h0, h);
                    continue;
                    }
                x = (section_pos * aq + qf) * at + tf;
                if(slot[x] < 0){
                    slot[x] = h0->len;
                    h = set->horizon[0][section_pos][qf][tf];
                    g_array_append_val(h0, h);
                    }
                chain_of[set_first[i] + k] = slot[x];
                }
            g_free(slot);
            }
        n_chains = h0->len;
        horizon0 = (gint32*)g_array_free(h0, FALSE);
        }
    if(shim_env("C4GPU_HSP_HOST")){
        /* the reference's own extension, one scratch HSPset per seed (nothing in its way) */
        for(i = 0; i < n_sets; i++){
            register ShimSeedSet *ss = hsp_order->pdata[i];
            register C4_Score keep = ss->set->param->threshold;
            ss->set->param->threshold = -1;                  /* store whatever grows: the replay applies the threshold */
            for(k = 0; k < ss->seeds->len; k++){
                register HSPset *one = HSPset_create(ss->set->query, ss->set->target, ss->set->param);
                register HSP *h;
                HSPset_seed_hsp_cpu(one, seeds[set_first[i] + k].query_start, seeds[set_first[i] + k].target_start);
                HSPset_finalise_cpu(one);
                h = one->hsp_list->pdata[0];
                hsps[set_first[i] + k].query_start = h->query_start; hsps[set_first[i] + k].target_start = h->target_start;
                hsps[set_first[i] + k].length = h->length; hsps[set_first[i] + k].score = h->score;
                HSPset_destroy(one);
                }
            ss->set->param->threshold = keep;
            }
    } else {
        /* one launch per (match type, seed length, dropoff): in practice one or two per scan */
        gboolean *done = g_new0(gboolean, n_sets);
        c4gpu_params params;
        shim_hsp_params(&params);
        for(i = 0; ok && (i < n_sets); i++){
            register ShimSeedSet *si = hsp_order->pdata[i];
            register gint kind = hsp_match_kind(si->set), first = -1, count = 0;
            c4gpu_hsp_seed *gs;
            c4gpu_hsp *go;
            if(done[i])
                continue;
            gs = g_new(c4gpu_hsp_seed, total + 1);
            go = g_new(c4gpu_hsp, total + 1);
            gc = chain_of ? g_new(gint32, total + 1) : NULL;
            for(k = i; k < n_sets; k++){
                register ShimSeedSet *sk = hsp_order->pdata[k];
                if(done[k] || (hsp_match_kind(sk->set) != kind) || (sk->set->param->seedlen != si->set->param->seedlen)
                || (sk->set->param->dropoff != si->set->param->dropoff))
                    continue;
                done[k] = TRUE;
                memcpy(gs + count, seeds + set_first[k], sizeof(c4gpu_hsp_seed) * (set_first[k+1] - set_first[k]));
                if(gc)
                    memcpy(gc + count, chain_of + set_first[k], sizeof(gint32) * (set_first[k+1] - set_first[k]));
                count += set_first[k+1] - set_first[k];
                (void)first;
                }
            if(gc){
                if(c4gpu_hsp_extend_chains(shim_get_ctx(), &params, kind, pairs, n_sets, si->set->param->seedlen,
                                           si->set->param->dropoff, gs, count, gc, n_chains, horizon0, go) != 0)
                    ok = FALSE;
            } else if(c4gpu_hsp_extend_batch(shim_get_ctx(), &params, kind, pairs, n_sets, si->set->param->seedlen,
                                      si->set->param->dropoff, gs, count, go) != 0)
                ok = FALSE;
            for(k = i, count = 0; ok && (k < n_sets); k++){            /* scatter back in the same order */
                register ShimSeedSet *sk = hsp_order->pdata[k];
                if((hsp_match_kind(sk->set) != kind) || (sk->set->param->seedlen != si->set->param->seedlen)
                || (sk->set->param->dropoff != si->set->param->dropoff) || (k < i))
                    continue;
                memcpy(hsps + set_first[k], go + count, sizeof(c4gpu_hsp) * (set_first[k+1] - set_first[k]));
                count += set_first[k+1] - set_first[k];
                }
            g_free(gs); g_free(go); g_free(gc);
            }
        g_free(done);
        }
    for(i = 0; i < n_sets; i++){
        register ShimSeedSet *ss = hsp_order->pdata[i];
        if(ok){
            hsp_replay(ss, hsps + set_first[i]);
        } else {                                             /* the device refused (e.g. a residue outside the alphabet) */
            ss->set->is_empty = TRUE;
            for(k = 0; k < ss->seeds->len; k++)
                HSPset_seed_hsp_cpu(ss->set, g_array_index(ss->seeds, guint, 2 * k), g_array_index(ss->seeds, guint, 2 * k + 1));
            }
        hst.sets++; hst.seeds += ss->seeds->len;
        g_array_free(ss->seeds, TRUE);
        g_free(ss);
        }
    if(!ok)
        g_warning("c4gpu: %s -- HSP extensions of this scan on the CPU", c4gpu_last_error());
    g_ptr_array_set_size(hsp_order, 0);
    g_hash_table_remove_all(hsp_pending);
    for(i = 0; i < strs->len; i++)
        g_free(strs->pdata[i]);
    g_ptr_array_free(strs, TRUE);
    g_hash_table_destroy(seq_index);
    g_free(pairs); g_free(seeds); g_free(hsps); g_free(set_pair); g_free(set_first); g_free(chain_of); g_free(horizon0);
    hst.flushes++;
    hst.device_ms += (g_get_monotonic_time() - t0) / 1e3;
    return;
    }

HSPset *HSPset_finalise(HSPset *hsp_set){
    if(hsp_order && hsp_order->len)
        hsp_flush();                 /* every set of this scan at once: the others are finalised right after */
    return HSPset_finalise_cpu(hsp_set);
    }

void shim_hsp_report(void){
    if(shim_env("C4GPU_VERBOSE") && hst.sets)
        g_message("c4gpu hsp: %ld word hits of %ld HSP sets extended in %ld device batch(es) (%.0f ms incl. flattening and "
                  "replay), %ld HSPs passed their horizon", hst.seeds, hst.sets, hst.flushes, hst.device_ms, hst.stored);
    return;
    }
