/* c4gpu_seed.c — the word-scan seam of the drop-in (fifth file of the exonerate-gpu shim).
 *
 * The reference's Seeder_add_target (src/comparison/seeder.c:852-915) walks every target — for translated matches its
 * three translated frames — through the automaton the queries built (FSM_traverse src/struct/fsm.c:186-198, or
 * Seeder_VFSM_traverse_single seeder.c:698-720) on one host core and calls Seeder_FSM_traverse_func (seeder.c:649-695) at
 * every word end, which hands each seed of the word and of its neighbours to HSPset_seed_hsp.  This front of
 * Seeder_add_target (seeder.o keeps its own under Seeder_add_target_cpu) does the same with the walk on the device:
 *   1. ONCE per seeder, before the automaton is compiled: the words the queries put into it, read off the trie
 *      (FSM, pre-compile: fsm.c:112-135) or the leaf table (VFSM), each with its emission list in the reference's order —
 *      the word's own seeds, then the seeds of each neighbour (seeder.c:676-692) — go into a c4gpu_wordtab; the automaton
 *      is then compiled as Seeder_prepare would (seeder.c:779-784), so the reference's own walk stays usable;
 *   2. per target / frame: the string the reference would walk (Sequence_mask, Sequence_translate: its own functions),
 *      mapped through the automaton's traversal filter to column numbers, goes to c4gpu_seed_scan; the hits come back in
 *      walk order and are handed to HSPset_seed_hsp exactly as Seeder_WordInfo_seed does (seeder.c:624-647) — whose
 *      front (c4gpu_hsp.c) batches the extensions on the device;
 *   3. the report loop of Seeder_add_target (seeder.c:899-913) follows unchanged.
 * Building the automaton (Seeder_add_query, word neighbourhoods) stays the reference's.  Not taken (the reference's own
 * function runs): --saturatethreshold (a per-word running count in walk order), --wordambiguity > 1, seeders with both
 * a DNA and a codon loader, a seeder whose automaton was compiled before this front saw it.
 * C4GPU_SEED_OFF=1 switches the seam off; C4GPU_SEED_HOST=1 replaces the device scan by a dictionary scan on the host
 * (CPU test of steps 1 and 3); C4GPU_SEED_CHECK=1 also runs the reference's own walk first and compares the two hit lists
 * seed for seed (abort on the first difference).
 */
#include <string.h>
#include <stdlib.h>
#include <ctype.h>

#include "seeder.h"
#include "comparison.h"
#include "hspset.h"
#include "sequence.h"
#include "fsm.h"
#include "vfsm.h"

#include "c4gpu.h"
#include "c4gpu_shim.h"

extern void Seeder_add_target_cpu(Seeder *seeder, Sequence *target);
extern void Seeder_destroy_cpu(Seeder *seeder);

typedef struct { Seeder_QueryInfo *query_info; Seeder_Loader *loader; gint query_pos; } ShimEmit;
typedef struct { Sequence *query; HSP_Param *param; guint query_pos, target_pos; } ShimSeedRec;

typedef struct {
    Seeder *seeder;
    gboolean usable, hopeless;           /* table built / this seeder keeps the reference's walk */
    gint64 walked;                        /* symbols the reference's walk has seen while the table was not worth building */
    gint width, wordlen;
    guchar column[ALPHABETSIZE];          /* residue -> automaton column (0: outside the alphabet) */
    gboolean upper;                       /* the VFSM walk upper-cases (seeder.c:704) */
    GArray *codes, *first, *emits;        /* guint64 per word; gint32 n_words + 1; ShimEmit */
    GHashTable *host_index;               /* C4GPU_SEED_HOST: code -> word number + 1 */
    c4gpu_wordtab *tab;
} ShimSeedTab;

static GHashTable *seed_tabs = NULL;      /* Seeder* -> ShimSeedTab* */
static GArray *seed_record = NULL;        /* C4GPU_SEED_CHECK: what the reference's own walk hands to HSPset_seed_hsp */
static struct { long targets, cpu_targets, scans, symbols, hits, checked, words, emits; double scan_ms, host_ms, table_ms; } sdst;

gboolean shim_seed_recording(HSPset *hsp_set, guint query_start, guint target_start){
    ShimSeedRec r;
    if(!seed_record)
        return FALSE;
    r.query = hsp_set->query; r.param = hsp_set->param; r.query_pos = query_start; r.target_pos = target_start;
    g_array_append_val(seed_record, r);
    return TRUE;
    }

static void seed_emit_word(ShimSeedTab *st, Seeder_WordInfo *word_info){
    register Seeder_Seed *seed;
    register Seeder_Neighbour *neighbour;
    ShimEmit e;
    for(seed = word_info->seed_list; seed; seed = seed->next){
        e.query_info = seed->context->query_info; e.loader = seed->context->loader; e.query_pos = seed->query_pos;
        g_array_append_val(st->emits, e);
        }
    for(neighbour = word_info->neighbour_list; neighbour; neighbour = neighbour->next)
        for(seed = neighbour->word_info->seed_list; seed; seed = seed->next){
            e.query_info = seed->context->query_info; e.loader = seed->context->loader; e.query_pos = seed->query_pos;
            g_array_append_val(st->emits, e);
            }
    return;
    }

static void seed_add_word(ShimSeedTab *st, guint64 code, Seeder_WordInfo *word_info){
    gint32 n;
    g_array_append_val(st->codes, code);
    seed_emit_word(st, word_info);
    n = st->emits->len;
    g_array_append_val(st->first, n);
    return;
    }

/* The last level of the walk: every word's code and the seeds it emits (its own, then those of its neighbours: seed_emit_word).
 * Three dependent loads into cold memory per word -- 0.23 us each, 130 ms for the 570 000 words of a 128-protein seeder on one
 * thread.  The nodes are read-only here and the words of different nodes are independent: slices of the node list go to a few
 * threads, each with a table of its own, and the tables are joined in slice order (so the numbering is that of the one-thread
 * walk).  C4GPU_SEED_THREADS=1: one thread. */
typedef struct { FSM_Node *node; guint64 code; } ShimTrieItem;
typedef struct { ShimSeedTab part; FSM *f; ShimTrieItem *item; guint n; } ShimWordSlice;
static void seed_read_slice(ShimSeedTab *st, FSM *f, ShimTrieItem *item, guint n){
    register guint k;
    register gint c;
    for(k = 0; k < n; k++)
        for(c = 1; c < f->width; c++)
            if(item[k].node[c].data)
                seed_add_word(st, item[k].code * f->width + c, item[k].node[c].data);
    return;
    }
static gpointer seed_read_slice_thread(gpointer data){
    register ShimWordSlice *sl = data;
    seed_read_slice(&sl->part, sl->f, sl->item, sl->n);
    return NULL;
    }
static void seed_read_words(ShimSeedTab *st, FSM *f, GArray *words){
    register gint n_threads = shim_env("C4GPU_SEED_THREADS") ? atoi(shim_env("C4GPU_SEED_THREADS")) : (gint)MIN(8, g_get_num_processors());
    register gint t;
    register guint k;
    ShimWordSlice *sl;
    GThread **th;
    if((n_threads < 2) || (words->len < (shim_env("C4GPU_SEED_THREADS") ? (guint)n_threads : 4096u))){     /* (asked for: any size) */
        seed_read_slice(st, f, (ShimTrieItem*)words->data, words->len);
        return;
        }
    sl = g_new0(ShimWordSlice, n_threads);
    th = g_new0(GThread*, n_threads);
    for(t = 0; t < n_threads; t++){
        register guint lo = (guint)(((guint64)words->len * t) / n_threads), hi = (guint)(((guint64)words->len * (t + 1)) / n_threads);
        gint32 zero = 0;
        sl[t].f = f;
        sl[t].item = ((ShimTrieItem*)words->data) + lo;
        sl[t].n = hi - lo;
        sl[t].part.codes = g_array_new(FALSE, FALSE, sizeof(guint64));
        sl[t].part.first = g_array_new(FALSE, FALSE, sizeof(gint32));
        sl[t].part.emits = g_array_new(FALSE, FALSE, sizeof(ShimEmit));
        g_array_append_val(sl[t].part.first, zero);
        th[t] = g_thread_new("c4gpu-words", seed_read_slice_thread, &sl[t]);
        }
    for(t = 0; t < n_threads; t++){
        register gint32 base = st->emits->len;
        g_thread_join(th[t]);
        g_array_append_vals(st->codes, sl[t].part.codes->data, sl[t].part.codes->len);
        g_array_append_vals(st->emits, sl[t].part.emits->data, sl[t].part.emits->len);
        for(k = 1; k < sl[t].part.first->len; k++){          /* (entry 0 is the slice's own leading zero) */
            gint32 v = base + g_array_index(sl[t].part.first, gint32, k);
            g_array_append_val(st->first, v);
            }
        g_array_free(sl[t].part.codes, TRUE);
        g_array_free(sl[t].part.first, TRUE);
        g_array_free(sl[t].part.emits, TRUE);
        }
    g_free(sl);
    g_free(th);
    return;
    }

/* The words of the automaton, level by level.  Before FSM_compile (fsm.c:136-184) `next` is NULL where the trie has no
 * child; after it every `next` is set, but a failure link leads to a node no deeper than its origin, so in a level-order
 * walk the edges into nodes not seen before are exactly the trie's own: the walk reads both forms.  Column c of a node at
 * the last level holds the word's data. */
static void seed_walk_trie(ShimSeedTab *st, FSM *f){
    register GArray *level = g_array_new(FALSE, FALSE, sizeof(ShimTrieItem)), *next_level, *words;
    /* (before FSM_compile every non-NULL `next` is a trie edge: no need to remember the nodes seen -- a hash insertion per node
     * was half of the 130 ms a 256-protein seeder's 1.7 million words took to read) */
    register GHashTable *seen = f->is_compiled ? g_hash_table_new(g_direct_hash, g_direct_equal) : NULL;
    register gint depth, c;
    register guint k;
    ShimTrieItem it;
    it.node = f->root; it.code = 0;
    g_array_append_val(level, it);
    if(seen)
        g_hash_table_insert(seen, f->root, f->root);
    for(depth = 0; depth + 1 < st->wordlen; depth++){
        next_level = g_array_new(FALSE, FALSE, sizeof(ShimTrieItem));
        for(k = 0; k < level->len; k++){
            register ShimTrieItem *p = &g_array_index(level, ShimTrieItem, k);
            for(c = 1; c < f->width; c++){
                register FSM_Node *child = p->node[c].next;
                if(child && ((!seen) || !g_hash_table_lookup(seen, child))){
                    if(seen)
                        g_hash_table_insert(seen, child, child);
                    it.node = child; it.code = p->code * f->width + c;
                    g_array_append_val(next_level, it);
                    }
                }
            }
        g_array_free(level, TRUE);
        level = next_level;
        }
    if(seen)
        g_hash_table_destroy(seen);
    /* the reference's traversal meets the words in no particular order; ours does not depend on it either (hash table) */
    words = level;
    seed_read_words(st, f, words);
    g_array_free(words, TRUE);
    return;
    }

/* Is the table worth building yet?  Reading the words off the automaton costs about as much as the reference's walk over
 * a few symbols per trie node; until the targets of this seeder add up to that, the reference's own walk serves them (a
 * 100-protein seeder with word neighbourhoods against one 10 kaa target is walked in a millisecond and its 660 000 words
 * would take 270 ms to read).  The factor was 16 until round 4 (0.5 us per node against 20-50 ns per walked symbol, measured
 * on short targets); config 5's heuristic leg measured it again on a 10 Mb chromosome (profiles/r04_c5_breakdown.md): the
 * reference's three-frame walk of 10 Mb is 700-800 ms, reading a 256-protein seeder's words 200 ms and scanning the
 * chromosome on the device 100 ms with delivery -- break-even at 2.7 symbols per node, and with 16 the first strand of every
 * query chunk was still walked on the CPU.  Now 4 (C4GPU_SEED_FACTOR). */
static gboolean seed_worth_it(ShimSeedTab *st, Sequence *target){
    register Seeder *seeder = st->seeder;
    register gdouble nodes = seeder->seeder_fsm ? (gdouble)seeder->seeder_fsm->fsm->chunk_count
                                                : (gdouble)seeder->seeder_vfsm->vfsm->lrw / 64.0;
    register gdouble factor = shim_env("C4GPU_SEED_FACTOR") ? atof(shim_env("C4GPU_SEED_FACTOR")) : 4.0;
    return (gdouble)(st->walked + target->len) >= factor * nodes;
    }

static ShimSeedTab *seed_state(Seeder *seeder){
    register ShimSeedTab *st;
    if(!seed_tabs)
        seed_tabs = g_hash_table_new(g_direct_hash, g_direct_equal);
    if((st = g_hash_table_lookup(seed_tabs, seeder)))
        return st;
    st = g_new0(ShimSeedTab, 1);
    st->seeder = seeder;
    g_hash_table_insert(seed_tabs, seeder, st);
    if(seeder->saturate_threshold || (seeder->sas->word_ambiguity > 1) || (seeder->dna_loader && seeder->codon_loader))
        st->hopeless = TRUE;
    return st;
    }

static gboolean seed_build(ShimSeedTab *st){
    register Seeder *seeder = st->seeder;
    register gint i;
    gint32 zero = 0;
    gint64 t_begin = g_get_monotonic_time();
    st->wordlen = seeder->any_hsp_param->wordlen;
    st->codes = g_array_new(FALSE, FALSE, sizeof(guint64));
    st->first = g_array_new(FALSE, FALSE, sizeof(gint32));
    st->emits = g_array_new(FALSE, FALSE, sizeof(ShimEmit));
    g_array_append_val(st->first, zero);
    if(seeder->seeder_fsm){
        register FSM *f = seeder->seeder_fsm->fsm;
        st->width = f->width;
        for(i = 0; i < ALPHABETSIZE; i++)
            st->column[i] = f->traversal_filter[i];
        st->column[0] = 0;
        seed_walk_trie(st, f);
        shim_mark("  trie read");
        /* FSM_compile (Seeder_prepare, seeder.c:783-788) is left to the reference's own walk, which prepares the seeder when it is
         * first asked (seeder.c:865) -- that is only where a scan could not be served.  The failure links are of no use to the
         * word table, and compiling a 256-protein seeder's automaton took as long as reading its words (C4GPU_SEED_COMPILE=1:
         * compile here, as rounds 3-4 did) */
        if((!f->is_compiled) && shim_env("C4GPU_SEED_COMPILE")){
            FSM_compile(f);
            seeder->is_prepared = TRUE;
            shim_mark("  automaton compiled");
            }
    } else {
        register VFSM *vfsm = seeder->seeder_vfsm->vfsm;
        register VFSM_Int leaf;
        register gchar *word = g_new0(gchar, vfsm->depth + 1);
        st->width = vfsm->alphabet_size + 1;
        st->upper = TRUE;
        for(i = 0; i < ALPHABETSIZE; i++)
            st->column[i] = (guchar)vfsm->index[toupper(i)];
        st->column[0] = 0;
        if((gint)vfsm->depth != st->wordlen){
            g_free(word);
            return FALSE;
            }
        for(leaf = 0; leaf < vfsm->lrw; leaf++){
            register Seeder_WordInfo *word_info = seeder->seeder_vfsm->leaf[leaf];
            register guint64 code = 0;
            if(!word_info)
                continue;
            VFSM_state2word(vfsm, VFSM_leaf2state(vfsm, leaf), word);
            for(i = 0; i < st->wordlen; i++)
                code = code * st->width + (guchar)vfsm->index[(guchar)word[i]];
            seed_add_word(st, code, word_info);
            }
        g_free(word);
        }
    if(!seeder->seeder_fsm)
        seeder->is_prepared = TRUE;                            /* (nothing to compile: Seeder_prepare only sets the flag) */
    if(shim_env("C4GPU_SEED_HOST")){
        st->host_index = g_hash_table_new(g_int64_hash, g_int64_equal);
        for(i = 0; i < (gint)st->codes->len; i++)
            g_hash_table_insert(st->host_index, &g_array_index(st->codes, guint64, i), GINT_TO_POINTER(i + 1));
    } else {
        st->tab = c4gpu_wordtab_create(shim_get_ctx(), st->width, st->wordlen, (const uint64_t*)st->codes->data,
                                       (const int32_t*)st->first->data, st->codes->len);
        shim_mark("  word table on the device");
        if(!st->tab){
            g_warning("c4gpu: %s -- the word scan stays on the CPU", c4gpu_last_error());
            return FALSE;
            }
        }
    st->usable = TRUE;
    sdst.words += st->codes->len; sdst.emits += st->emits->len;
    sdst.table_ms += (g_get_monotonic_time() - t_begin) / 1e3;
    return TRUE;
    }

/* Seeder_WordInfo_seed, seeder.c:624-647 (no saturation threshold here) */
static void seed_deliver(Seeder *seeder, Sequence *target, ShimEmit *e, gint target_pos, GArray *mine){
    register Seeder_QueryInfo *query_info = e->query_info;
    register HSPset *hspset;
    seeder->comparison_count++;
    if(!query_info->curr_comparison){
        query_info->curr_comparison = Comparison_create(seeder->comparison_param, query_info->query, target);
        g_ptr_array_add(seeder->active_queryinfo_list, query_info);
        }
    hspset = *(HSPset**)((gchar*)query_info->curr_comparison + e->loader->hspset_offset);   /* OFFSET_ITEM, seeder.c:644 */
    if(mine){
        ShimSeedRec r;
        r.query = hspset->query; r.param = hspset->param; r.query_pos = e->query_pos; r.target_pos = target_pos;
        g_array_append_val(mine, r);
        }
    HSPset_seed_hsp(hspset, e->query_pos, target_pos);
    return;
    }

/* one walk: `seq` as the reference would hand it to FSM_traverse, frame 0 (untranslated) or 1 .. 3.  The hits of a frame are
 * only written down here; Seeder_add_target delivers them once every frame of the target has been scanned, so that a scan
 * that fails (no device memory for the hit buffer, a HIP error) leaves nothing half done and the target goes to the
 * reference's own walk */
typedef struct {
    c4gpu_word_hit *hits;
    gint64 n_hits;
    gint frame;
    } ShimFrameHits;

static gboolean seed_scan_string(ShimSeedTab *st, gchar *seq, gint frame, ShimFrameHits *out){
    register gint n = strlen(seq), i;
    register guchar *sym = g_new(guchar, n + 1);
    register c4gpu_word_hit *hits = NULL;
    int64_t n_hits = 0, cap;
    gint64 t0 = g_get_monotonic_time();
    for(i = 0; i < n; i++)
        sym[i] = st->column[(guchar)seq[i]];
    if(st->host_index){
        register GArray *h = g_array_new(FALSE, FALSE, sizeof(c4gpu_word_hit));
        for(i = st->wordlen - 1; i < n; i++){
            guint64 code = 0;
            register gint w, word;
            gboolean ok = TRUE;
            for(w = 0; w < st->wordlen; w++){
                register guchar s = sym[i - st->wordlen + 1 + w];
                if(!s)
                    ok = FALSE;
                code = code * st->width + s;
                }
            if(!ok)
                continue;
            word = GPOINTER_TO_INT(g_hash_table_lookup(st->host_index, &code));
            if(word){
                register gint32 e;
                for(e = g_array_index(st->first, gint32, word - 1); e < g_array_index(st->first, gint32, word); e++){
                    c4gpu_word_hit x;
                    x.pos = i; x.emit = e;
                    g_array_append_val(h, x);
                    }
                }
            }
        n_hits = h->len;
        hits = (c4gpu_word_hit*)g_array_free(h, FALSE);
    } else {
        cap = 4096 + n / 8;
        for(;;){
            hits = g_new(c4gpu_word_hit, cap);
            if(c4gpu_seed_scan(shim_get_ctx(), st->tab, sym, n, hits, cap, &n_hits) != 0){
                g_warning("c4gpu: %s -- this target is scanned on the CPU", c4gpu_last_error());
                g_free(hits); g_free(sym);
                return FALSE;
                }
            if(n_hits <= cap)
                break;
            g_free(hits);
            cap = n_hits;
            }
        }
    sdst.scans++; sdst.symbols += n; sdst.hits += n_hits;
    sdst.scan_ms += (g_get_monotonic_time() - t0) / 1e3;
    out->hits = hits; out->n_hits = n_hits; out->frame = frame;
    g_free(sym);
    return TRUE;
    }

static void seed_deliver_frame(ShimSeedTab *st, Sequence *target, ShimFrameHits *fh, GArray *mine){
    register gint64 k;
    gint64 t0 = g_get_monotonic_time();
    for(k = 0; k < fh->n_hits; k++){
        register ShimEmit *e = &g_array_index(st->emits, ShimEmit, fh->hits[k].emit);
        register gint tpos = fh->frame ? (fh->hits[k].pos * 3) + fh->frame - 1 : fh->hits[k].pos;       /* seeder.c:657-660 */
        seed_deliver(st->seeder, target, e, tpos - e->loader->tpos_modifier, mine);
        }
    sdst.host_ms += (g_get_monotonic_time() - t0) / 1e3;
    g_free(fh->hits);
    fh->hits = NULL;
    return;
    }

void Seeder_add_target(Seeder *seeder, Sequence *target){
    register ShimSeedTab *st = NULL;
    register Match *match = seeder->any_hsp_param->match;
    register gint i;
    register Seeder_QueryInfo *query_info;
    register GArray *mine = NULL, *theirs = NULL;
    ShimFrameHits frames[3];
    gint n_frames = 0;
    gboolean ok = TRUE;
    static gint off = -1;
    if(off < 0)
        off = (shim_env("C4GPU_SEED_OFF") || (shim_batch_size() <= 0)) ? 1 : 0;
    shim_mark("Seeder_add_target");
    if((!off) && (shim_env("C4GPU_SEED_HOST") || shim_ctx_nowait())){
        st = seed_state(seeder);
        if(st->hopeless)
            st = NULL;
        else if(!st->usable){
            if(!seed_worth_it(st, target)){
                st->walked += target->len;
                sdst.cpu_targets++;
                st = NULL;
            } else if(!seed_build(st)){
                st->hopeless = TRUE;
                st = NULL;
                }
            }
        }
    shim_mark("word table ready");
    if(!st){
        Seeder_add_target_cpu(seeder, target);
        return;
        }
    if(shim_env("C4GPU_SEED_CHECK")){
        /* the reference's own walk first, its seeds written down instead of extended (no set becomes non-empty, so its
         * report loop finds nothing to report) */
        theirs = seed_record = g_array_new(FALSE, FALSE, sizeof(ShimSeedRec));
        Seeder_add_target_cpu(seeder, target);
        seed_record = NULL;
        mine = g_array_new(FALSE, FALSE, sizeof(ShimSeedRec));
        }
    sdst.targets++;
    Sequence_share(target);
    if(match->target->is_translated){                          /* seeder.c:868-884 */
        for(i = 0; (i < 3) && ok; i++){
            register Sequence *aa_seq = Sequence_translate(target, match->mas->translate, i + 1);
            register Sequence *masked = Sequence_mask(aa_seq);
            register gchar *seq = Sequence_get_str(masked);
            Sequence_destroy(aa_seq);
            Sequence_destroy(masked);
            ok = seed_scan_string(st, seq, i + 1, &frames[n_frames]);
            if(ok)
                n_frames++;
            g_free(seq);
            }
    } else {
        register Sequence *masked = Sequence_mask(target);
        register gchar *seq = Sequence_get_str(masked);
        Sequence_destroy(masked);
        ok = seed_scan_string(st, seq, 0, &frames[0]);
        if(ok)
            n_frames = 1;
        g_free(seq);
        }
    if(!ok){
        /* nothing has been delivered yet: this target (and the rest of this seeder's) keeps the reference's walk */
        for(i = 0; i < n_frames; i++)
            g_free(frames[i].hits);
        st->hopeless = TRUE;
        sdst.targets--;
        sdst.cpu_targets++;
        if(theirs){                                   /* check mode: the reference's walk has already run, its seeds only written down */
            g_array_free(theirs, TRUE);
            g_array_free(mine, TRUE);
            }
        Seeder_add_target_cpu(seeder, target);
        Sequence_destroy(target);
        return;
        }
    for(i = 0; i < n_frames; i++)
        seed_deliver_frame(st, target, &frames[i], mine);
    Sequence_destroy(target);
    if(theirs){
        if(theirs->len != mine->len)
            g_error("c4gpu seed check: the reference's walk finds %u seeds, the device scan %u", theirs->len, mine->len);
        for(i = 0; i < (gint)mine->len; i++){
            register ShimSeedRec *a = &g_array_index(theirs, ShimSeedRec, i), *b = &g_array_index(mine, ShimSeedRec, i);
            if((a->query != b->query) || (a->param != b->param) || (a->query_pos != b->query_pos) || (a->target_pos != b->target_pos))
                g_error("c4gpu seed check: seed %d differs: reference (%u, %u), device scan (%u, %u)", i, a->query_pos,
                        a->target_pos, b->query_pos, b->target_pos);
            }
        sdst.checked += mine->len;
        g_array_free(theirs, TRUE);
        g_array_free(mine, TRUE);
        }
    shim_mark("target scanned, seeds delivered");
    /* Report matches, seeder.c:899-913 */
    for(i = 0; i < (gint)seeder->active_queryinfo_list->len; i++){
        query_info = seeder->active_queryinfo_list->pdata[i];
        if(Comparison_has_hsps(query_info->curr_comparison)){
            Comparison_finalise(query_info->curr_comparison);
            seeder->report_func(query_info->curr_comparison, seeder->user_data);
            }
        Comparison_destroy(query_info->curr_comparison);
        query_info->curr_comparison = NULL;
        }
    g_ptr_array_set_size(seeder->active_queryinfo_list, 0);
    shim_mark("comparisons reported");
    return;
    }

/* a seeder's word table goes with it (analysis.c builds a new seeder for every --fsmmemory load of queries) */
void Seeder_destroy(Seeder *seeder){
    register ShimSeedTab *st = seed_tabs ? g_hash_table_lookup(seed_tabs, seeder) : NULL;
    if(st){
        g_hash_table_remove(seed_tabs, seeder);
        if(st->tab)
            c4gpu_wordtab_destroy(st->tab);
        if(st->host_index)
            g_hash_table_destroy(st->host_index);
        if(st->codes){
            g_array_free(st->codes, TRUE); g_array_free(st->first, TRUE); g_array_free(st->emits, TRUE);
            }
        g_free(st);
        }
    Seeder_destroy_cpu(seeder);
    return;
    }

void shim_seed_report(void){
    if(shim_env("C4GPU_VERBOSE") && sdst.cpu_targets)
        g_message("c4gpu seed: %ld targets left to the reference's walk (their seeder's word table was not worth reading yet)",
                  sdst.cpu_targets);
    if(shim_env("C4GPU_VERBOSE") && sdst.targets)
        g_message("c4gpu seed: %ld targets walked in %ld device scans (%ld symbols): %ld word hits; scans %.0f ms, "
                  "delivery %.0f ms, word tables (%ld words, %ld emissions) %.0f ms%s", sdst.targets, sdst.scans, sdst.symbols,
                  sdst.hits, sdst.scan_ms, sdst.host_ms, sdst.words, sdst.emits, sdst.table_ms,
                  sdst.checked ? "; every seed equal to the reference's own walk" : "");
    return;
    }
