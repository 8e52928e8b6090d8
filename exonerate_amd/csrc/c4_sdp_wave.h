// c4_sdp_wave.h — SDP (seeded dynamic programming, exonerate src/sdp/) as a SPARSE wavefront: one 64-lane wave per
// query x target pair, lanes = 64 consecutive query positions of a strip, anti-diagonal stepping along the target, only
// over the steps where something is alive.
//
// What it computes: both passes of Scheduler_Pair_calculate (scheduler.c:1445-1500) with the cell function
// Scheduler_Cell_process (scheduler.c:859-1065) in the reference's candidate order, for both flavours of SDP_create
// (sdp.c:322-366): bidirectional from the seeds (affine, protein2dna) and boundary + spans (est2genome, protein2genome).
// The recurrence, its pruning rules and the traceback record are those of the dense sweep this file replaces (round 2's
// c4_sdp.inc / c4_sdp_bnd.inc, proven against 8 reference vector sets and two fuzzers); what is new is WHERE it runs:
//
//   * The reference's scheduler is push-style and sparse: a cell exists once something was pushed into it.  A state
//     nothing reaches holds (-987654321, max 0), which every pruning test (scheduler.c:1020-1024) rejects exactly as "no
//     such cell" does, so evaluating a superset of the live cells gives the same values.  The superset here: every cell
//     of a step (64 cells on one anti-diagonal of a strip) in which at least one cell can be reached.
//   * A strip is 64 query rows; lane l owns row ubase + l and at step c evaluates column v = c - l.  All transitions
//     advance the query by 0 or 1: the (u, v - at) sources sit in the lane's own ring of earlier cells (VGPRs), the
//     (u - 1, v - at) sources in the ring of the lane above, fetched once per step with DPP wave_shr:1.  Lane 0's upper
//     neighbour is the bottom row of the strip before, which that strip left behind as a sparse list of live cells.
//   * A strip's sweep is event driven: it starts at the first step that holds a seed, a boundary cell or a live cell of
//     the carried row, runs while anything in the rings is alive, and jumps to the next event when everything is dead.
//     Steps in which no cell is alive leave nothing behind.  The work is the reference's own (cells inside the X-drop),
//     rounded up to steps of 64 lanes.
//   * Every executed step appends one record (the traceback bytes of its 64 cells; in the reverse pass of the boundary
//     flavour: the boundary seed of each cell, scheduler.c:1198-1216) to the pair's stream, in 64 KB chunks taken from
//     one arena with an atomic counter: memory is proportional to the cells the X-drop visits, not to (Q+1) x (T+1).
//     The forward pass of the boundary flavour reads the reverse pass's records of the mirrored strip back to front:
//     a reverse step c' of lane l' is forward step T + 63 - c' of lane 63 - l'.
//   * Span (intron) loops are never DP steps (scheduler.c:888-921): a cell whose span state holds a score >= 0 freezes
//     it into the store of its query position, a boundary cell thaws the store of its own query position.  With
//     max_query = 0 that store is a recurrence along the target axis of ONE query row: it lives in the registers of the
//     lane that owns the row.
//   * The best start / end of a seed (sdp.c:110-153,257-289) is the maximum over the assignments that are path maxima,
//     ties to the first in the reference's processing order (target row, query position, transition id descending):
//     kept per seed as (score, 64-bit order) in LDS, updated by the one wave that owns the pair — no atomics.
//
// The per-lane cell function (Eval::cell) and the traceback walk are __host__ __device__: tests/sdp_sim.hip drives them
// on the CPU with plain loops over 64 lanes (same records, same walk) against the pinned oracle; the device driver
// below differs only in how lanes exchange cells (DPP), how loads are issued ahead, and where the streams live.
#pragma once
#include "c4_viterbi_kernel.h"
#include "c4gpu.h"

namespace c4sdp {
using namespace c4k;

#define SDP_HD __host__ __device__ __forceinline__

template <int V>
struct KI {
    static constexpr int value = V;
    SDP_HD constexpr operator int() const { return V; }
};
template <class F, int... I>
SDP_HD void sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(KI<I>{}), ...); }
template <int N, class F>
SDP_HD void sfor(F &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int CHUNK_LOG = 16, CHUNK_BYTES = 1 << CHUNK_LOG;     // arena granularity
constexpr int REC_HEAD = 64;                                     // bytes in front of a step record's lane arrays: {c}
enum { ST_REVREC = 0, ST_REVCARRY, ST_FWDREC, ST_FWDCARRY, ST_COUNT };
enum { SDP_OK = 0, SDP_FAIL_ARENA = 1, SDP_FAIL_WALK = 2 };
constexpr int SDP_NO_EVENT = 0x7fffffff;

struct SdpJob {                      // one pair
    long long q_off, t_off;          // into the coded sequences (t_off also: splice arrays, tn4)
    int Q, T, n_strips, n_seeds;
    int seed_off;                    // first seed in the launch's seed arrays (either pass) and in the best arrays
    int dir_off;                     // ST_COUNT directories of n_strips + 1 ints each: first entry of every strip
    long long tab_off[ST_COUNT];     // chunk tables (ints: chunk number of every CHUNK_BYTES piece of a stream)
    int tab_cap[ST_COUNT];
};
struct SdpDSeed { int strip, c, lane, val, sid; };   // val: start score (reverse pass) / hsp score >> 1 (seeded forward pass)
struct SdpBest { int score; unsigned ohi, olo; int pad; };
struct SdpLaunch {
    const KParams *kp;
    const SdpJob *jobs;
    const SdpDSeed *seeds;           // this pass's, per job sorted by (strip, c)
    const uint8_t *qcode, *tcode;
    const int *ss;
    long long ss_stride;
    const uint16_t *tn4;
    uint8_t *arena;
    unsigned *arena_next;
    unsigned n_chunks;
    int *tabs, *dirs;
    SdpBest *best_rev, *best_fwd;
    int *status;                     // per job
    int dropoff;
};

// ---- streams -------------------------------------------------------------------------------------------------------
// entry `idx` of a stream whose entries are BYTES long, 1 << LOG of them per chunk
SDP_HD uint8_t *stream_entry(uint8_t *arena, const int *tab, long long idx, int log, int bytes) {
    return arena + ((size_t)(unsigned)tab[idx >> log] << CHUNK_LOG) + (size_t)(idx & ((1 << log) - 1)) * bytes;
}
constexpr int per_chunk_log(int bytes) { int l = 0; while ((2 << l) * bytes <= CHUNK_BYTES) l++; return l; }

// ---- what one pass direction makes of a model ------------------------------------------------------------------
template <class M, bool FWD>
struct Plan {
    static constexpr int NS = M::NS, NT = M::NT;
    static constexpr bool is_span(int k) { return M::tr[k].in == M::tr[k].out && M::tr[k].calc < 0; }   // C4_Transition_is_span, c4.h:246
    static constexpr int src(int k) { return FWD ? M::tr[k].in : M::tr[k].out; }
    static constexpr int dst(int k) { return FWD ? M::tr[k].out : M::tr[k].in; }
    static constexpr bool is_adv(int k) { return !is_span(k) && (M::tr[k].aq || M::tr[k].at); }
    static constexpr bool is_stat(int k) { return !is_span(k) && !M::tr[k].aq && !M::tr[k].at; }
    static constexpr int SEEDST = FWD ? M::START : M::END, CB = FWD ? M::END : M::START;
    struct List { int n; int id[M::NT]; };
    // arrival order of the candidates of a state (c4_sdp.inc's header): by source row (larger target advance first), then
    // source column (larger query advance first), then transition id descending; lists are grouped by destination
    static constexpr bool before(int a, int b) {
        if (dst(a) != dst(b)) return dst(a) < dst(b);
        if (M::tr[a].at != M::tr[b].at) return M::tr[a].at > M::tr[b].at;
        if (M::tr[a].aq != M::tr[b].aq) return M::tr[a].aq > M::tr[b].aq;
        return a > b;
    }
    static constexpr List make_adv() {
        List l{};
        for (int k = 0; k < NT; k++) if (is_adv(k)) l.id[l.n++] = k;
        for (int i = 1; i < l.n; i++)
            for (int j = i; j > 0 && before(l.id[j], l.id[j - 1]); j--) { const int x = l.id[j]; l.id[j] = l.id[j - 1]; l.id[j - 1] = x; }
        return l;
    }
    static constexpr List make_stat() {                                  // the cell's own sweep: id descending
        List l{};
        for (int k = NT - 1; k >= 0; k--) if (is_stat(k)) l.id[l.n++] = k;
        return l;
    }
    static constexpr List ADV = make_adv(), STAT = make_stat();
    // every advancing transition has a lower id than every static one (C4_Model_topological_sort, c4.c:1418-1486): a cell
    // is only ever read by other cells in its final state
    static constexpr bool adv_below_stat() {
        int hi_adv = -1, lo_stat = NT;
        for (int k = 0; k < NT; k++) { if (is_adv(k) || is_span(k)) hi_adv = k; }
        for (int k = NT - 1; k >= 0; k--) if (is_stat(k)) lo_stat = k;
        return hi_adv < lo_stat;
    }
    static constexpr bool cb_is_static() { for (int k = 0; k < NT; k++) if (!is_span(k) && dst(k) == CB && !is_stat(k)) return false; return true; }
    static_assert(adv_below_stat(), "advancing transitions must precede the static ones");
    static_assert(cb_is_static(), "the call-back state is entered by static transitions only");
    static_assert(STAT.n <= 16, "one bit per static transition in the cell's sweep record");
    static_assert(M::MAXAQ == 1, "lanes exchange one query row per step");
    // states whose value crosses to the lane below (sources of transitions that advance the query)
    static constexpr bool exported(int s) { for (int k = 0; k < NT; k++) if (is_adv(k) && M::tr[k].aq && src(k) == s) return true; return false; }
    static constexpr int n_exported() { int n = 0; for (int s = 0; s < NS; s++) n += exported(s); return n; }
    static constexpr int exported_index(int s) { int n = 0; for (int x = 0; x < s; x++) n += exported(x); return n; }
    // spans (C4_Span, intron.c:660-672): one per state with a loop transition, in state order (Scheduler_get_span_map)
    static constexpr int n_spans() { int n = 0; for (int k = 0; k < NT; k++) n += is_span(k); return n; }
    static constexpr int span_loop(int i) { int n = 0; for (int s = 0; s < NS; s++) for (int k = 0; k < NT; k++) if (is_span(k) && M::tr[k].in == s) { if (n == i) return k; n++; } return -1; }
    static constexpr int span_state(int i) { return M::tr[span_loop(i) < 0 ? 0 : span_loop(i)].in; }
    static constexpr int span_of_state(int s) { for (int i = 0; i < n_spans(); i++) if (span_state(i) == s) return i; return -1; }
    static constexpr bool spans_ok() {
        for (int i = 0; i < n_spans(); i++) {
            const int l = span_loop(i);
            if (M::tr[l].aq || !M::tr[l].at) return false;                       // target introns only (max_query = 0)
            for (int k = 0; k < NT; k++) if (M::tr[k].in == span_state(i) && k != l && is_stat(k)) return false;
        }
        return true;
    }
    static_assert(spans_ok(), "spans: target loops, left by advancing transitions");
    // shadows: one designation in scope, always a target position
    static_assert(M::NDES <= 1, "one shadow designation");
    static constexpr bool owns(int s) { for (int h = 0; h < M::NSH; h++) if (M::sh[h].src_state_mask >> s & 1) return true; return false; }
    static constexpr bool shadows_on_target() { for (int h = 0; h < M::NSH; h++) if (!M::sh[h].on_target) return false; return true; }
    static_assert(shadows_on_target(), "shadows carry target positions");
    static constexpr bool consumes(int k) { return M::tr[k].dst_shadow_mask != 0; }
    // positions the calcs read
    static constexpr int match_at() { return Facts<M>::match_at(); }
    static constexpr int splice_at() { for (int k = 0; k < NT; k++) if (M::tr[k].calc >= 0 && (M::calc[M::tr[k].calc].kind == CALC_SPLICE_PRE || M::calc[M::tr[k].calc].kind == CALC_SPLICE_POST)) return M::tr[k].at; return 0; }
    static constexpr bool splice_at_ok() { for (int k = 0; k < NT; k++) if (M::tr[k].calc >= 0 && (M::calc[M::tr[k].calc].kind == CALC_SPLICE_PRE || M::calc[M::tr[k].calc].kind == CALC_SPLICE_POST) && M::tr[k].at != splice_at()) return false; return true; }
    static_assert(splice_at_ok(), "one target advance for all splice transitions");
    static constexpr bool uses_ss(int p) { for (int c = 0; c < M::NC; c++) if ((M::calc[c].kind == CALC_SPLICE_PRE || M::calc[c].kind == CALC_SPLICE_POST) && M::calc[c].param == p) return true; return false; }
    static constexpr bool has_match() { for (int c = 0; c < M::NC; c++) if (M::calc[c].kind >= CALC_MATCH_DNA && M::calc[c].kind <= CALC_MATCH_P2D) return true; return false; }
};

template <int NS>
struct SCell { int sc[NS], mx[NS], sd[NS], sh[NS]; };
struct SpanCache { int valid, score, max, seed, entry_t, thawed, sh; };   // Scheduler_SpanSeed of one query position
struct TVals { int mcode; int ss[4]; };                                    // what a cell reads of its target column(s)

// layout of a pass's streams
template <class M, bool FWD, bool BND>
struct Layout {
    using P = Plan<M, FWD>;
    static constexpr bool SH = FWD && M::NDES > 0;          // shadows only travel forward (scheduler.c:801-813)
    static constexpr bool TB = FWD || !BND;                  // the pass keeps a traceback
    static constexpr bool CBK = FWD || !BND;                 // ... and reports the best start / end of every seed
    static constexpr int NSP = (FWD && BND) ? P::n_spans() : 0;
    static constexpr int PTB = M::NS + 2, PTW = (PTB + 3) / 4;   // per cell: a pointer byte per state + 16 sweep bits
    static constexpr int RECW = TB ? PTW + NSP : 1;          // ints per lane in a step record
    static constexpr int REC_BYTES = REC_HEAD + 256 * RECW, RPC_LOG = per_chunk_log(REC_BYTES);
    static constexpr int CW = 3 + (SH ? 1 : 0);              // ints per carried state
    static constexpr int CENT_INTS = 2 + P::n_exported() * CW, CENT_BYTES = 4 * CENT_INTS, CPC_LOG = per_chunk_log(CENT_BYTES);
    static constexpr int ST_REC = FWD ? ST_FWDREC : ST_REVREC, ST_CARRY = FWD ? ST_FWDCARRY : ST_REVCARRY;
};

// ---- one cell -------------------------------------------------------------------------------------------------------
template <class M, bool FWD, bool BND>
struct Eval {
    using P = Plan<M, FWD>;
    using L = Layout<M, FWD, BND>;
    using C = SCell<M::NS>;
    static constexpr int NS = M::NS, NSPA = L::NSP > 0 ? L::NSP : 1;
    struct In {
        int u, v, Q, T;
        bool inside;                  // the cell lies in the lattice
        bool seed_here;               // a seed / boundary cell: SEEDST starts with seed_score, id seed_id
        int seed_score, seed_id;
        int mscore;                   // the match calc of this cell (substitution matrix, from LDS)
        int qrow24;                   // 24 * matrix row of the query residue the (1, x) transitions consume
        TVals tv;
        int dropoff, min_intron, max_intron, span_max_target;
        const uint16_t *tn4;          // split-codon calcs: 4-bit base masks by target position (this pair's)
        const int *cv;                // calc constants
    };
    struct Out {
        unsigned ptw[L::PTW];         // pointer bytes as the advancing arrivals left them, then the sweep bits
        int tf[NSPA];                 // per span: thawed here from the cell frozen at row (tf >> 1) - 1, bit 0: that cell's own thaw came first
        int bnd;                      // reverse boundary pass: seed id + 1 of a boundary cell
        bool alive;
    };

    // CBF(fire, seed id, score, transition id): an assignment into the call-back state that is a maximum of its path
    template <class CBF>
    SDP_HD static void cell(C &cur, const C (&own)[M::MAXAT + 1], const C (&up)[M::MAXAT + 2], SpanCache (&cache)[NSPA],
                            const In &in, Out &out, const KParams *kp, CBF &&cbf) {
        int pt[NS];
        sfor<NS>([&](auto S_) { constexpr int S = S_; cur.sc[S] = LOW; cur.mx[S] = 0; cur.sd[S] = 0; cur.sh[S] = 0; pt[S] = 0; });
        {   // Scheduler_Cell_seed (scheduler.c:1068-1082) / a boundary cell (sdp.c:233-255): nothing leads into that state
            const bool s = in.seed_here & in.inside;
            cur.sc[P::SEEDST] = s ? in.seed_score : LOW;
            cur.mx[P::SEEDST] = s ? in.seed_score : 0;
            cur.sd[P::SEEDST] = s ? in.seed_id : 0;
        }
        const int u = in.u, v = in.v;
        auto arrive = [&](auto K_, int s_sc, int s_mx, int s_sd, int s_sh, int sv) -> bool {
            constexpr int k = K_;
            constexpr TrDesc t = M::tr[k];
            constexpr int D = P::dst(k), S = P::src(k);
            int tscore = 0;
            if constexpr (!FWD && P::consumes(k)) tscore = 0;                       // scheduler.c:1004-1006
            else if constexpr (t.calc >= 0) {
                constexpr CalcDesc cd = M::calc[t.calc];
                if constexpr (cd.kind == CALC_CONST) tscore = in.cv[t.calc];
                else if constexpr (cd.kind >= CALC_MATCH_DNA && cd.kind <= CALC_MATCH_P2D) tscore = in.mscore;
                else if constexpr (cd.kind == CALC_SPLICE_PRE) tscore = in.cv[t.calc] + in.tv.ss[cd.param];
                else if constexpr (cd.kind == CALC_SPLICE_POST) {                  // Intron_calc_*, post (intron.c:150-160)
                    const int len = sv - s_sh + 2;
                    tscore = ((len < in.min_intron) | (len > in.max_intron)) ? LOW : in.tv.ss[cd.param];
                } else if constexpr (cd.kind == CALC_PHASE_POST) {                 // phase.c:188-213 (forward only: consumes)
                    const int cis = s_sh, tpos = sv;
                    const bool good = cis >= cd.param;
                    const int c1 = good ? cis - 1 : 0, tpc = tpos < 0 ? 0 : tpos;
                    unsigned n1, n2, n3;
                    if constexpr (cd.param == 1) {
                        const unsigned a = in.tn4[c1], b = in.tn4[tpc + 1];
                        n1 = a & 0xf; n2 = (b >> 4) & 0xf; n3 = b & 0xf;
                    } else {
                        const unsigned a = in.tn4[c1], b = in.tn4[tpc];
                        n1 = (a >> 4) & 0xf; n2 = a & 0xf; n3 = b & 0xf;
                    }
                    const int row = kp->codon_row[n1 | (n2 << 4) | (n3 << 8)];
                    tscore = good ? kp->submat[in.qrow24 + row] : LOW;
                }
            }
            const int dsc = s_sc + tscore;
            bool ok = ((s_mx - dsc) <= in.dropoff) & (dsc > cur.sc[D]) & in.inside;   // :1023-1024, :1047-1051
            if constexpr (FWD) ok = ok & (dsc >= 0);                                  // :1020-1022
            const bool newmax = dsc >= s_mx;                                          // a new maximum of its path (:813-835)
            cur.sc[D] = ok ? dsc : cur.sc[D];
            cur.sd[D] = ok ? s_sd : cur.sd[D];
            cur.mx[D] = ok ? (newmax ? dsc : s_mx) : cur.mx[D];
            pt[D] = ok ? k + 1 : pt[D];
            if constexpr (L::SH) {                                                    // shadow start + transport (:733-747,801-813)
                const int shv = P::owns(S) ? sv : s_sh;
                cur.sh[D] = ok ? shv : cur.sh[D];
            }
            if constexpr (L::CBK && D == P::CB) cbf(ok & newmax, s_sd, dsc, KI<k>{});
            return ok;
        };
        // advancing transitions, in arrival order
        sfor<P::ADV.n>([&](auto I_) {
            constexpr int k = P::ADV.id[I_];
            constexpr TrDesc t = M::tr[k];
            constexpr int S = P::src(k);
            const C &s = t.aq ? up[t.at + 1] : own[t.at];
            arrive(KI<k>{}, s.sc[S], s.mx[S], s.sd[S], s.sh[S], v - t.at);
        });
        // the pointers as the advancing arrivals left them; then the cell's own sweep, one bit per static transition that assigned
        unsigned bytes[L::PTW * 4];
        sfor<L::PTW * 4>([&](auto B_) { constexpr int B = B_; if constexpr (B < NS) bytes[B] = (unsigned)pt[B]; else bytes[B] = 0; });
        unsigned assigned = 0;
        sfor<P::STAT.n>([&](auto I_) {
            constexpr int k = P::STAT.id[I_];
            constexpr int S = P::src(k);
            const bool ok = arrive(KI<k>{}, cur.sc[S], cur.mx[S], cur.sd[S], cur.sh[S], v);
            assigned |= ok ? (1u << I_) : 0u;
        });
        bytes[NS] = assigned & 0xff; bytes[NS + 1] = assigned >> 8;
        sfor<L::PTW>([&](auto W_) { constexpr int W = W_;
            out.ptw[W] = bytes[4 * W] | (bytes[4 * W + 1] << 8) | (bytes[4 * W + 2] << 16) | (bytes[4 * W + 3] << 24);
        });
        // span events of the cell's sweep, id descending: freeze at the loop (scheduler.c:890-921 -> :619-643), thaw at a way out
        // of the span state (:940-986 -> :567-613) — boundary cells only, and only if that way leads into the lattice
        if constexpr (L::NSP > 0) {
            sfor<L::NSP>([&](auto I_) {
                constexpr int i = I_;
                constexpr int X = P::span_state(i), LOOP = P::span_loop(i);
                SpanCache &st = cache[i];
                int thawed = 0;
                out.tf[i] = 0;
                sfor<M::NT>([&](auto R_) {
                    constexpr int k = M::NT - 1 - R_;
                    if constexpr (M::tr[k].in == X) {
                        if constexpr (k == LOOP) {
                            if ((cur.sc[X] >= 0) & (in.span_max_target != 0) & ((!st.valid) | (st.score <= cur.sc[X]))) {
                                st.valid = 1; st.score = cur.sc[X]; st.max = cur.mx[X]; st.seed = cur.sd[X];
                                st.entry_t = v; st.thawed = thawed; st.sh = cur.sh[X];
                            }
                        } else {
                            if (in.seed_here & in.inside & (u + M::tr[k].aq <= in.Q) & (v + M::tr[k].at <= in.T) & (st.valid != 0)) {
                                if (st.entry_t + in.span_max_target >= v) {
                                    if (cur.sc[X] < st.score) {
                                        cur.sc[X] = st.score; cur.mx[X] = st.max; cur.sd[X] = st.seed; cur.sh[X] = st.sh;
                                        out.tf[i] = ((st.entry_t + 1) << 1) | st.thawed;
                                        thawed = 1;
                                    }
                                } else st.valid = 0;
                            }
                        }
                    }
                });
            });
        }
        bool any = false;
        sfor<NS>([&](auto S_) { constexpr int S = S_; any = any | (cur.sc[S] != LOW); });
        out.alive = any;
        out.bnd = 0;
        if constexpr (BND && !FWD) {                              // a boundary cell? (scheduler.c:1198-1216)
            int b = 0;
            constexpr int NSPR = P::n_spans();
            sfor<NSPR>([&](auto R_) { constexpr int i = NSPR - 1 - R_; constexpr int X = P::span_state(i);
                b = cur.sc[X] > 0 ? cur.sd[X] + 1 : b;            // the first span state above 0 wins: assigned last
            });
            b = cur.sc[P::CB] >= 0 ? cur.sd[P::CB] + 1 : b;
            out.bnd = any ? b : 0;
        }
    }
};

// ---- the walk ---------------------------------------------------------------------------------------------------
// One pass's traceback as the walk sees it (table driven: it is a chain of dependent loads, not arithmetic)
struct SdpWalkTab {
    int n_states, n_tr, n_stat, forward, bnd, n_spans;
    int ptw, rec_bytes, rpc_log, st_rec;
    int tr_aq[C4GPU_MAX_TRANSITIONS], tr_at[C4GPU_MAX_TRANSITIONS], tr_in[C4GPU_MAX_TRANSITIONS], tr_out[C4GPU_MAX_TRANSITIONS];
    int stat_id[16], stat_dst[16], stat_index[C4GPU_MAX_TRANSITIONS];
    int span_of_state[C4GPU_MAX_STATES], span_loop[4];
};
struct SdpWalkOut { int score, q, t, n_runs, status, pad; long long runs_off; };

template <class M, bool FWD, bool BND>
inline SdpWalkTab make_walk_tab() {
    using P = Plan<M, FWD>;
    using L = Layout<M, FWD, BND>;
    SdpWalkTab w{};
    w.n_states = M::NS; w.n_tr = M::NT; w.n_stat = P::STAT.n; w.forward = FWD; w.bnd = BND; w.n_spans = L::NSP;
    w.ptw = L::PTW; w.rec_bytes = L::REC_BYTES; w.rpc_log = L::RPC_LOG; w.st_rec = L::ST_REC;
    for (int k = 0; k < C4GPU_MAX_TRANSITIONS; k++) w.stat_index[k] = -1;
    for (int k = 0; k < M::NT; k++) { w.tr_aq[k] = M::tr[k].aq; w.tr_at[k] = M::tr[k].at; w.tr_in[k] = M::tr[k].in; w.tr_out[k] = M::tr[k].out; }
    for (int i = 0; i < P::STAT.n; i++) { w.stat_id[i] = P::STAT.id[i]; w.stat_dst[i] = P::dst(P::STAT.id[i]); w.stat_index[P::STAT.id[i]] = i; }
    for (int s = 0; s < C4GPU_MAX_STATES; s++) w.span_of_state[s] = s < M::NS ? P::span_of_state(s) : -1;
    for (int i = 0; i < 4; i++) w.span_loop[i] = i < P::n_spans() ? P::span_loop(i) : -1;
    return w;
}

// pass coordinates of strip 0's first row: the forward pass of the boundary flavour anchors its strips at Q so that its
// rows are those of the reverse pass's strips, mirrored
SDP_HD int strip_ubase0(int Q, int n_strips, bool fwd_bnd) { return fwd_bnd ? Q - 64 * (n_strips - 1) - 63 : 0; }

// index of the record of step c of strip `strip` in [lo, hi], or -1: records of a strip are in ascending c
SDP_HD long long find_record(const uint8_t *arena, const int *tab, long long lo, long long hi, int c, int log, int bytes) {
    while (lo <= hi) {
        const long long mid = (lo + hi) >> 1;
        const int rc = *reinterpret_cast<const int *>(stream_entry(const_cast<uint8_t *>(arena), tab, mid, log, bytes));
        if (rc == c) return mid;
        if (rc < c) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// The path of a seed's best start / end: from the assignment that set it back to the seed cell (seeded flavour: sdp.c
// 640-734 without a boundary) or to the boundary cell it started from (SDP_Seed_find_start, sdp.c:640-659).  EMIT(id, n):
// n operations of transition id, in walk order (end -> start).  Returns SDP_OK or SDP_FAIL_WALK.
template <class EMIT>
SDP_HD int walk_path(const SdpWalkTab &W, const SdpJob &job, const uint8_t *arena, const int *tabs, const int *dirs,
                     const SdpBest &best, int *out_q, int *out_t, EMIT &&emit) {
    const int Q = job.Q, T = job.T, NT = W.n_tr;
    const int *tab = tabs + job.tab_off[W.st_rec];
    const int *dir = dirs + job.dir_off + W.st_rec * (job.n_strips + 1);
    const int ubase0 = strip_ubase0(Q, job.n_strips, W.forward && W.bnd);
    unsigned long long order = ((unsigned long long)best.ohi << 32) | best.olo;
    int id = NT - 1 - (int)(order % (unsigned)NT);
    order /= (unsigned)NT;
    int u = (int)(order % (unsigned)(Q + 1)), v = (int)(order / (unsigned)(Q + 1));   // the cell the assignment came FROM
    *out_q = W.forward ? u + W.tr_aq[id] : Q - (u + W.tr_aq[id]);                   // its destination, sequence coordinates
    *out_t = W.forward ? v + W.tr_at[id] : T - (v + W.tr_at[id]);
    emit(id, 1);
    int X = W.forward ? W.tr_in[id] : W.tr_out[id];
    int when = W.stat_index[id] >= 0 ? W.stat_index[id] : W.n_stat, see_thaw = 1;
    int strip = -1, lane = 0;
    long long idx = -1;
    const unsigned *lanes = nullptr;
    // the record of cell (u, v): `back` steps before the current one when that is the same strip and the step left a record
    auto locate = [&](int back) -> bool {
        const int ns = (u - ubase0) >> 6, nl = (u - ubase0) & 63, c = v + nl;
        long long lo = dir[ns], hi = (long long)dir[ns + 1] - 1, found = -1;
        if (ns == strip && back >= 0 && idx - back >= lo) {
            if (*reinterpret_cast<const int *>(stream_entry(const_cast<uint8_t *>(arena), tab, idx - back, W.rpc_log, W.rec_bytes)) == c) found = idx - back;
            else hi = idx;
        }
        if (found < 0) found = find_record(arena, tab, lo, hi, c, W.rpc_log, W.rec_bytes);
        if (found < 0) return false;
        strip = ns; lane = nl; idx = found;
        lanes = reinterpret_cast<const unsigned *>(stream_entry(const_cast<uint8_t *>(arena), tab, idx, W.rpc_log, W.rec_bytes) + REC_HEAD);
        return true;
    };
    if (!locate(-1)) return SDP_FAIL_WALK;
    for (;;) {
        auto byte_at = [&](int b) -> unsigned { return (lanes[(b >> 2) * 64 + lane] >> ((b & 3) * 8)) & 0xffu; };
        const int span = W.span_of_state[X];
        if (W.n_spans && when == W.n_stat && span >= 0 && see_thaw) {
            const int tf = (int)lanes[(W.ptw + span) * 64 + lane];
            if (tf) {
                // thawed here: the rows since the freeze as loop operations, then on in the cell that froze the state, as
                // the freeze saw it there
                const int entry_t = (tf >> 1) - 1;
                if (v - entry_t > 0) emit(W.span_loop[span], v - entry_t);
                v = entry_t;
                see_thaw = tf & 1;
                if (!locate(-1)) return SDP_FAIL_WALK;
                continue;
            }
        }
        const unsigned assigned = byte_at(W.n_states) | (byte_at(W.n_states + 1) << 8);
        int by = -1;
        for (int j = when - 1; j >= 0; j--)
            if (((assigned >> j) & 1u) && W.stat_dst[j] == X) { by = j; break; }
        if (by >= 0) id = W.stat_id[by];                              // same cell, an earlier step of its sweep
        else {
            const int p = (int)byte_at(X);
            if (!p) break;                                            // the seed state of a seed / boundary cell: the path begins here
            id = p - 1;
            u -= W.tr_aq[id]; v -= W.tr_at[id];
            if (!locate(W.tr_aq[id] + W.tr_at[id])) return SDP_FAIL_WALK;
        }
        emit(id, 1);
        X = W.forward ? W.tr_in[id] : W.tr_out[id];
        when = by >= 0 ? by : W.n_stat;
        see_thaw = 1;
    }
    return SDP_OK;
}


#ifndef C4SDP_HOST_SIM          // tests/sdp_sim.hip compiles everything above for the host only
// ---- the device driver ------------------------------------------------------------------------------------------
// One wave per pair (block = 64 threads, blockIdx.x = job).  Strips in ascending order; inside a strip the event-driven
// step loop described at the top of this file.
constexpr int SDP_BEST_LDS = 256;                      // seeds of a pair whose best start / end lives in LDS (the rest: global)

template <class M, bool FWD, bool BND>
struct WaveSweep {
    using P = Plan<M, FWD>;
    using L = Layout<M, FWD, BND>;
    using E = Eval<M, FWD, BND>;
    using C = SCell<M::NS>;
    static constexpr int NS = M::NS, NEXP = P::n_exported();

    __device__ __forceinline__ static int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }

    __device__ static void run(const SdpLaunch &A, KParams *kp, SdpBest *lbest) {
        const int lane = (int)threadIdx.x;
        const int jx = (int)blockIdx.x;
        const SdpJob job = A.jobs[jx];
        const int Q = job.Q, T = job.T, n_strips = job.n_strips;
        // constants to LDS / registers
        {
            const int *src = reinterpret_cast<const int *>(A.kp);
            int *dst = reinterpret_cast<int *>(kp);
            for (int i = lane; i < (int)(sizeof(KParams) / 4); i += 64) dst[i] = src[i];
        }
        SdpBest *gbest = (FWD ? A.best_fwd : A.best_rev) + job.seed_off;
        if constexpr (L::CBK)
            for (int i = lane; i < job.n_seeds && i < SDP_BEST_LDS; i += 64) lbest[i] = SdpBest{LOW, 0xffffffffu, 0xffffffffu, 0};
        if constexpr (L::CBK)
            for (int i = SDP_BEST_LDS + lane; i < job.n_seeds; i += 64) gbest[i] = SdpBest{LOW, 0xffffffffu, 0xffffffffu, 0};
        __syncthreads();
        int cv[M::NC];
        sfor<M::NC>([&](auto I_) { cv[I_] = kp->calc_value[I_]; });
        const int min_intron = kp->min_intron, max_intron = kp->max_intron;
        const uint8_t *qc = A.qcode + job.q_off, *tc = A.tcode + job.t_off;
        const int *ssb = A.ss ? A.ss + job.t_off : nullptr;
        const uint16_t *tn4 = A.tn4 ? A.tn4 + job.t_off : nullptr;
        int *tab_rec = A.tabs + job.tab_off[L::ST_REC], *tab_carry = A.tabs + job.tab_off[L::ST_CARRY];
        int *dir_rec = A.dirs + job.dir_off + L::ST_REC * (n_strips + 1), *dir_carry = A.dirs + job.dir_off + L::ST_CARRY * (n_strips + 1);
        const int *tab_rrec = A.tabs + job.tab_off[ST_REVREC];                       // forward boundary pass: the reverse pass's records
        const int *dir_rrec = A.dirs + job.dir_off + ST_REVREC * (n_strips + 1);
        using LR = Layout<M, false, BND>;
        int rec_count = 0, carry_count = 0, prev_carry_first = 0;
        uint8_t *rec_chunk = nullptr, *carry_chunk = nullptr;
        const SdpDSeed *seeds = A.seeds + job.seed_off;
        int sp = 0;
        const int ubase0 = strip_ubase0(Q, n_strips, FWD && BND);
        const int tlast = T > 0 ? T - 1 : 0;
        bool failed = false;

        // a chunk for entry idx of a stream (lane-uniform): returns its base, or NULL when the arena / table is exhausted
        auto new_chunk = [&](int *tab, int cap, int slot) -> uint8_t * {
            unsigned ch = 0;
            if (lane == 0) ch = atomicAdd(A.arena_next, 1u);
            ch = (unsigned)rfl((int)ch);
            if (ch >= A.n_chunks || slot >= cap) return nullptr;
            if (lane == 0) tab[slot] = (int)ch;
            return A.arena + ((size_t)ch << CHUNK_LOG);
        };

        // a pending update of a seed's best start / end: resolved lane by lane, the one wave of the pair is the only writer
        auto best_update = [&](bool fire, int sid, int score, unsigned long long order) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(fire);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int s_sid = __builtin_amdgcn_readlane(sid, l), s_sc = __builtin_amdgcn_readlane(score, l);
                const unsigned s_hi = (unsigned)__builtin_amdgcn_readlane((int)(order >> 32), l);
                const unsigned s_lo = (unsigned)__builtin_amdgcn_readlane((int)(order & 0xffffffffu), l);
                if (lane == 0) {
                    SdpBest *e = s_sid < SDP_BEST_LDS ? &lbest[s_sid] : &gbest[s_sid];
                    const SdpBest cur = *e;
                    const bool better = (s_sc > cur.score) | ((s_sc == cur.score) & ((s_hi < cur.ohi) | ((s_hi == cur.ohi) & (s_lo < cur.olo))));
                    if (better) *e = SdpBest{s_sc, s_hi, s_lo, 0};
                }
            }
        };

        for (int strip = 0; strip < n_strips && !failed; strip++) {
            if (lane == 0) { dir_rec[strip] = rec_count; dir_carry[strip] = carry_count; }
            const int u = ubase0 + 64 * strip + lane;
            const bool row_ok = (u >= 0) & (u <= Q);
            // the query residue the (1, x) transitions into this row consume
            int qpos = FWD ? u - 1 : Q - u;
            qpos = qpos < 0 ? 0 : (qpos > Q - 1 ? (Q > 0 ? Q - 1 : 0) : qpos);
            const int qrow24 = 24 * (int)qc[qpos];
            C cur, own[M::MAXAT + 1], up[M::MAXAT + 2];
            auto kill = [&](C &x) { sfor<NS>([&](auto S_) { constexpr int S = S_; x.sc[S] = LOW; x.mx[S] = 0; x.sd[S] = 0; x.sh[S] = 0; }); };
            kill(cur);
            sfor<M::MAXAT + 1>([&](auto I_) { kill(own[I_]); });
            sfor<M::MAXAT + 2>([&](auto I_) { kill(up[I_]); });
            SpanCache cache[E::NSPA];
            sfor<E::NSPA>([&](auto I_) { cache[I_] = SpanCache{0, 0, 0, 0, 0, 0, 0}; });
            // ---- event sources ----
            // seeds of this strip (sorted by c)
            while (sp < job.n_seeds && seeds[sp].strip < strip) sp++;
            SdpDSeed nseed = SdpDSeed{-1, SDP_NO_EVENT, 0, 0, 0};
            auto load_seed = [&]() {
                nseed = SdpDSeed{-1, SDP_NO_EVENT, 0, 0, 0};
                if constexpr (FWD && BND) return;              // the forward boundary pass starts from the boundary cells only
                if (sp < job.n_seeds) {
                    const SdpDSeed s = seeds[sp];
                    if (s.strip == strip) {
                        nseed = s;
                        if constexpr (FWD && !BND) nseed.val = A.best_rev[job.seed_off + s.sid].score - s.val;   // sdp.c:79-93
                    }
                }
            };
            load_seed();
            // carried row of the strip before: entries {v, -, exported states}
            int ci = 0, ci_end = 0, cin_v = SDP_NO_EVENT;
            int cin[NEXP > 0 ? NEXP * L::CW : 1];
            if (strip > 0) { ci = prev_carry_first; ci_end = carry_count; }
            prev_carry_first = carry_count;
            auto load_carry = [&]() {
                cin_v = SDP_NO_EVENT;
                if (ci < ci_end) {
                    const int *e = reinterpret_cast<const int *>(stream_entry(A.arena, tab_carry, ci, L::CPC_LOG, L::CENT_BYTES));
                    cin_v = e[0];
                    sfor<(NEXP > 0 ? NEXP * L::CW : 1)>([&](auto I_) { cin[I_] = e[2 + I_]; });
                }
            };
            // boundary cells (forward boundary pass): the reverse pass's records of the mirrored strip, last to first
            int bi = -1, bi_lo = 0, b_cf = SDP_NO_EVENT, b_val = 0;
            auto load_bnd = [&]() {
                b_cf = SDP_NO_EVENT;
                if constexpr (FWD && BND) {
                    if (bi >= bi_lo) {
                        const uint8_t *r = stream_entry(A.arena, tab_rrec, bi, LR::RPC_LOG, LR::REC_BYTES);
                        b_cf = T + 63 - *reinterpret_cast<const int *>(r);
                        b_val = reinterpret_cast<const int *>(r + REC_HEAD)[63 - lane];
                    }
                }
            };
            if constexpr (FWD && BND) { bi_lo = dir_rrec[n_strips - 1 - strip]; bi = dir_rrec[n_strips - strip] - 1; }
            load_bnd();

            load_carry();

            auto next_event = [&]() -> int {
                int e = nseed.c;
                e = cin_v < e ? cin_v : e;
                e = b_cf < e ? b_cf : e;
                return e;
            };
            // target-side values of a step, one step ahead
            auto load_tv = [&](int c) -> TVals {
                const int v = c - lane;
                int mpos = FWD ? v - P::match_at() : T - v, spos = FWD ? v - P::splice_at() : T - v;
                mpos = mpos < 0 ? 0 : (mpos > tlast ? tlast : mpos);
                spos = spos < 0 ? 0 : (spos > tlast ? tlast : spos);
                TVals tv;
                tv.mcode = P::has_match() ? (int)tc[mpos] : 0;
                sfor<4>([&](auto K_) { constexpr int K = K_;
                    if constexpr (P::uses_ss(K)) tv.ss[K] = ssb[(long long)K * A.ss_stride + spos]; else tv.ss[K] = 0;
                });
                return tv;
            };

            int c = next_event();
            int dead_run = M::MAXAT + 2;
            TVals tv = load_tv(c == SDP_NO_EVENT ? 0 : c);
            while (c <= T + 63) {
                TVals tv_next = load_tv(c + 1);
                const int v = c - lane;
                const bool inside = row_ok & (v >= 0) & (v <= T);
                // ---- rings: the cell above (lane l - 1's last cell; lane 0: the carried row), then age everything ----
                const bool has_cin = cin_v == c;
                sfor<M::MAXAT + 1>([&](auto R_) { constexpr int a = M::MAXAT + 1 - R_; if constexpr (a >= 2) up[a] = up[a - 1]; });
                sfor<NS>([&](auto S_) { constexpr int S = S_;
                    if constexpr (P::exported(S)) {
                        constexpr int X = P::exported_index(S) * L::CW;
                        const int o_sc = has_cin ? cin[X] : LOW, o_mx = has_cin ? cin[X + 1] : 0, o_sd = has_cin ? cin[X + 2] : 0;
                        up[1].sc[S] = dpp_shr1(o_sc, cur.sc[S]);
                        up[1].mx[S] = dpp_shr1(o_mx, cur.mx[S]);
                        up[1].sd[S] = dpp_shr1(o_sd, cur.sd[S]);
                        if constexpr (L::SH) { const int o_sh = has_cin ? cin[X + 3] : 0; up[1].sh[S] = dpp_shr1(o_sh, cur.sh[S]); }
                    }
                });
                if (has_cin) { ci++; load_carry(); dead_run = 0; }     // the carried cell ages through the ring before the next jump
                sfor<M::MAXAT>([&](auto R_) { constexpr int a = M::MAXAT - R_; if constexpr (a >= 2) own[a] = own[a - 1]; });
                own[1] = cur;
                // ---- seeds of this step ----
                typename E::In in;
                in.u = u; in.v = v; in.Q = Q; in.T = T; in.inside = inside;
                in.seed_here = false; in.seed_score = 0; in.seed_id = 0;
                while (nseed.c == c) {
                    if (lane == nseed.lane) { in.seed_here = true; in.seed_score = nseed.val; in.seed_id = nseed.sid; }
                    sp++;
                    load_seed();
                }
                if constexpr (FWD && BND) {
                    if (b_cf == c) {
                        if (b_val) { in.seed_here = true; in.seed_score = 0; in.seed_id = b_val - 1; }
                        bi--;
                        load_bnd();
                    }
                }
                in.tv = tv;
                in.mscore = P::has_match() ? kp->submat[qrow24 + tv.mcode] : 0;
                in.qrow24 = qrow24;
                in.dropoff = A.dropoff; in.min_intron = min_intron; in.max_intron = max_intron; in.span_max_target = max_intron;
                in.tn4 = tn4; in.cv = cv;
                typename E::Out out;
                E::cell(cur, own, up, cache, in, out, kp, [&](bool fire, int sid, int score, auto K_) {
                    constexpr int k = K_;
                    const unsigned long long order = ((unsigned long long)(unsigned)v * (unsigned)(Q + 1) + (unsigned)u) * (unsigned)M::NT + (unsigned)(M::NT - 1 - k);
                    best_update(fire, sid, score, order);
                });
                const unsigned long long alive = __builtin_amdgcn_ballot_w64(out.alive);
                if (alive) {
                    // ---- this step's record ----
                    if ((rec_count & ((1 << L::RPC_LOG) - 1)) == 0) {
                        rec_chunk = new_chunk(tab_rec, job.tab_cap[L::ST_REC], rec_count >> L::RPC_LOG);
                        if (!rec_chunk) { failed = true; break; }
                    }
                    uint8_t *r = rec_chunk + (size_t)(rec_count & ((1 << L::RPC_LOG) - 1)) * L::REC_BYTES;
                    if (lane == 0) *reinterpret_cast<int *>(r) = c;
                    unsigned *ln = reinterpret_cast<unsigned *>(r + REC_HEAD);
                    if constexpr (L::TB) {
                        sfor<L::PTW>([&](auto W_) { constexpr int W = W_; ln[W * 64 + lane] = out.ptw[W]; });
                        sfor<L::NSP>([&](auto I_) { constexpr int I = I_; ln[(L::PTW + I) * 64 + lane] = (unsigned)out.tf[I]; });
                    } else ln[lane] = (unsigned)out.bnd;
                    rec_count++;
                    // ---- the bottom row's cell for the strip below ----
                    if (NEXP > 0 && strip + 1 < n_strips && (alive >> 63)) {
                        if ((carry_count & ((1 << L::CPC_LOG) - 1)) == 0) {
                            carry_chunk = new_chunk(tab_carry, job.tab_cap[L::ST_CARRY], carry_count >> L::CPC_LOG);
                            if (!carry_chunk) { failed = true; break; }
                        }
                        if (lane == 63) {
                            int *e = reinterpret_cast<int *>(carry_chunk + (size_t)(carry_count & ((1 << L::CPC_LOG) - 1)) * L::CENT_BYTES);
                            e[0] = v; e[1] = 0;
                            sfor<NS>([&](auto S_) { constexpr int S = S_;
                                if constexpr (P::exported(S)) {
                                    constexpr int X = 2 + P::exported_index(S) * L::CW;
                                    e[X] = cur.sc[S]; e[X + 1] = cur.mx[S]; e[X + 2] = cur.sd[S];
                                    if constexpr (L::SH) e[X + 3] = cur.sh[S];
                                }
                            });
                        }
                        carry_count++;
                    }
                    dead_run = 0;
                } else dead_run++;
                c++;
                tv = tv_next;
                if (dead_run > M::MAXAT + 1) {                  // every ring entry is dead: on to the next event
                    const int e = next_event();
                    if (e == SDP_NO_EVENT) break;
                    if (e > c) { c = e; tv = load_tv(c); }
                }
            }
        }
        if (lane == 0) { dir_rec[n_strips] = rec_count; dir_carry[n_strips] = carry_count; }
        if (failed && lane == 0) A.status[jx] = SDP_FAIL_ARENA;
        if constexpr (L::CBK) {
            __syncthreads();
            for (int i = lane; i < job.n_seeds && i < SDP_BEST_LDS; i += 64) gbest[i] = lbest[i];
        }
    }
};

template <class M, bool FWD, bool BND>
__global__ __launch_bounds__(64) void sdp_wave_kernel(const SdpLaunch A) {
    __shared__ KParams kp;
    __shared__ SdpBest lbest[SDP_BEST_LDS];
    WaveSweep<M, FWD, BND>::run(A, &kp, lbest);
}

// one thread per seed: the path of its best start / end as (transition, count) runs; two walks — count, then write
struct SdpWalkLaunch {
    SdpWalkTab tab;
    const SdpJob *jobs;
    const int *seed_job;             // per seed slot: its job
    const SdpBest *best, *gate;      // gate: the forward pass's best (a seed below the threshold is never asked for its path)
    const uint8_t *arena;
    const int *tabs, *dirs;
    const int *status;
    int n_seeds, threshold;
    unsigned *runs;                  // pairs (transition, count)
    unsigned long long *runs_used, runs_cap;
    SdpWalkOut *out;
};
static __global__ void sdp_walk_kernel2(const SdpWalkLaunch A) {
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= A.n_seeds) return;
    SdpWalkOut w = {LOW, 0, 0, 0, SDP_OK, 0, 0};
    const SdpBest b = A.best[s];
    const int jx = A.seed_job[s];
    if (A.status[jx] == SDP_OK && b.score != LOW && A.gate[s].score >= A.threshold) {
        const SdpJob job = A.jobs[jx];
        w.score = b.score;
        int n = 0, last = -1;
        w.status = walk_path(A.tab, job, A.arena, A.tabs, A.dirs, b, &w.q, &w.t, [&](int id, int cnt) { (void)cnt; if (id != last) { n++; last = id; } });
        if (w.status == SDP_OK) {
            const unsigned long long off = atomicAdd(A.runs_used, (unsigned long long)n);
            if (off + (unsigned long long)n > A.runs_cap) w.status = SDP_FAIL_ARENA;
            else {
                unsigned *o = A.runs + 2 * off;
                int k = -1;
                last = -1;
                walk_path(A.tab, job, A.arena, A.tabs, A.dirs, b, &w.q, &w.t, [&](int id, int cnt) {
                    if (id != last) { k++; o[2 * k] = (unsigned)id; o[2 * k + 1] = 0; last = id; }
                    o[2 * k + 1] += (unsigned)cnt;
                });
                w.n_runs = n; w.runs_off = (long long)off;
            }
        }
    }
    A.out[s] = w;
}
#endif  // C4SDP_HOST_SIM

}  // namespace c4sdp
