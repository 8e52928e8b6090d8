// c4_win16_kernel.h — the region windows of the two-pass FIND_REGION (c4_viterbi_kernel.h, SEED = 2; viterbi.c:403-412 for
// the payload they carry) with TWO windows per lane in packed 16-bit halves: the form the score pass (c4_viterbi16_kernel.h)
// and the checkpoint pass (c4_ckpt16_kernel.h) took, now with the region-start payload.
//
// What a cell holds per inner state, low half = window A, high half = window B:
//   sc   the score (v_pk_add_i16 clamp / v_pk_max_i16; "unset" = -32 768, adds saturate);
//   rq   where the path into this cell left START: the query row (window rows are lattice rows: a window starts in row 0)
//        — or, bit 15 set, the row of the dumped cell the path entered the window through;
//   rt   ... and the window column (unsigned 16 bits: a window spans one dump interval plus the dumped columns, the host
//        keeps dump intervals below 2^15) — or, for an entry cell, state x dumped columns + dumped column;
//   il   for the states whose intron-start shadow something can still read: the open intron's length so far (saturating).
// A transition is: candidate = source + calc (one packed add), mask = candidate beats the holder (v_pk_sub_i16 clamp +
// v_pk_ashrrev_i16, strict <, viterbi.c:766-775), score = packed max, payloads = v_bfi_b32 under the mask: six or seven
// instructions for two cells where the 32-bit window spends add, compare and a select per slot on one.
//
// START and END occupy no registers: the transitions out of START propose 0 with the cell's own position as payload in
// every cell (local scope), and END is never computed: the score pass reports the state END was entered from in the best
// end cell (its "root", DevResult::last_srp), a pair's first window ends in THAT state, and only the states that can reach
// the root are computed at all (Roots, c4_viterbi16_kernel.h: for est2genome one strand's four states instead of eight —
// the other strand's states feed nothing a window reports).
//
// The windows start from the packed score pass's 16-bit dumps (Dump16, c4_viterbi16_kernel.h) and chain on the device as the
// 32-bit ones do (viterbi_kernel_mw, SEED 2): a window whose corner payload names an entry cell goes on, in the same
// workgroup, with the window one dump interval further left, until the payload is a real start or the hop budget is spent.
// The two windows of a lane hop independently; a pair that is finished idles as a one-cell window while the other goes on.
//
// Exactness: that of the packed score pass (its header) — every cell on the optimal path holds its reference value (a
// local path's prefixes score between 0 and the best score, far inside 16 bits under the host's guard pk16_fits), a cell
// off the path may saturate at -32 768 and then loses against every path cell's real candidate exactly as the -987 654 321
// candidates of the 32-bit kernels do.  Winner, tie-break and therefore payload of every path cell are the reference's;
// the host checks every chain's first corner score against the score pass as before.
#pragma once
#include "c4_ckpt16_kernel.h"

namespace c4k {

template <class M, int R, int ROOT>
struct WaveWin16 {
    using F = Facts<M>;
    using D16 = Dump16<M>;
    using RT = Roots<M>;
    static constexpr int NS = M::NS, NCOL = M::MAXAT + 1, W = 64 * R, MAXAT = M::MAXAT, DC = M::MAXAT;
    static constexpr int SEEDW = D16::SEEDW16;
    static constexpr bool live(int s) { return D16::live(s); }
    static constexpr bool inner(int s) { return RT::member(ROOT, s); }       // the states this pass computes
    static constexpr bool exported(int s) { return inner(s) && F::exported(s); }
    static constexpr int n_exp() { int n = 0; for (int s = 0; s < NS; s++) n += exported(s); return n; }
    static constexpr int n_exp_live() { int n = 0; for (int s = 0; s < NS; s++) n += (exported(s) && live(s)); return n; }
    static constexpr int n_exp_live_all() { int n = 0; for (int s = 0; s < NS; s++) n += (F::exported(s) && live(s)); return n; }
    static constexpr int BND_ALL = F::n_exported() * 3 + n_exp_live_all();
    static constexpr int BND = n_exp() * 3 + n_exp_live();   // ints per column between strips (at most BND_ALL: what the host lays out)
    static constexpr int CS = 1 + M::NDES + 2;            // the reference's FIND_REGION cell (viterbi.c:154-173)
    static_assert(!F::has_phase(), "split-codon calcs are not packed");
    static_assert(M::NDES <= 1, "one shadow designation");
    static_assert(M::START == 0 && M::END == 1, "state numbering of the closed model");
    static_assert(!F::exported(M::START), "START advances nothing");
    static_assert(NS * DC < 0x8000, "entry-cell identity in 16 bits");
    struct C16 { int sc[NS]; int il[NS]; int rq[NS]; int rt[NS]; };
    typedef __attribute__((address_space(3))) int lds_int;
    __device__ __forceinline__ static lds_int *lds_at(int a) { return (lds_int *)(size_t)(unsigned)a; }
    __device__ __forceinline__ static int lds_addr(const lds_int *p) { return (int)(unsigned)(size_t)p; }
    using P16 = Prof16<R>;
    using ST = Stage16<BND>;

    const KParams *kp;
    int lane;
    const uint8_t *qc[2], *tc[2];
    const uint2 *ss16[2];
    const int *seed_rd[2];
    int Q[2], T[2], q0[2], t0[2], tlast[2], seed_rows[2], final_state[2];
    bool seeded[2];
    int Qm, Tm;
    // the intron length counter is kept minus (min_intron - 4), as in c4_viterbi16_kernel.h: an intron opens at open_il_pk, the
    // 3' site's length test is the counter's sign; what leaves the kernel (and what the dumps bring) is the length itself
    int open_il_pk, lim_pk, fifteen, at_pk[4], cv_pk[16];
    C16 col[NCOL][R], nbr[NCOL], expo, nx_carry;
    int prof_a[2];                                        // this lane's query profile entry of dense code 0, per window (Prof16, c4_ckpt16_kernel.h)
    const uint8_t *tdense;
    // the next column, from the wave's column stage (Stage16, c4_ckpt16_kernel.h): packed splice values, profile offsets of its two
    // codes and, fetched in the middle of a step, the profile entries (window A's NP ints, then window B's)
    int nx_sp4[4], nx_off[2], nx_prof[2 * P16::NP];
    int stage_a, stage_base, carry_a, carry_base;
    bool carry_cols;
    int corner_sc[2], corner_rq[2], corner_rt[2];
    bool corner_set[2];

    template <class Fn>
    __device__ __forceinline__ static void for_exported(Fn &&fn) {
        int slot = 0;
        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (exported(S)) { fn(S_, slot); slot += 3 + (live(S) ? 1 : 0); }
        });
    }
    __device__ __forceinline__ static void write_empty_column(int *colp) {
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            colp[slot] = NEG16; colp[slot + 1] = 0; colp[slot + 2] = 0;
            if constexpr (live(S)) colp[slot + 3] = 0;
        });
    }
    // the next carry column from the stage's carry planes; the 64 columns a chunk reads into them (WaveCK16)
    __device__ __forceinline__ void prefetch_carry() {
        const lds_int *p = lds_at(carry_a);
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            nx_carry.sc[S] = p[slot * ST::COLS];
            nx_carry.rq[S] = p[(slot + 1) * ST::COLS];
            nx_carry.rt[S] = p[(slot + 2) * ST::COLS];
            if constexpr (live(S)) nx_carry.il[S] = p[(slot + 3) * ST::COLS];
        });
        carry_a = ((carry_a + 4) & (ST::COLS * 4 - 1)) | carry_base;
    }
    __device__ __forceinline__ void fill_carry(lds_int *stage, int c0, const int *bnd_in) {
        const int c = c0 + lane;
        const int jx = c < 0 ? 0 : (c > Tm ? Tm : c);
        const int jc = carry_cols ? jx : 0;
        const int *g = bnd_in + (long long)jc * BND;
        int v[BND];
        static_for<BND>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; v[K] = g[K]; });
        lds_int *p = stage + ST::CARRY0 * ST::COLS + (c & (ST::COLS - 1));
        static_for<BND>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; p[K * ST::COLS] = v[K]; });
    }
    // this lane's rows of window H against every dense code (Prof16, c4_ckpt16_kernel.h)
    template <int H>
    __device__ __forceinline__ void build_profile(int i0) {
        using P16 = Prof16<R>;
        typedef __attribute__((address_space(3))) int lds_int;
        int qr[R];
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            const int i = i0 + RR;
            qr[RR] = 24 * ((i >= 1 && i <= Q[H]) ? (int)qc[H][q0[H] + i - 1] : 0);
        });
        for (int d = 0; d < P16::NCODE; d++) {
            const int code = tdense[24 + d];
            int v[R];
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_; v[RR] = kp->submat[qr[RR] + code]; });
            lds_int *p = (lds_int *)(size_t)(unsigned)(prof_a[H] + d * P16::CODE);
            static_for<P16::NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                p[K] = pk_pack(v[2 * K], v[2 * K + 1 < R ? 2 * K + 1 : 2 * K]);
            });
        }
    }
    __device__ __forceinline__ void prefetch_column() {
        const lds_int *p = lds_at(stage_a);
        static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_sp4[K] = p[K * ST::COLS]; });
        nx_off[0] = p[4 * ST::COLS]; nx_off[1] = p[5 * ST::COLS];
        stage_a = ((stage_a + 4) & (ST::COLS * 4 - 1)) | stage_base;
    }
    __device__ __forceinline__ void prefetch_profile() {
        const lds_int *pa = lds_at(prof_a[0] + nx_off[0]), *pb = lds_at(prof_a[1] + nx_off[1]);
        static_for<P16::NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_prof[K] = pa[K]; nx_prof[P16::NP + K] = pb[K]; });
    }
    // columns c0 + lane of both windows into the stage (WaveCK16::fill_stage)
    __device__ __forceinline__ void fill_stage(lds_int *stage, int c0) {
        constexpr int mat = F::match_at();
        const int c = c0 + lane;
        uint2 sv[2]; int off[2];
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            int ti = t0[H] + c - mat;
            ti = ti < 0 ? 0 : (ti > tlast[H] ? tlast[H] : ti);
            off[H] = (int)tc[H][(unsigned)ti] * P16::CODE;
            sv[H] = uint2{0u, 0u};
            if constexpr (F::has_splice()) {
                int tp = t0[H] + c - 2;
                tp = tp < 0 ? 0 : (tp > tlast[H] ? tlast[H] : tp);
                sv[H] = ss16[H][(unsigned)tp];
            }
        });
        lds_int *p = stage + (c & (ST::COLS - 1));
        p[0 * ST::COLS] = (int)__builtin_amdgcn_perm(sv[1].x, sv[0].x, 0x05040100u);
        p[1 * ST::COLS] = (int)__builtin_amdgcn_perm(sv[1].x, sv[0].x, 0x07060302u);
        p[2 * ST::COLS] = (int)__builtin_amdgcn_perm(sv[1].y, sv[0].y, 0x05040100u);
        p[3 * ST::COLS] = (int)__builtin_amdgcn_perm(sv[1].y, sv[0].y, 0x07060302u);
        p[4 * ST::COLS] = off[0]; p[5 * ST::COLS] = off[1];
    }

    // one cell of both windows; ipk / jpk: the cell's own row and column in both halves (the payload of a path that starts here)
    template <int RR, int PH, bool JINT>
    __device__ __forceinline__ void eval_cell(int j, int ms, const int (&sp)[4], int ipk, int jpk) {
        C16 &c = col[PH][RR];
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
            constexpr TrDesc t = M::tr[K];
            if constexpr (!inner(t.out)) return;                              // END, or a state that cannot reach the root
            static_assert(!(t.in == M::START && F::code(K) == 1), "a state's first transition is never the one out of START");
            constexpr int PD = (PH - t.at + NCOL) % NCOL;
            const C16 &src = (t.aq == 0) ? col[PD][RR] : (RR > 0 ? col[PD][RR > 0 ? RR - 1 : 0] : nbr[PD]);
            int cand, rqc, rtc, ilc = 0;
            if constexpr (t.in == M::START) {
                static_assert(t.in != M::START || (t.aq == 0 && t.at == 0 && t.calc < 0), "START leaves silently");
                cand = 0; rqc = ipk; rtc = jpk;                               // viterbi.c:403-412
            } else {
                cand = src.sc[t.in];
                rqc = src.rq[t.in];
                rtc = src.rt[t.in];
                if constexpr (t.calc >= 0) {
                    constexpr CalcDesc cd = M::calc[t.calc];
                    if constexpr (cd.kind == CALC_CONST) cand = pk_add<1>(cand, cv_pk[t.calc]);
                    else if constexpr (cd.kind >= CALC_MATCH_DNA && cd.kind <= CALC_MATCH_P2D) cand = pk_add<1>(cand, ms);
                    else if constexpr (cd.kind == CALC_SPLICE_PRE) cand = pk_add<1>(cand, sp[cd.param]);
                    else if constexpr (cd.kind == CALC_SPLICE_POST) {
                        static_assert(live(t.in), "post-splice calc without a length");
                        const int bad = pk_neg_mask(src.il[t.in], fifteen);            // length so far < min - at - 2: the counter's sign
                        const int sv = bfi32(bad, NEG16, sp[cd.param]);
                        cand = pk_add<1>(cand, sv);
                    }
                }
                if constexpr (!JINT && t.at > 0) cand = (j >= t.at) ? cand : NEG16;
                if constexpr (live(t.out)) {
                    if constexpr (F::owns_shadow(t.in, 0)) ilc = open_il_pk;
                    else if constexpr (live(t.in)) ilc = pk_add<1>(src.il[t.in], at_pk[t.at]);
                }
            }
            if constexpr (F::code(K) == 1) {                     // the first transition into this state assigns
                c.sc[t.out] = cand;
                c.rq[t.out] = rqc;
                c.rt[t.out] = rtc;
                if constexpr (live(t.out)) c.il[t.out] = ilc;
            } else {
                const int win = pk_lt_mask<1>(c.sc[t.out], cand, 0);                  // strict <: the newcomer wins
                c.rq[t.out] = bfi32(win, rqc, c.rq[t.out]);
                c.rt[t.out] = bfi32(win, rtc, c.rt[t.out]);
                if constexpr (live(t.out)) c.il[t.out] = bfi32(win, ilc, c.il[t.out]);
                c.sc[t.out] = pk_max<1>(c.sc[t.out], cand);
            }
        });
    }

    template <bool JINT, int PH>
    __device__ __forceinline__ void step(int s, int i0, bool last_strip, const int *bnd_in, int *bnd_out) {
        const int j = s - lane;
        int ms[R];
        static_for<P16::NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
            const int ea = nx_prof[K], eb = nx_prof[P16::NP + K];
            ms[2 * K] = (int)__builtin_amdgcn_perm((unsigned)eb, (unsigned)ea, 0x05040100u);
            if constexpr (2 * K + 1 < R) ms[2 * K + 1] = (int)__builtin_amdgcn_perm((unsigned)eb, (unsigned)ea, 0x07060302u);
        });
        int sp[4];
        static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; sp[K] = nx_sp4[K]; });
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            nbr[PH].sc[S] = dpp_shr1(nx_carry.sc[S], expo.sc[S]);
            nbr[PH].rq[S] = dpp_shr1(nx_carry.rq[S], expo.rq[S]);
            nbr[PH].rt[S] = dpp_shr1(nx_carry.rt[S], expo.rt[S]);
            if constexpr (live(S)) nbr[PH].il[S] = dpp_shr1(nx_carry.il[S], expo.il[S]);
        });
        prefetch_carry();
        prefetch_column();
        const int jpk = (j & 0xffff) | (j << 16);
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            const int i = i0 + RR;
            eval_cell<RR, PH, JINT>(j, ms[RR], sp, i | (i << 16), jpk);
            if constexpr (RR == (R + 1) / 2 - 1) prefetch_profile();       // the stage entry read above has arrived
        });
        // the window's first columns are the whole-rectangle pass's own cells, read from its dump; their payload is the
        // cell's identity (c4_viterbi_kernel.h, SEED 2).  Only in the steps that can hold those columns.
        if constexpr (!JINT) {
            if (s < DC + 64) {
                const bool early = (j >= 0) & (j < DC);
                const bool sd0 = early & seeded[0], sd1 = early & seeded[1];
                if (__builtin_amdgcn_ballot_w64(sd0 | sd1)) {
                    const int sdm = (sd0 ? 0x0000ffff : 0) | (sd1 ? (int)0xffff0000u : 0);
                    const int jc = j < 0 ? 0 : (j > DC - 1 ? DC - 1 : j);
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0 + RR;
                        int w[2][SEEDW];
                        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                            const int ic = i > seed_rows[H] - 1 ? seed_rows[H] - 1 : i;
                            const int *p = seed_rd[H] + ((long long)jc * seed_rows[H] + ic) * SEEDW;
                            static_for<SEEDW>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; w[H][K] = p[K]; });
                        });
                        const int ident = 0x8000 | (i & 0x7fff);
                        const int ident_pk = ident | (ident << 16);
                        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                            if constexpr (inner(S)) {
                                constexpr int hs = D16::half_of_sc(S);
                                const int v = (int)__builtin_amdgcn_perm((unsigned)w[1][hs / 2], (unsigned)w[0][hs / 2],
                                                                         (hs & 1) ? 0x07060302u : 0x05040100u);
                                col[PH][RR].sc[S] = bfi32(sdm, v, col[PH][RR].sc[S]);
                                if constexpr (live(S)) {
                                    constexpr int hl = D16::half_of_il(S);
                                    const int vl = (int)__builtin_amdgcn_perm((unsigned)w[1][hl / 2], (unsigned)w[0][hl / 2],
                                                                              (hl & 1) ? 0x07060302u : 0x05040100u);
                                    col[PH][RR].il[S] = bfi32(sdm, pk_sub(vl, lim_pk), col[PH][RR].il[S]);      // the dump holds the length
                                }
                                col[PH][RR].rq[S] = bfi32(sdm, ident_pk, col[PH][RR].rq[S]);
                                const int st = S * DC + jc;
                                col[PH][RR].rt[S] = bfi32(sdm, st | (st << 16), col[PH][RR].rt[S]);
                            }
                        });
                    });
                }
            }
        }
        // the bottom row for the lane below
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            expo.sc[S] = col[PH][R - 1].sc[S];
            expo.rq[S] = col[PH][R - 1].rq[S];
            expo.rt[S] = col[PH][R - 1].rt[S];
            if constexpr (live(S)) expo.il[S] = col[PH][R - 1].il[S];
        });
        if (!last_strip && lane == 63 && j >= 0 && j <= Tm) {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                int *p = bnd_out + (long long)j * BND + slot;
                p[0] = expo.sc[S];
                p[1] = expo.rq[S];
                p[2] = expo.rt[S];
                if constexpr (live(S)) p[3] = expo.il[S];
            });
        }
        // the corner cell (Q, T) of each window: score and payload of the requested state
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            if (s >= T[H]) {
                const bool mine = (j == T[H]) & (Q[H] >= i0) & (Q[H] < i0 + R);
                if (__builtin_amdgcn_ballot_w64(mine)) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        if (mine & (i0 + RR == Q[H])) {
                            static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                if constexpr (inner(S)) {
                                    if (final_state[H] == S) {
                                        corner_sc[H] = pk_half(col[PH][RR].sc[S], H);
                                        corner_rq[H] = (int)(((unsigned)col[PH][RR].rq[S] >> (16 * H)) & 0xffffu);
                                        corner_rt[H] = (int)(((unsigned)col[PH][RR].rt[S] >> (16 * H)) & 0xffffu);
                                        corner_set[H] = true;
                                    }
                                }
                            });
                        }
                    });
                }
            }
        });
    }

    // NW cooperating waves: wave `wid` runs the strips wid, wid + NW, ... of the window; a strip's carry row goes through the
    // workgroup's slab as with one wave, and a strip starts a chunk of steps once the strip above has finished the columns that
    // chunk reads (progress counters in LDS, prog[wave] = pass x PS + steps finished; c4_viterbi16_kernel.h, VAR 2).  The slab
    // of strip b + 1 is the one strip b - 1 wrote and strip b reads: b + 1 is at least 64 steps behind b, which has then read
    // (one step ahead) every column b + 1 overwrites.
    template <int NW>
    __device__ __forceinline__ void run(const DevJob &ja, const DevJob &jb, const DevSeqs &seqs, int *bnd, int wid, int *prog, lds_int *stage) {
        const DevJob *jp[2] = {&ja, &jb};
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            const DevJob &jx = *jp[H];
            Q[H] = jx.Q; T[H] = jx.T; q0[H] = jx.q0; t0[H] = jx.t0;
            tlast[H] = seqs.tlen[jx.pair] > 0 ? seqs.tlen[jx.pair] - 1 : 0;
            qc[H] = seqs.qcode + seqs.qoff[jx.pair];
            tc[H] = reinterpret_cast<const uint8_t *>(seqs.sub_rows) + seqs.toff[jx.pair];        // dense codes (Prof16)
            ss16[H] = F::has_splice() ? seqs.ss16 + seqs.toff[jx.pair] : nullptr;
            seeded[H] = jx.seed_off >= 0;
            seed_rd[H] = seqs.seed + (seeded[H] ? jx.seed_off : 0);
            seed_rows[H] = jx.seed_rows > 0 ? jx.seed_rows : 1;
            final_state[H] = jx.final_state;
            corner_sc[H] = LOW; corner_rq[H] = 0; corner_rt[H] = 0; corner_set[H] = false;
        });
        Qm = Q[0] > Q[1] ? Q[0] : Q[1]; Tm = T[0] > T[1] ? T[0] : T[1];
        static_for<M::NC>([&](auto CI_) __attribute__((always_inline)) { constexpr int CI = CI_;
            const int v = clamp16(kp->calc_value[CI]);
            cv_pk[CI] = pk_pack(v, v);
        });
        static_for<4>([&](auto A_) __attribute__((always_inline)) { constexpr int A = A_; at_pk[A] = pk_pack(A, A); });
        {
            const int lim = clamp16(kp->min_intron - 4);
            lim_pk = pk_pack(lim, lim);
            open_il_pk = pk_pack(-lim, -lim);
            fifteen = 0x000f000f;
        }
        const int nstrips = (Qm + 1 + W - 1) / W;
        const int nsteps = Tm + 64;
        const int main_lo = 63 + (MAXAT > DC ? MAXAT : DC), main_hi = Tm;
        const int nsteps_r = (nsteps + NCOL - 1) / NCOL * NCOL;
        const int main_lo_r = (main_lo + NCOL - 1) / NCOL * NCOL;
        constexpr int CHK = (63 / NCOL) * NCOL;             // steps per chunk: of the progress protocol, and between two refills of the stage
        const int PS = nsteps_r + 1;
        stage_base = lds_addr(stage);
        carry_base = stage_base + ST::CARRY0 * ST::COLS * 4;
        for (int b = wid; b < nstrips; b += NW) {
            const int i0 = b * W + lane * R;
            stage_a = stage_base + ((0 - lane) & (ST::COLS - 1)) * 4;
            carry_a = carry_base;
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_; build_profile<H>(i0); });
            static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                expo.sc[S] = NEG16; expo.il[S] = 0; expo.rq[S] = 0; expo.rt[S] = 0;
                static_for<NCOL>([&](auto D_) __attribute__((always_inline)) { constexpr int D = D_;
                    nbr[D].sc[S] = NEG16; nbr[D].il[S] = 0; nbr[D].rq[S] = 0; nbr[D].rt[S] = 0;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        col[D][RR].sc[S] = NEG16; col[D][RR].il[S] = 0; col[D][RR].rq[S] = 0; col[D][RR].rt[S] = 0;
                    });
                });
            });
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const bool first = (b == 0), last = (b == nstrips - 1);
            // slab layout: [empty column][carry row A (Tm + 1 columns)][carry row B]
            carry_cols = !first;
            const int *bnd_in = first ? bnd : bnd + BND + (long long)((b + 1) & 1) * (Tm + 1) * BND;
            int *bnd_out = bnd + BND + (long long)(b & 1) * (Tm + 1) * BND;
            auto group = [&](auto JI_, int s0) __attribute__((always_inline)) {
                constexpr bool JI = decltype(JI_)::value != 0;
                static_for<NCOL>([&](auto P_) __attribute__((always_inline)) { constexpr int P = P_;
                    step<JI, P>(s0 + P, i0, last, bnd_in, bnd_out);
                });
            };
            {
                const int above = (wid + NW - 1) % NW, above_base = ((b - 1) / NW) * PS, my_base = (b / NW) * PS;
                // the steps before c1 read carry columns up to c1 (one step ahead): written by the strip above in its step c1 + 63
                auto wait_above = [&](int c1) __attribute__((always_inline)) {
                    if constexpr (NW == 1) return;
                    if (first) return;
                    const int need = above_base + (c1 + 64 < nsteps_r ? c1 + 64 : nsteps_r);
                    while (__hip_atomic_load(prog + above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                };
                fill_stage(stage, -63);                    // columns -63 ... 0: what the first steps of the lanes read
                wait_above(CHK < nsteps_r ? CHK : nsteps_r);
                fill_carry(stage, -63, bnd_in);            // carry column 0
                prefetch_column();
                prefetch_profile();
                prefetch_carry();
                for (int c0 = 0; c0 < nsteps_r; c0 += CHK) {
                    const int c1 = c0 + CHK < nsteps_r ? c0 + CHK : nsteps_r;
                    fill_stage(stage, c0 + 1);             // columns c0 + 1 ... c0 + 64: what this chunk's steps read ahead
                    if (c0) wait_above(c1);
                    fill_carry(stage, c0 + 1, bnd_in);     // ... and the row above at those columns
                    int s = c0;
                    for (; s < main_lo_r && s < c1; s += NCOL) group(IC<0>{}, s);
                    for (; s + NCOL - 1 <= main_hi && s < c1; s += NCOL) group(IC<1>{}, s);
                    for (; s < c1; s += NCOL) group(IC<0>{}, s);
                    if constexpr (NW > 1) {
                        if (!last) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) __hip_atomic_store(prog + wid, my_base + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // carry row visible to the next strip
        }
    }
};

// The window chains of a pair of jobs (both with root ROOT) on one wave: results in the 32-bit window kernel's form — score
// of the chain's first corner, end_set, n_vsa = windows run, pad >= 0 with (qs, ts) where the chain found the start, pad < 0
// where the hop budget ran out.  job_lds[1] is an idle window when the pair holds one job.
template <class M, int R, int ROOT, int NW>
__device__ __forceinline__ void win16_chains(const KParams *kp_lds, const DevSeqs &seqs, DevJob *job_lds, int *more, int ia, int ib,
                                             DevResult *results, int *bnd, int *prog, int (*corner_lds)[4], int *prof_mem,
                                             const uint8_t *tdense, int *stage_mem) {
    using DP = WaveWin16<M, R, ROOT>;
    int hop = 0, first_score = 0;                      // threads 0 and 1: their window chain
    bool active = threadIdx.x < 2 && more[threadIdx.x & 1];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (;;) {
        DP dp{};                 // every member starts defined (c4_viterbi_kernel.h, viterbi_kernel)
        dp.kp = kp_lds;
        dp.lane = threadIdx.x & 63;
        dp.tdense = tdense;
        {
            typedef __attribute__((address_space(3))) int lds_int;
            for (int h = 0; h < 2; h++)
                dp.prof_a[h] = (int)(unsigned)(size_t)((lds_int *)prof_mem + (wid * 2 + h) * Prof16<R>::INTS) + dp.lane * Prof16<R>::EB;
        }
        if constexpr (NW > 1) {
            if (threadIdx.x < NW) prog[threadIdx.x] = 0;
            if (threadIdx.x < 2) corner_lds[threadIdx.x][3] = 0;
            __syncthreads();
        }
        dp.template run<NW>(job_lds[0], job_lds[1], seqs, bnd, wid, prog, (typename DP::lds_int *)stage_mem + wid * DP::ST::INTS);
        // the lane that owned a window's corner cell hands it to the thread that keeps that window's chain
        int sc[2], rq[2], rt[2];
        bool set[2];
        if constexpr (NW == 1) {
            for (int h = 0; h < 2; h++) {
                const unsigned long long owners = __ballot(dp.corner_set[h]);
                const int owner = owners ? __ffsll((long long)owners) - 1 : 0;
                sc[h] = __shfl(dp.corner_sc[h], owner); rq[h] = __shfl(dp.corner_rq[h], owner); rt[h] = __shfl(dp.corner_rt[h], owner);
                set[h] = owners != 0;
            }
            __syncthreads();
        } else {                                           // ... through LDS: the corner's strip ran on one of the waves
            for (int h = 0; h < 2; h++)
                if (dp.corner_set[h]) {
                    corner_lds[h][0] = dp.corner_sc[h]; corner_lds[h][1] = dp.corner_rq[h]; corner_lds[h][2] = dp.corner_rt[h];
                    corner_lds[h][3] = 1;
                }
            __syncthreads();
            for (int h = 0; h < 2; h++) {
                sc[h] = corner_lds[h][0]; rq[h] = corner_lds[h][1]; rt[h] = corner_lds[h][2]; set[h] = corner_lds[h][3] != 0;
            }
        }
        if (active) {
            const int h = threadIdx.x;
            DevJob &job = job_lds[h];
            if (hop == 0) first_score = sc[h];
            const bool ident = (rq[h] & 0x8000) != 0;
            bool go_on = false;
            DevResult res;
            res.flags = set[h] ? 0 : FLAG_NO_END; res.n_ops = 0; res.last_srp = 0; res.ops_off = 0;
            res.cell_size = DP::CS;
            for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
            res.score = first_score; res.end_set = set[h]; res.qe = job.Q; res.te = job.T; res.qs = 0; res.ts = 0;
            res.n_vsa = hop + 1; res.pad = ident ? -1 : 0;
            if (set[h] && !ident) {                                      // a real region start (window coordinates)
                res.qs = rq[h]; res.ts = rt[h] + job.win_t0w;
            } else if (set[h] && job.win_d >= 1 && hop + 1 < job.win_hops) {
                // entered through the dump: the next window ends in that cell and state
                constexpr int DC = DP::DC;
                const int row = rq[h] & 0x7fff, jc = rt[h] % DC, state = rt[h] / DC;
                const int d2 = job.win_d - 1, t0w2 = d2 >= 1 ? (d2 << job.seed_kshift) - (DC - 1) : 0;
                const int endcol = job.win_t0w + jc;
                job.Q = row; job.final_state = state;
                job.T = endcol - t0w2; job.t0 = job.win_t0_base + t0w2;
                job.seed_off = d2 >= 1 ? job.seed_base + (long long)(d2 - 1) * DC * job.seed_rows * DP::SEEDW : -1;
                job.win_d = d2; job.win_t0w = t0w2;
                go_on = true;
            }
            if (!go_on) {
                results[h ? ib : ia] = res;
                more[h] = 0;
                active = false;
                job.Q = 0; job.T = DP::DC; job.seed_off = -1;              // idles as a one-cell window
            }
        }
        hop++;
        __syncthreads();
        if (!(more[0] | more[1])) break;
        __syncthreads();
    }
}

// persistent waves; workgroup p of the queue runs the window chains of the p-th pair of the host's list (LaunchArgs::aux:
// two job indices with the same root, the second -1 where a job runs alone)
template <class M, int R, int WPE, int NW = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void win16_kernel(const KParams *kparams, DevSeqs seqs, const DevJob *jobs, const int *pairs, int n_pairs, DevResult *results,
                  DevScratch scratch, int *queue) {
    using RT = Roots<M>;
    // (the launch constants stay in memory: a strip reads them once, for its profile)
    // (one stage layout for every root)
    constexpr int STAGE_BND = WaveWin16<M, R, RT::disjoint() ? RT::root(0) : -1>::BND;
    static_assert(!RT::disjoint() || STAGE_BND == WaveWin16<M, R, RT::root(RT::count() - 1)>::BND, "one stage layout for every root");
    __shared__ __attribute__((aligned(512))) int stage_mem[NW * Stage16<STAGE_BND>::INTS];
    __shared__ int next_job;
    __shared__ DevJob job_lds[2];
    __shared__ int more[2];
    __shared__ int prog[NW];
    __shared__ int corner_lds[2][4];
    __shared__ __attribute__((aligned(16))) int prof_mem[NW * 2 * Prof16<R>::INTS];
    __shared__ uint8_t tdense_lds[32];
    if (threadIdx.x < 32) tdense_lds[threadIdx.x] = reinterpret_cast<const uint8_t *>(seqs.sub_colptr)[threadIdx.x];
    __syncthreads();
    const KParams *kp_lds = kparams;
    int *bnd = scratch.bnd + (long long)blockIdx.x * scratch.bnd_stride;
    // the empty column: every exported state unset, in every root's layout (they differ in which states they hold, not in
    // what an unset state looks like: score -32 768 per slot group of 3 or 4 ints would need the layout; instead every int of
    // the column is written per root below, before the root's first run)
    for (;;) {
        if (threadIdx.x == 0) next_job = atomicAdd(queue, 1);
        __syncthreads();
        const int pid = __builtin_amdgcn_readfirstlane(next_job);      // wave-uniform: job descriptions and pointers in scalar registers
        __syncthreads();
        if (pid >= n_pairs) break;
        const int ia = pairs[2 * pid], ib = pairs[2 * pid + 1];
        if (threadIdx.x < 2) {
            const int h = threadIdx.x;
            const bool real = h == 0 || ib >= 0;
            job_lds[h] = jobs[real ? (h ? ib : ia) : ia];
            more[h] = real ? 1 : 0;
            if (!real) { job_lds[h].Q = 0; job_lds[h].T = M::MAXAT; job_lds[h].seed_off = -1; }
        }
        __syncthreads();
        const int root = job_lds[0].root;                   // both jobs of a pair have it
        bool ran = false;
        if constexpr (RT::disjoint()) {
            static_for<RT::count()>([&](auto X_) __attribute__((always_inline)) { constexpr int X = X_;
                constexpr int ROOT = RT::root(X);
                if (!ran && root == ROOT) {
                    if (threadIdx.x == 0) WaveWin16<M, R, ROOT>::write_empty_column(bnd);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __syncthreads();
                    win16_chains<M, R, ROOT, NW>(kp_lds, seqs, job_lds, more, ia, ib, results, bnd, prog, corner_lds, prof_mem, tdense_lds, stage_mem);
                    ran = true;
                }
            });
        }
        if constexpr (!RT::disjoint()) {                    // one component: every inner state
            if (threadIdx.x == 0) WaveWin16<M, R, -1>::write_empty_column(bnd);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            win16_chains<M, R, -1, NW>(kp_lds, seqs, job_lds, more, ia, ib, results, bnd, prog, corner_lds, prof_mem, tdense_lds, stage_mem);
        } else if (!ran) {                                  // a root the model does not have: the host's mistake, say so
            if (threadIdx.x < 2 && more[threadIdx.x]) {
                DevResult res;
                res.flags = FLAG_NO_END; res.n_ops = 0; res.last_srp = 0; res.ops_off = 0; res.cell_size = 0; res.score = LOW;
                res.end_set = 0; res.qe = res.te = res.qs = res.ts = 0; res.n_vsa = 0; res.pad = -1;
                for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
                results[threadIdx.x ? ib : ia] = res;
            }
        }
        __syncthreads();
    }
}

}  // namespace c4k
