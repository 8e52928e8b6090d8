// c4_ckpt16_kernel.h — the FIND_CHECKPOINTS continuation pass (viterbi.c:605-631, optimal.c:160-230) with TWO jobs per
// lane in packed 16-bit halves: the form the score pass took in c4_viterbi16_kernel.h, now with the two payloads a
// checkpoint cell carries beside its score.
//
// What a cell holds per state, low half = job A, high half = job B:
//   sc   the score (v_pk_add_i16 clamp / v_pk_max_i16; "unset" = -32 768, adds saturate);
//   srp  the checkpoint payload ((row x states) + state) x max_target_advance + k of viterbi.c:515-522 — an unsigned
//        16-bit number (queries up to 3 275 rows for est2genome), moved with the winner by v_bfi_b32;
//   il   for the states whose intron-start shadow something can still read (the two intron states of est2genome): the
//        length of the open intron so far, the saturating counter of the packed score pass; shadow = column - il - 2.
// A transition is: candidate = source + calc (one packed add), mask = candidate beats the holder (v_pk_sub_i16 clamp +
// v_pk_ashrrev_i16, strict <: the first transition into a state assigns, later ones replace on strict <,
// viterbi.c:766-775), score = packed max, payloads = v_bfi_b32 under the mask — six instructions for two cells where
// the 32-bit kernel spends add, compare and one select per slot on one.
//
// Which jobs: top-level checkpoint passes of Optimal_find_path_reduced_space (first state START with the zero cell,
// final state END: what fused_reduced_paths launches) under the continuation kernels' row-0 shortcut (CONT && LOCAL of
// c4_viterbi_kernel.h).  START is valid in the origin cell only and END in the far corner only (CORNER scopes,
// viterbi.c:68-76): the START transitions are evaluated in the one cell that can hold the origin, END is evaluated once
// per job, in 32-bit arithmetic on the halves of the corner cell, and neither state occupies registers in the column loop.
//
// Exactness.  Every cell ON the optimal path holds its reference value: the path is a local optimum, so every prefix
// of it scores at least minus one gap/intron opening (dropping a negative prefix would score higher), and a prefix is
// at most the best score; both are far inside 16 bits under the host's guard ((Q + 1) x largest substitution score
// plus what introns can gain <= 16 000).  A cell OFF the path may saturate at -32 768 (its reference value lies below
// that); a saturated or unset value plus everything a path can gain stays below -16 000 and loses against the real
// candidate of every path cell, exactly as the -987 654 321 candidates of the 32-bit kernels do.  Winner, payload and
// tie-break of every path cell are therefore the reference's, which is all the checkpoint traceback reads
// (viterbi.c:537-601 follows the payloads from the corner cell; cells it does not visit never leave the kernel:
// calls whose checkpoint cells go to the caller keep the 32-bit kernels).
// The length counter saturates at 32 767 columns: an intron that long passes the minimum test either way, the maximum
// cannot fail (T + 4 <= max_intron, checked by the host), and the shadow written into a checkpoint cell is then
// "at least 32 767 columns back" — the host and the stitch kernel compare such shadows as equivalent (shadow_equiv in
// c4_engine.hip): a continuation seeded with either computes the same scores and the same path.
#pragma once
#include "c4_viterbi16_kernel.h"

namespace c4k {

// (mask & a) | (~mask & b) as ONE instruction (v_bitop3_b32 with the bit-select truth table): written with & | ~ the compiler
// merges the masks of a chain of selects and shares them between the payloads of a state, which comes to three logic
// instructions per select instead of one (117 against 64 per step of the packed region windows at four rows per lane)
__device__ __forceinline__ int bfi32(int mask, int a, int b) {
    return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA);
}

// ROOT: the state the path's END is entered from where the caller knows it (the region pass reports it), -1 otherwise.  The
// pass then computes the states that can reach ROOT and no others (Roots, c4_viterbi16_kernel.h): the checkpoint pass runs
// from the region's start to its end cell, the optimal path there is the one the region pass found, its END is entered
// from ROOT (a corner-to-corner path of another component is also a local path ending in that cell, so it scores no more
// than that component's state did in the region pass, which lost against — or, behind it in transition order, did not beat —
// ROOT's), and every cell, payload and checkpoint the traceback follows lies in ROOT's component.
// The substitution scores of a lane's R query rows against each residue code of the launch's targets as a QUERY PROFILE in LDS
// (round 5; the packed score pass has had one since round 4, c4_viterbi16_kernel.h IO 1): one entry of NP ints per (job, code,
// lane) -- two rows' scores per int, exactly the halves pk_pack(submat[row code x 24 + target code], ...) made per step --, rebuilt
// per strip.  A step reads one entry per job (the bank depends on the lane only: conflict-free) where it read 2 R words of the
// 24 x 24 table at computed addresses (56-64 % of the LDS-active cycles of these two passes were bank conflicts,
// profiles/r04_i_sq.csv).  The targets arrive as DENSE codes (0 .. 7: ResidentSeqs::tcode_dense, built at staging where the
// batch's targets hold at most eight residue codes; the host takes the packed checkpoint pass and windows only then), handed over
// in DevSeqs::sub_rows, which no packed kernel reads otherwise; dense index d stands for row code tdense[24 + d] of the matrix.
template <int R>
struct Prof16 {
    static constexpr int NCODE = 8;
    static constexpr int NP = (R + 1) / 2, EB = NP == 3 ? 12 : 8;        // ints / bytes per entry (8 bytes also where one int is used)
    static constexpr int CODE = 64 * EB;                  // bytes per code: the 64 lanes' entries, contiguous
    static constexpr int INTS = NCODE * CODE / 4;         // ints per (wave, job)
};

// The column inputs of a step as a per-wave STAGE in LDS (round 6; the packed score pass has had one since round 4,
// c4_viterbi16_kernel.h IO 1): 128 columns x 6 planes of one int -- the four splice values of both jobs already interleaved into
// packed halves, and the byte offsets of the two dense residue codes into the query profile --, plane-major (the lanes of a wave
// read consecutive columns: 64 consecutive words, no bank conflict).  The wave refills it once per chunk of 63 steps with 64
// coalesced, clamped columns (one 8-byte and one 1-byte load per job and CHUNK where a step issued them per STEP, with their
// clamps, 64-bit address arithmetic and four v_perm: 24 of ~200 VALU instructions of a step, four of its six vector memory
// instructions and the wait for them at the top of every step), and a step reads its column with three ds_read2st64.
// The strip carry row rides in BND more planes of the same stage: the row above still travels through the workgroup's slab in
// memory (written by lane 63 of the strip above, a store nobody waits for), but the strip below fetches the 64 columns a chunk
// reads with ONE coalesced load per lane once the strip above has published them, and a step reads its column from LDS at a
// wave-uniform address (a broadcast) -- the column loop issues no vector memory load at all.
template <int BND>
struct Stage16 {
    static constexpr int COLS = 128, CARRY0 = 6, PLANES = 6 + BND, INTS = COLS * PLANES;     // planes of 512 bytes, 512-byte aligned: a running address wraps with one v_and_or
};

template <class M, int R, int ROOT = -1>
struct WaveCK16 {
    using F = Facts<M>;
    using RT = Roots<M>;
    using W32 = WaveDP<M, R, MODE_CKPT, true, true>;
    static constexpr int NS = M::NS, NCOL = M::MAXAT + 1, W = 64 * R, MAXAT = M::MAXAT;
    static constexpr int CS = W32::CS;                    // the reference's cell: score, designations, checkpoint slot
    static constexpr bool live(int s) { return M::NDES > 0 && W32::slot_live(s, 0); }
    static constexpr bool inner(int s) { return RT::member(ROOT, s); }       // the states this pass computes
    static constexpr bool exported(int s) { return inner(s) && F::exported(s); }
    static constexpr int n_live() { int n = 0; for (int s = 0; s < NS; s++) n += (inner(s) && live(s)); return n; }
    static constexpr int n_inner() { int n = 0; for (int s = 0; s < NS; s++) n += inner(s); return n; }
    // a checkpoint row in the job's slab: one word (score | payload << 16) per inner state, then the live lengths two per word
    static constexpr int word_of(int s) { int n = 0; for (int x = 0; x < s; x++) n += inner(x); return n; }
    static constexpr int live_index(int s) { int n = 0; for (int x = 0; x < s; x++) n += (inner(x) && live(x)); return n; }
    static constexpr int CKW = n_inner() + (n_live() + 1) / 2;
    static constexpr int n_exp() { int n = 0; for (int s = 0; s < NS; s++) n += exported(s); return n; }
    static constexpr int n_exp_live() { int n = 0; for (int s = 0; s < NS; s++) n += (exported(s) && live(s)); return n; }
    static constexpr int BND = n_exp() * 2 + n_exp_live();   // ints per column between strips
    static_assert(!F::has_phase(), "split-codon calcs are not packed");
    static_assert(M::NDES <= 1, "one shadow designation");
    static_assert(M::START == 0 && M::END == 1, "state numbering of the closed model");
    static_assert(!F::exported(M::START), "START advances nothing");
    struct C16 { int sc[NS]; int il[NS]; int srp[NS]; };
    typedef __attribute__((address_space(3))) int lds_int;
    __device__ __forceinline__ static lds_int *lds_at(int a) { return (lds_int *)(size_t)(unsigned)a; }
    __device__ __forceinline__ static int lds_addr(const lds_int *p) { return (int)(unsigned)(size_t)p; }
    using P16 = Prof16<R>;
    using ST = Stage16<BND>;

    const KParams *kp;
    int lane;
    const uint8_t *qc[2], *tc[2];
    const uint2 *ss16[2];
    int Q[2], T[2], q0[2], t0[2], tlast[2], cp_count[2], section[2];
    int *ckp[2];                                          // each job's checkpoint rows
    int cp_next_j[2], cp_next_i[2];
    int cp_phase[2];                                      // wave-uniform: (step + MAXAT - 1) mod section, see step()
    int Qm, Tm, Tmin;
    // the intron length counter is kept minus (min_intron - 4), as in c4_viterbi16_kernel.h: an intron opens at open_il_pk, the
    // 3' site's length test is the counter's sign; what leaves the kernel (and what the dumps bring) is the length itself
    int open_il_pk, lim_pk, fifteen, at_pk[4], cv_pk[16];
    C16 col[NCOL][R], nbr[NCOL], expo, nx_carry;
    int prof_a[2];                                        // LDS byte address of this lane's profile entry of dense code 0, per job
    const uint8_t *tdense;                                // the launch's code table: [24 + d] = matrix row code of dense index d
    // the next column, from the stage: packed splice values, the profile byte offsets of its two codes, and (fetched in the
    // middle of a step, when those have arrived) the profile entries: job A's NP ints, then job B's
    int nx_sp4[4], nx_off[2], nx_prof[2 * P16::NP];
    int stage_a, stage_base;                              // LDS byte address of the next column's stage entry; of the wave's stage
    int carry_a, carry_base;                              // ... of the next carry column's first plane (wave-uniform); of that plane
    bool carry_cols;
    int corner_sc[2], corner_srp[2];
    bool corner_set[2];

    template <class Fn>
    __device__ __forceinline__ static void for_exported(Fn &&fn) {
        int slot = 0;
        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (exported(S)) { fn(S_, slot); slot += 2 + (live(S) ? 1 : 0); }
        });
    }
    __device__ __forceinline__ static void write_empty_column(int *colp) {
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            colp[slot] = NEG16; colp[slot + 1] = 0;
            if constexpr (live(S)) colp[slot + 2] = 0;
        });
    }
    // the next carry column, from the stage's carry planes (steps follow each other: a running, wave-uniform address)
    __device__ __forceinline__ void prefetch_carry() {
        const lds_int *p = lds_at(carry_a);
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            nx_carry.sc[S] = p[slot * ST::COLS];
            nx_carry.srp[S] = p[(slot + 1) * ST::COLS];
            if constexpr (live(S)) nx_carry.il[S] = p[(slot + 2) * ST::COLS];
        });
        carry_a = ((carry_a + 4) & (ST::COLS * 4 - 1)) | carry_base;
    }
    // carry columns c0 + lane (clamped to the row's columns as the per-step loads were; the first strip: the empty column) into
    // the stage -- after the strip above has published them
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
    // this lane's rows i0 .. i0 + R - 1 of job H against every dense code: the halves step() used to build from the matrix per
    // step (rows outside the job score as row code 0 did: they feed nothing a result reads)
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
    // the next column's entry of the stage (a lane's columns follow each other: a running address)
    __device__ __forceinline__ void prefetch_column() {
        const lds_int *p = lds_at(stage_a);
        static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_sp4[K] = p[K * ST::COLS]; });
        nx_off[0] = p[4 * ST::COLS]; nx_off[1] = p[5 * ST::COLS];
        stage_a = ((stage_a + 4) & (ST::COLS * 4 - 1)) | stage_base;
    }
    // the profile entries of the next column's codes
    __device__ __forceinline__ void prefetch_profile() {
        const lds_int *pa = lds_at(prof_a[0] + nx_off[0]), *pb = lds_at(prof_a[1] + nx_off[1]);
        static_for<P16::NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_prof[K] = pa[K]; nx_prof[P16::NP + K] = pb[K]; });
    }
    // columns c0 + lane of both jobs into the stage: clamped as the per-step loads were, the splice values of the two jobs
    // interleaved into packed halves, the dense residue codes as profile offsets
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

    // one cell of both jobs; ORIGIN: this instantiation can hold the origin cell (row 0 of the lane, steps before the
    // main loop), the only cell a transition out of START is valid in
    template <int RR, int PH, bool JINT>
    __device__ __forceinline__ void eval_cell(int j, bool origin, int ms, const int (&sp)[4]) {
        C16 &c = col[PH][RR];
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
            constexpr TrDesc t = M::tr[K];
            if constexpr (!inner(t.out)) return;                              // END: once per job, in the corner cell (step); or a state
                                                                              // that cannot reach the root
            if constexpr (t.in == M::START && (RR > 0 || JINT)) return;       // cannot be the origin cell
            static_assert(!(t.in == M::START && F::code(K) == 1), "a state's first transition is never the one out of START");
            constexpr int PD = (PH - t.at + NCOL) % NCOL;
            const C16 &src = (t.aq == 0) ? col[PD][RR] : (RR > 0 ? col[PD][RR > 0 ? RR - 1 : 0] : nbr[PD]);
            int cand, srpc = 0, ilc = 0;
            if constexpr (t.in == M::START) {
                static_assert(t.in != M::START || (t.aq == 0 && t.at == 0 && t.calc < 0), "START leaves silently");
                cand = origin ? 0 : NEG16;                                    // the zero cell (viterbi.c:705-714)
            } else {
                cand = src.sc[t.in];
                srpc = src.srp[t.in];
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
                c.srp[t.out] = srpc;
                if constexpr (live(t.out)) c.il[t.out] = ilc;
            } else {
                const int win = pk_lt_mask<1>(c.sc[t.out], cand, 0);                  // strict <: the newcomer wins
                c.srp[t.out] = bfi32(win, srpc, c.srp[t.out]);
                if constexpr (live(t.out)) c.il[t.out] = bfi32(win, ilc, c.il[t.out]);
                c.sc[t.out] = pk_max<1>(c.sc[t.out], cand);
            }
        });
    }

    template <bool JINT, int PH>
    __device__ __forceinline__ void step(int s, int i0, bool first_strip, bool last_strip, const int *bnd_in, int *bnd_out) {
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
            nbr[PH].srp[S] = dpp_shr1(nx_carry.srp[S], expo.srp[S]);
            if constexpr (live(S)) nbr[PH].il[S] = dpp_shr1(nx_carry.il[S], expo.il[S]);
        });
        prefetch_carry();
        prefetch_column();
        const bool origin = first_strip & (lane == 0) & (s == 0);
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            eval_cell<RR, PH, JINT>(j, origin, ms[RR], sp);
            // half-way through the step the stage entry read above has arrived; the profile entries it points to are then there
            // when the next step starts
            if constexpr (RR == (R + 1) / 2 - 1) prefetch_profile();
        });
        // the bottom row for the lane below, BEFORE any checkpoint edit (that lane still needs column j as it was)
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            expo.sc[S] = col[PH][R - 1].sc[S];
            expo.srp[S] = col[PH][R - 1].srp[S];
            if constexpr (live(S)) expo.il[S] = col[PH][R - 1].il[S];
        });
        if (!last_strip && lane == 63 && j >= 0 && j <= Tm) {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                int *p = bnd_out + (long long)j * BND + slot;
                p[0] = expo.sc[S];
                p[1] = expo.srp[S];
                if constexpr (live(S)) p[2] = expo.il[S];
            });
        }
        // the corner cell (Q, T) of each job: END is entered here and nowhere else (viterbi.c:813-832).  Its
        // transitions in id order on the 32-bit halves: the first assigns, later ones replace on strict <.
        if (s >= Tmin) {
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                const bool mine = (j == T[H]) & (Q[H] >= i0) & (Q[H] < i0 + R);
                if (__builtin_amdgcn_ballot_w64(mine)) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        if (mine & (i0 + RR == Q[H])) {
                            bool set = false;
                            int e_sc = 0, e_srp = 0;
                            static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                                constexpr TrDesc t = M::tr[K];
                                if constexpr (t.out == M::END && inner(t.in)) {
                                    static_assert(t.out != M::END || (t.aq == 0 && t.at == 0 && t.calc < 0 && t.in != M::START),
                                                  "END is entered silently from an inner state");
                                    const int v = pk_half(col[PH][RR].sc[t.in], H);
                                    const int p = (int)(((unsigned)col[PH][RR].srp[t.in] >> (16 * H)) & 0xffffu);
                                    const bool win = !set | (e_sc < v);
                                    e_sc = win ? v : e_sc;
                                    e_srp = win ? p : e_srp;
                                    set = true;
                                }
                            });
                            corner_sc[H] = e_sc; corner_srp[H] = e_srp; corner_set[H] = set;
                        }
                    });
                }
            });
        }
        // checkpoint rows (Viterbi_Checkpoint_process, viterbi.c:605-631): at checkpoint column c the reference copies rows
        // c, c-1, .. and then stamps their payload slots.  Each of those columns is copied out at the step that computes it
        // (c4_viterbi_kernel.h, step (7)); the stamp stays at column c.  The two jobs have their own columns.
        // A lane stands on a column a checkpoint keeps when a multiple of the job's section length lies in [j, j + MAXAT - 1];
        // some lane of the wave does when one lies in [s - 63, s + MAXAT - 1], i.e. when (s + MAXAT - 1) mod section <= 63 + MAXAT - 1:
        // a scalar test (cp_phase is kept per step) in front of the per-lane tests, ballots and stores below -- they were
        // a sixth of the instructions of a step and apply to 65 steps in every section.
        const bool cp_near = (section[0] <= 0) | (cp_phase[0] <= 63 + MAXAT - 1) | (section[1] <= 0) | (cp_phase[1] <= 63 + MAXAT - 1);
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            const int nx = cp_phase[H] + 1;
            cp_phase[H] = nx >= section[H] ? nx - section[H] : nx;
        });
        if (cp_near) {
            unsigned stamp = 0;
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                const bool cp_live = (j >= 0) & (j <= T[H]) & (cp_next_i[H] < cp_count[H]);
                const unsigned ahead = (unsigned)(cp_next_j[H] - j);            // 0 .. MAXAT-1: a column the checkpoint keeps
                if (__builtin_amdgcn_ballot_w64(cp_live & (ahead < (unsigned)MAXAT))) {
                    if (cp_live & (ahead < (unsigned)MAXAT)) {
                        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                            const int i = i0 + RR;
                            if (i <= Q[H]) {
                                int *p = ckp[H] + ((((long long)cp_next_i[H] * MAXAT + ahead) * (Q[H] + 1) + i)) * CKW;
                                constexpr unsigned sel = H ? 0x07060302u : 0x05040100u;
                                int lw[(n_live() + 1) / 2 > 0 ? (n_live() + 1) / 2 : 1] = {0};
                                static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                    if constexpr (inner(S)) {
                                        p[word_of(S)] = (int)__builtin_amdgcn_perm((unsigned)col[PH][RR].srp[S], (unsigned)col[PH][RR].sc[S], sel);
                                        if constexpr (live(S)) {
                                            constexpr int li = live_index(S);
                                            const unsigned h = ((unsigned)pk_add<1>(col[PH][RR].il[S], lim_pk) >> (16 * H)) & 0xffffu;    // the row holds the length
                                            lw[li / 2] |= (int)(h << (16 * (li & 1)));
                                        }
                                    }
                                });
                                static_for<(n_live() + 1) / 2>([&](auto L_) __attribute__((always_inline)) { constexpr int L = L_; p[n_inner() + L] = lw[L]; });
                            }
                        });
                    }
                    stamp |= (cp_live & (ahead == 0u)) ? (H ? 0xffff0000u : 0x0000ffffu) : 0u;
                }
            });
            if (__builtin_amdgcn_ballot_w64(stamp != 0u)) {
                // the stamps only depend on the lane's rows: keep the compiler from hoisting them out of the column loop
                int i0v = i0;
                asm volatile("" : "+v"(i0v));
                static_for<MAXAT>([&](auto ROW_) __attribute__((always_inline)) { constexpr int ROW = ROW_;
                    constexpr int PR = (PH - ROW + NCOL) % NCOL;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int rowid = (i0v + RR) * (NS * MAXAT);
                        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                            if constexpr (inner(S)) {
                                const int id = (rowid + S * MAXAT + ROW) & 0xffff;              // viterbi.c:515-522 (the full model's state count)
                                col[PR][RR].srp[S] = bfi32((int)stamp, id | (id << 16), col[PR][RR].srp[S]);
                            }
                        });
                    });
                    // our copies of row i0-1 at these columns get the same edit
                    const int rowid = (i0v - 1) * (NS * MAXAT);
                    static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                        if constexpr (exported(S)) {
                            const int id = (rowid + S * MAXAT + ROW) & 0xffff;
                            nbr[PR].srp[S] = bfi32((int)stamp, id | (id << 16), nbr[PR].srp[S]);
                        }
                    });
                });
                static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                    const bool mine = (stamp & (H ? 0xffff0000u : 0x0000ffffu)) != 0u;
                    cp_next_i[H] += mine ? 1 : 0;
                    cp_next_j[H] += mine ? section[H] : 0;
                });
            }
        }
    }

    // NW cooperating waves: wave `wid` runs the strips wid, wid + NW, ...; carry rows through the workgroup's slab, a strip starts a
    // chunk of steps once the strip above has finished the columns it reads (progress counters in LDS: c4_win16_kernel.h, run)
    template <int NW>
    __device__ __forceinline__ void run(const DevJob &ja, const DevJob &jb, const DevSeqs &seqs, int *bnd, int *ckpt_a, int *ckpt_b,
                                        int wid, int *prog, lds_int *stage) {
        const DevJob *jp[2] = {&ja, &jb};
        ckp[0] = ckpt_a; ckp[1] = ckpt_b;
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            const DevJob &jx = *jp[H];
            Q[H] = jx.Q; T[H] = jx.T; q0[H] = jx.q0; t0[H] = jx.t0;
            tlast[H] = seqs.tlen[jx.pair] > 0 ? seqs.tlen[jx.pair] - 1 : 0;
            qc[H] = seqs.qcode + seqs.qoff[jx.pair];
            tc[H] = reinterpret_cast<const uint8_t *>(seqs.sub_rows) + seqs.toff[jx.pair];        // dense codes (Prof16)
            ss16[H] = F::has_splice() ? seqs.ss16 + seqs.toff[jx.pair] : nullptr;
            cp_count[H] = jx.cp_count;
            section[H] = jx.T / (jx.cp_count + 1);
            corner_sc[H] = LOW; corner_srp[H] = 0; corner_set[H] = false;
        });
        Qm = Q[0] > Q[1] ? Q[0] : Q[1]; Tm = T[0] > T[1] ? T[0] : T[1]; Tmin = T[0] < T[1] ? T[0] : T[1];
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
        const int main_lo = 63 + MAXAT, main_hi = Tm;
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
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                build_profile<H>(i0);
                cp_next_j[H] = section[H] > 0 ? section[H] : 0x7fffffff; cp_next_i[H] = 0;
                cp_phase[H] = section[H] > 0 ? (MAXAT - 1) % section[H] : 0;
            });
            static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                expo.sc[S] = NEG16; expo.il[S] = 0; expo.srp[S] = 0;
                static_for<NCOL>([&](auto D_) __attribute__((always_inline)) { constexpr int D = D_;
                    nbr[D].sc[S] = NEG16; nbr[D].il[S] = 0; nbr[D].srp[S] = 0;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        col[D][RR].sc[S] = NEG16; col[D][RR].il[S] = 0; col[D][RR].srp[S] = 0;
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
                    step<JI, P>(s0 + P, i0, first, last, bnd_in, bnd_out);
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

    // Viterbi_Checkpoint_traceback (viterbi.c:537-601) over the packed rows of one job; the list is written last section
    // first, cells in the reference's layout (score, intron-start shadow, payload)
    struct Cell3 { int sc, shadow, srp; };
    __device__ static Cell3 cell_at(const int *ck, const DevJob &job, int section_length, int cp, int row, int qpos, int state) {
        const int *p = ck + (((long long)cp * MAXAT + row) * (job.Q + 1) + qpos) * CKW;
        Cell3 c; c.sc = 0; c.shadow = 0; c.srp = 0;
        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (inner(S)) {
                if (state == S) {
                    const unsigned w = (unsigned)p[word_of(S)];
                    c.sc = (int)(short)(w & 0xffffu);
                    c.srp = (int)(w >> 16);
                    if constexpr (live(S)) {
                        constexpr int li = live_index(S);
                        const int il = (int)(((unsigned)p[n_inner() + li / 2] >> (16 * (li & 1))) & 0xffffu);
                        c.shadow = job.t0 + (section_length * (cp + 1) - row) - il - 2;
                    }
                }
            }
        });
        return c;
    }
    __device__ __noinline__ static void checkpoint_traceback(const int *ck, const DevJob &job, DevVsa *vsa, DevResult &res) {
        const int q0 = job.q0, t0 = job.t0;
        const int cpn = job.cp_count, section_length = job.T / (cpn + 1);
        auto decode = [&](int srp, int &state, int &row, int &pos) {
            row = srp % MAXAT;
            const int rem = srp / MAXAT;
            state = rem % NS;
            pos = rem / NS;
        };
        int state, row, pos, n = 0;
        decode(res.last_srp, state, row, pos);
        int query_start = q0 + pos, target_start = t0 + section_length * cpn - row;
        DevVsa v;
        v.qs = query_start; v.ts = target_start;
        v.ql = (q0 + job.Q) - query_start; v.tl = (t0 + job.T) - target_start;
        v.first_state = state;
        for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = l < CS ? res.final_cell[l] : 0;
        vsa[n++] = v;
        for (int c = cpn - 1; c >= 1; c--) {
            const DevVsa p = v;
            const int prev_row = row;
            const Cell3 cell = cell_at(ck, job, section_length, c, prev_row, p.qs - q0, p.first_state);
            decode(cell.srp, state, row, pos);
            query_start = q0 + pos;
            target_start = p.ts - section_length - row + prev_row;
            v.qs = query_start; v.ts = target_start; v.ql = p.qs - query_start; v.tl = p.ts - target_start;
            v.first_state = state;
            for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = 0;
            v.final_cell[0] = cell.sc; v.final_cell[1] = cell.shadow; v.final_cell[CS - 1] = cell.srp;
            vsa[n++] = v;
        }
        {
            const DevVsa p = v;
            const Cell3 cell = cell_at(ck, job, section_length, 0, row, p.qs - q0, p.first_state);
            v.qs = q0; v.ts = t0; v.ql = query_start - q0; v.tl = target_start - t0;
            v.first_state = job.first_state;
            for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = 0;
            v.final_cell[0] = cell.sc; v.final_cell[1] = cell.shadow; v.final_cell[CS - 1] = cell.srp;
            vsa[n++] = v;
        }
        res.n_vsa = n;
    }
};

// One pair of jobs (both with root ROOT; jb = ja where the pair holds one job) on one wave
template <class M, int R, int ROOT, int NW>
__device__ __forceinline__ void ckpt16_pair(const KParams *kp_lds, const DevSeqs &seqs, const DevJob *jobs, int ia, int ib,
                                            DevResult *results, DevVsa *vsas, int *bnd, int *ck_a, int *ck_b, int *prog,
                                            int (*corner_lds)[4], int *prof_mem, const uint8_t *tdense, int *stage_mem) {
    using DP = WaveCK16<M, R, ROOT>;
    if (threadIdx.x == 0) DP::write_empty_column(bnd);
    if constexpr (NW > 1) {
        if (threadIdx.x < NW) prog[threadIdx.x] = 0;
        if (threadIdx.x < 2) corner_lds[threadIdx.x][3] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    DP dp{};                 // every member starts defined (c4_viterbi_kernel.h, viterbi_kernel)
    dp.kp = kp_lds;
    dp.lane = threadIdx.x & 63;
    dp.tdense = tdense;
    {
        typedef __attribute__((address_space(3))) int lds_int;
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        for (int h = 0; h < 2; h++)
            dp.prof_a[h] = (int)(unsigned)(size_t)((lds_int *)prof_mem + (w * 2 + h) * Prof16<R>::INTS) + dp.lane * Prof16<R>::EB;
    }
    {
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        dp.template run<NW>(jobs[ia], jobs[ib], seqs, bnd, ck_a, ck_b, w, prog, (typename DP::lds_int *)stage_mem + w * DP::ST::INTS);
    }
    // the lane that owned a job's corner cell hands it to the lane that walks the job's checkpoints
    int sc[2], srp[2];
    bool set[2];
    if constexpr (NW == 1) {
        for (int h = 0; h < 2; h++) {
            const unsigned long long owners = __ballot(dp.corner_set[h]);
            const int owner = owners ? __ffsll((long long)owners) - 1 : 0;
            sc[h] = __shfl(dp.corner_sc[h], owner); srp[h] = __shfl(dp.corner_srp[h], owner);
            set[h] = owners != 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
    } else {                                               // ... through LDS: the corner's strip ran on one of the waves
        for (int h = 0; h < 2; h++)
            if (dp.corner_set[h]) { corner_lds[h][0] = dp.corner_sc[h]; corner_lds[h][1] = dp.corner_srp[h]; corner_lds[h][3] = 1; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        for (int h = 0; h < 2; h++) { sc[h] = corner_lds[h][0]; srp[h] = corner_lds[h][1]; set[h] = corner_lds[h][3] != 0; }
    }
    if (threadIdx.x < 2 && (threadIdx.x == 0 || ib != ia)) {
        const int h = threadIdx.x;
        const DevJob &job = jobs[h ? ib : ia];
        DevResult res;
        res.flags = set[h] ? 0 : FLAG_NO_END; res.n_ops = 0; res.n_vsa = 0; res.pad = 0; res.qs = res.ts = 0;
        res.cell_size = DP::CS; res.ops_off = 0;
        for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
        res.final_cell[0] = sc[h]; res.final_cell[DP::CS - 1] = srp[h];
        res.score = sc[h]; res.end_set = set[h]; res.qe = job.Q; res.te = job.T; res.last_srp = srp[h];
        if (set[h]) DP::checkpoint_traceback(h ? ck_b : ck_a, job, vsas + job.vsa_off, res);
        results[h ? ib : ia] = res;
    }
    __syncthreads();
}

// persistent waves; workgroup p of the queue runs the p-th pair of the host's list (LaunchArgs::aux: two job indices, the second
// -1 where a job runs alone: its high half repeats it).  scratch.ckpt holds two job slabs per wave (ckpt_stride ints each).
// ROOTED: the jobs name their root (DevJob::root) and both jobs of a pair have the same one; else every inner state is computed.
template <class M, int R, int WPE, bool ROOTED, int NW = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void ckpt16_kernel(const KParams *kparams, DevSeqs seqs, const DevJob *jobs, const int *pairs, int n_pairs, DevResult *results,
                   DevVsa *vsas, DevScratch scratch, int *queue) {
    using RT = Roots<M>;
    static_assert(!ROOTED || RT::disjoint(), "a rooted pass needs components to choose from");
    // the launch constants stay in memory (a strip reads them once, for its profile); the LDS goes to the stages (column inputs and
    // carry row: 5 KB per wave for one strand's states) and the query profiles: 52.3 KB per workgroup of four waves, three per CU
    __shared__ __attribute__((aligned(512))) int stage_mem[NW * Stage16<WaveCK16<M, R, ROOTED ? Roots<M>::root(0) : -1>::BND>::INTS];
    static_assert(!ROOTED || WaveCK16<M, R, Roots<M>::root(0)>::BND == WaveCK16<M, R, Roots<M>::root(Roots<M>::count() - 1)>::BND, "one stage layout for every root");
    __shared__ int next_job;
    __shared__ int prog[NW];
    __shared__ int corner_lds[2][4];
    __shared__ __attribute__((aligned(16))) int prof_mem[NW * 2 * Prof16<R>::INTS];
    __shared__ uint8_t tdense_lds[32];
    if (threadIdx.x < 32) tdense_lds[threadIdx.x] = reinterpret_cast<const uint8_t *>(seqs.sub_colptr)[threadIdx.x];
    __syncthreads();
    const KParams *kp_lds = kparams;
    int *bnd = scratch.bnd + (long long)blockIdx.x * scratch.bnd_stride;
    int *ck_a = scratch.ckpt + (long long)blockIdx.x * 2 * scratch.ckpt_stride, *ck_b = ck_a + scratch.ckpt_stride;
    for (;;) {
        if (threadIdx.x == 0) next_job = atomicAdd(queue, 1);
        __syncthreads();
        const int pid = __builtin_amdgcn_readfirstlane(next_job);      // wave-uniform: job descriptions and pointers in scalar registers
        __syncthreads();
        if (pid >= n_pairs) break;
        const int ia = pairs[2 * pid], ib = pairs[2 * pid + 1] >= 0 ? pairs[2 * pid + 1] : ia;
        if constexpr (ROOTED) {
            const int root = jobs[ia].root;
            bool ran = false;
            static_for<RT::count()>([&](auto X_) __attribute__((always_inline)) { constexpr int X = X_;
                constexpr int ROOT = RT::root(X);
                if (!ran && root == ROOT) {
                    ckpt16_pair<M, R, ROOT, NW>(kp_lds, seqs, jobs, ia, ib, results, vsas, bnd, ck_a, ck_b, prog, corner_lds, prof_mem, tdense_lds, stage_mem);
                    ran = true;
                }
            });
            if (!ran && threadIdx.x < 2 && (threadIdx.x == 0 || ib != ia)) {      // a root the model does not have
                DevResult res;
                res.flags = FLAG_NO_END; res.n_ops = 0; res.n_vsa = 0; res.pad = 0; res.qs = res.ts = res.qe = res.te = 0;
                res.cell_size = 0; res.ops_off = 0; res.score = LOW; res.end_set = 0; res.last_srp = 0;
                for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
                results[threadIdx.x ? ib : ia] = res;
            }
        } else {
            ckpt16_pair<M, R, -1, NW>(kp_lds, seqs, jobs, ia, ib, results, vsas, bnd, ck_a, ck_b, prog, corner_lds, prof_mem, tdense_lds, stage_mem);
        }
    }
}

}  // namespace c4k
