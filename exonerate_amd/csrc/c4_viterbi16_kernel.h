// c4_viterbi16_kernel.h — the whole-rectangle FIND_SCORE pass of the windowed region scheme (c4_viterbi_kernel.h, SEED = 1)
// with TWO jobs per lane in packed 16-bit halves.
//
// The score pass is pure max-plus when nothing but the score is carried (first valid transition assigns, later ones
// replace on strict <, viterbi.c:766-775 = a maximum), and the local-scope shortcuts of the 32-bit kernel already make the
// validity of every transition a compile-time fact.  So a register can hold the same cell of two independent jobs —
// job A in the low half, job B in the high half — and v_pk_add_i16 (saturating) / v_pk_max_i16 evaluate both at once:
// the DPP row exchange, the LDS rings between the cooperating waves and every add / max serve two rectangles.  What does
// not pack — the substitution-score reads (two LDS reads and a v_perm per row), the per-column splice scores (clamp
// and v_perm per array), the end-cell bookkeeping (behind the same rare wave-uniform branch) — is per column or rare.
//
// Exactness (the host only picks this kernel when all of it holds, Engine::pk16_ok):
//   * every score a result can depend on fits 16 bits: (Q + 1) x the largest substitution score <= 16 000, calc constants
//     below 16 000 in magnitude;
//   * "unset" is -32 768 and adds saturate: a value that is unset or saturated in the reference's sense (-987 654 321 + x)
//     would need +32 768 to reach the 0 every match state has from START in the same cell — more than a whole query can
//     contribute — so it never wins a maximum that a result reads, exactly as the 32-bit kernel's phantom candidates
//     (c4_viterbi_kernel.h, eval_cell: the row-0 note);
//   * the intron-start shadow (a target position: 17+ bits) becomes the intron's length so far, a saturating 15-bit
//     counter: the post-splice calc only asks whether the length lies in [min, max] (intron.c:150-160); lengths beyond
//     32 767 saturate, which is still "long enough", and the upper limit cannot fail when T + 4 <= max_intron (checked);
//   * the column dumps the region windows start from are written in the 32-bit kernel's own format (scores sign-extended,
//     shadow = column - length - 2): a window computes from them what it computes from the 32-bit pass's dumps wherever a
//     result can see it, and the host checks every window's corner score against this pass's score as before.
// The two jobs of a lane need not have the same size: cells outside a job's own rectangle hold garbage that only flows
// away from the rectangle (every transition advances), its loads are clamped, its end cells are ignored.
#pragma once
#include "c4_viterbi_kernel.h"

namespace c4k {

// inline asm keeps each of these ONE instruction: the builtin forms (__builtin_elementwise_add_sat on short2 ...) are folded
// into per-half compares and selects around the mask logic and cost 8 % of the pass; operands in VGPRs: with the launch
// constants as scalar operands the kernel spills 62 SGPRs and is 2 % slower (measured, profiles/r03_pk16.md)
typedef short pk_s2 __attribute__((ext_vector_type(2)));
// VAR 0: every packed instruction is its own asm statement.  VAR 1: add / max through clang's vector builtins and the
// "a < b per half" mask as ONE asm statement: the compiler's hazard pass puts a wait state (s_nop) between two dependent asm
// statements that follow each other (it cannot see that they write whole registers), 58 of them per step in the VAR 0 form
template <int VAR> __device__ __forceinline__ int pk_add(int a, int b) {
    if constexpr (VAR >= 1) return __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
    else { int r; asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
}
template <int VAR> __device__ __forceinline__ int pk_max(int a, int b) {
    if constexpr (VAR >= 1) return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(pk_s2, a), __builtin_bit_cast(pk_s2, b)));
    else { int r; asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
}
__device__ __forceinline__ int pk_sub(int a, int b) { int r; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
// per half: 0xffff where the half is negative, else 0
__device__ __forceinline__ int pk_neg_mask(int d, int fifteen) { int r; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(fifteen), "v"(d)); return r; }
// per half: 0xffff where a < b (the saturating difference is negative), else 0
template <int VAR> __device__ __forceinline__ int pk_lt_mask(int a, int b, int fifteen) {
    if constexpr (VAR >= 1) {
        int r;
        asm("v_pk_sub_i16 %0, %1, %2 clamp\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(b));
        return r;
    } else {
        return pk_neg_mask(pk_sub(a, b), fifteen);
    }
}
__device__ __forceinline__ int pk_pack(int lo, int hi) { return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }
__device__ __forceinline__ int pk_half(int x, int h) { return h ? (x >> 16) : ((x << 16) >> 16); }
__device__ __forceinline__ int clamp16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }

constexpr int NEG16 = (int)0x80008000u;          // -32 768 in both halves

// (Roots<M>, the components of a model as seen from END: c4_viterbi_kernel.h)

// The column dumps in 16-bit form (DUMP16: what the packed region windows of c4_win16_kernel.h start from): per dumped row
// the scores of the inner states (START is never set and nothing reads END), then the intron lengths something can still
// read, one 16-bit half each in that order, two halves per int.  est2genome: 8 + 2 halves = 5 ints per row (the 32-bit
// format: 12).  The length is stored as the packed passes carry it (saturating counter), not as a target position.
template <class M>
struct Dump16 {
    using W1 = WaveDP<M, 1, MODE_SCORE, false, true, false, false, 0, 1>;
    static constexpr bool live(int s) { return M::NDES > 0 && W1::slot_live(s, 0); }
    static constexpr bool inner(int s) { return s != M::START && s != M::END; }
    static constexpr int n_inner() { int n = 0; for (int s = 0; s < M::NS; s++) n += inner(s); return n; }
    static constexpr int n_live() { int n = 0; for (int s = 0; s < M::NS; s++) n += (inner(s) && live(s)); return n; }
    static constexpr int NH = n_inner() + n_live();                 // halves per row
    static constexpr int SEEDW16 = (NH + 1) / 2;                    // ints per row
    static constexpr int half_of_sc(int s) { int n = 0; for (int x = 0; x < s; x++) n += inner(x); return n; }
    static constexpr int half_of_il(int s) { int n = n_inner(); for (int x = 0; x < s; x++) n += (inner(x) && live(x)); return n; }
    // the state whose score (h < n_inner()) or length sits in half h; -1: padding
    static constexpr int state_of_half(int h) {
        if (h < n_inner()) { for (int s = 0; s < M::NS; s++) if (inner(s) && half_of_sc(s) == h) return s; return -1; }
        for (int s = 0; s < M::NS; s++) if (inner(s) && live(s) && half_of_il(s) == h) return s;
        return -1;
    }
};

// IO 1 (the launch's queries fit ONE super-strip of the cooperating waves and its targets hold at most NCODE different
// residue codes): nothing of the column loop goes through vector memory or a wave-uniform branch.
//   * every strip boundary is an LDS ring (the first wave reads a constant empty column, the last one writes a column
//     nobody reads): no HBM carry row, no "ring or memory" branch around the carry reads and writes;
//   * the column inputs -- the four splice values of both jobs, already interleaved into packed halves, and the residue
//     codes -- come from a per-wave LDS stage of 128 columns that the wave refills every chunk (64 coalesced columns,
//     clamped once there): a step reads its column with two LDS reads instead of four global loads, their clamps, address
//     arithmetic and the four v_perm;
//   * the substitution scores of a lane's R query rows against each residue code sit in LDS as one 8-byte entry per
//     (job, code, lane) -- a query profile, rebuilt per strip: a step reads two entries (conflict-free: the bank depends on
//     the lane only) instead of 2 R table words at computed addresses.
// 301 -> ~225 instructions per step of 8 cells (profiles/r04_score_budget.md).
template <class M, int R, int VAR = 0, bool DUMP16 = false, int IO = 0, int NCODE_ = 6, bool MEMC = false>
struct WaveDP16 {
    using F = Facts<M>;
    using W32 = WaveDP<M, R, MODE_SCORE, false, true, false, false, 0, 1>;     // the 32-bit score pass: dump layout
    using D16 = Dump16<M>;
    static constexpr int NS = M::NS, NCOL = M::MAXAT + 1, W = 64 * R, DC = M::MAXAT, SEEDW = DUMP16 ? D16::SEEDW16 : W32::SEEDW;
    static constexpr bool live(int s) { return M::NDES > 0 && W32::slot_live(s, 0); }
    static constexpr int NEXP = F::n_exported();
    static constexpr int n_exported_live() { int n = 0; for (int x = 0; x < NS; x++) n += (F::exported(x) && live(x)); return n; }
    // ints per column between strips: score pair + length pair per exported state (IO 1: a length pair only where one is live)
    static constexpr int BND = IO ? (NEXP + n_exported_live() + 3) / 4 * 4 : NEXP * 2;
    static constexpr int RING = 256;
    // steps between two meetings of the cooperating waves; IO 1: also between two refills of the column stage, whose 128
    // columns hold the 63 + 64 columns the lanes of a wave read during a chunk
    static constexpr int CH = IO ? 63 / NCOL * NCOL : (64 + NCOL - 1) / NCOL * NCOL;
    static constexpr int NCODE = NCODE_;                 // IO 1: residue codes a launch's targets may hold (6; 8 in the form for IUPAC-coded targets)
    // column stage: 128 columns x 8 planes of one int (six used), plane-major -- the lanes of a wave read consecutive columns, so a
    // plane read is 64 consecutive words: no bank conflict (column-major entries of 32 bytes put 64 lanes on 8 banks: 72 % of the
    // LDS-active cycles of the first LDS-fed form were conflicts, profiles/r04_c_sq.csv)
    static constexpr int STAGE_COLS = 128, STAGE_INTS = 8;
    // query profile: per (wave, job) [code][lane] entries of NP ints (two rows' scores per int); 8 bytes per entry for 2 and 4 rows
    // per lane (a b64 read; half of it unused with 2 rows), 12 for 6 rows (three ints: 64 lanes x 12 contiguous bytes per code, no
    // bank conflict either)
    static constexpr int NP = (R + 1) / 2, PROF_EB = (R == 6) ? 12 : 8, PROF_CODE = 64 * PROF_EB;
    static constexpr int PROF_INTS = NCODE * PROF_CODE / 4;
    static_assert(!IO || (VAR >= 1 && F::has_splice() && (R == 6 || R == 4 || R == 2)), "the staged form is built for the packed splice entries and 2, 4 or 6 rows per lane");
    static_assert(!F::has_phase(), "split-codon calcs are not packed");
    static_assert(M::NDES <= 1, "one shadow designation");
    typedef __attribute__((address_space(3))) int lds_int;
    struct C16 { int sc[NS]; int il[NS]; };
    // an LDS byte address kept as a number (it is advanced and masked like one) back to a pointer
    __device__ __forceinline__ static lds_int *lds_at(int a) { return (lds_int *)(size_t)(unsigned)a; }
    __device__ __forceinline__ static int lds_addr(const lds_int *p) { return (int)(unsigned)(size_t)p; }

    const KParams *kp;
    int lane;
    // per job (0 = low half, 1 = high half)
    const uint8_t *qc[2], *tc[2];
    const int *ss[2];
    const uint2 *ss16[2];             // VAR 1: the four splice values of a column in one 8-byte load (ss16_kernel)
    long long ss_stride;
    int Q[2], T[2], q0[2], t0[2], tlast[2], seed_rows[2], seed_kshift;
    int *seed_wr[2];
    int Qm, Tm;                                         // the larger of the two
    // the intron length counter is kept as (length so far) - (min_intron - 4), saturating: an intron opens at open_il_pk =
    // -(min_intron - 4) and "long enough for the 3' site" is "not negative"; the dumps hold the length itself (lim_pk is added)
    int open_il_pk, lim_pk, at_pk[4], cv_pk[16], fifteen;
    C16 col[NCOL][R], nbr[NCOL], expo, nx_carry;
    int qrow[2][R];
    int nx_tcode[2], nx_sp[2][4];
    uint2 nx_sp16[2];
    lds_int *ring_in, *ring_out;
    bool use_ring_in, use_ring_out, carry_ok, carry_cols;
    // IO 1 with MEMC (queries of more rows than one workgroup's strips hold): between two super-strips the bottom row of the last
    // wave goes through the workgroup's slab in memory as in the IO 0 form -- written by the last wave (mem_out), read by the first
    // wave of the next super-strip (mem_in); every other boundary stays a ring in LDS.  Wave-uniform.
    bool mem_in, mem_out;
    // IO 1
    int ring_in_mask, ring_out_mask;                    // 255, or 0 for the constant column of the first / last wave
    int nx_sp4[4], nx_off[2];                           // next column: packed splice values; profile byte offsets of its two codes
    int nx_prof[6];                                     // ... and the profile entries of those codes: job A rows 0-1, 2-3, 4-5 in [0..2], job B in [3..5]
    int prof_a[2];                                      // LDS byte address of this lane's profile entry of code 0, per job
    int stage_a, stage_base;                            // LDS byte address of the next column's stage entry; of the wave's stage (4 KB-aligned)
    const uint8_t *tdense;                              // [24] code -> dense index (0xff: not in the launch), [24 + d] code of index d
    int best[2], best_i[2], best_j[2], best_pk;
    int unset_pk;                       // what best_pk holds for a job without an end cell yet: -32 768; IO 1, a lane none of whose rows
                                        // belongs to the job: 32 767 (nothing of it can be an end cell, nothing beats it)
    // best_i: (query row << 4) | the state the cell's END was entered from (Roots): the row-major order of the rows is
    // that of these numbers, and no register is spent on the state
    static_assert(M::NS <= 16, "state in four bits");
    bool best_set[2];
    int sbest[2], sbest_i[2], sbest_j[2];
    bool sbest_set[2];

    template <class Fn>
    __device__ __forceinline__ static void for_exported(Fn &&fn) {
        int slot = 0;
        static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (F::exported(S)) { fn(S_, slot); slot += IO ? (live(S) ? 2 : 1) : 2; }
        });
    }
    template <class P>
    __device__ __forceinline__ static void write_empty_column(P colp) {
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            colp[slot] = NEG16;
            if constexpr (!IO || live(S)) colp[slot + 1] = 0;
        });
    }
    __device__ __forceinline__ void prefetch_carry(int s_next, const int *bnd_in) {
        const int jx = s_next < 0 ? 0 : (s_next > Tm ? Tm : s_next);
        if constexpr (IO == 1) {
            if constexpr (MEMC) {
                if (mem_in) {
                    const int *p = bnd_in + (long long)jx * BND;
                    for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                        nx_carry.sc[S] = p[slot];
                        if constexpr (live(S)) nx_carry.il[S] = p[slot + 1];
                    });
                    return;
                }
            }
            const lds_int *p = ring_in + (jx & ring_in_mask) * BND;
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                nx_carry.sc[S] = p[slot];
                if constexpr (live(S)) nx_carry.il[S] = p[slot + 1];
            });
            return;
        }
        const int jc = (carry_cols | use_ring_in) ? jx : 0;
        if (use_ring_in) {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                const lds_int *p = ring_in + (jc & (RING - 1)) * BND + slot;
                nx_carry.sc[S] = p[0];
                if constexpr (live(S)) nx_carry.il[S] = p[1];
            });
        } else {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                const int *p = bnd_in + (long long)jc * BND + slot;
                nx_carry.sc[S] = p[0];
                if constexpr (live(S)) nx_carry.il[S] = p[1];
            });
        }
    }
    // IO 1: the next column's entry of the stage (the lane's columns follow each other: a running address)
    __device__ __forceinline__ void prefetch_staged() {
        const lds_int *p = lds_at(stage_a);
        static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_sp4[K] = p[K * STAGE_COLS]; });
        nx_off[0] = p[4 * STAGE_COLS]; nx_off[1] = p[5 * STAGE_COLS];
        stage_a = ((stage_a + 4) & (STAGE_COLS * 4 - 1)) | stage_base;
    }
    // IO 1: the profile entries of the next column's codes; issued in the middle of a step, when the stage entry has arrived
    __device__ __forceinline__ void prefetch_profile() {
        const lds_int *pa = lds_at(prof_a[0] + nx_off[0]), *pb = lds_at(prof_a[1] + nx_off[1]);
        static_for<NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_prof[K] = pa[K]; nx_prof[3 + K] = pb[K]; });
    }
    // IO 1: columns c0 + lane of both jobs into the stage -- clamped as prefetch_column clamps them, the splice values of the two
    // jobs interleaved into packed halves (what step() did with four v_perm per step), the residue codes as profile offsets
    __device__ __forceinline__ void fill_stage(lds_int *stage, int c0) {
        constexpr int mat = F::match_at();
        const int c = c0 + lane;
        uint2 sv[2]; int off[2];
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            int ti = t0[H] + c - mat;
            ti = ti < 0 ? 0 : (ti > tlast[H] ? tlast[H] : ti);
            int tp = t0[H] + c - 2;
            tp = tp < 0 ? 0 : (tp > tlast[H] ? tlast[H] : tp);
            sv[H] = ss16[H][(unsigned)tp];
            off[H] = (int)tdense[tc[H][(unsigned)ti]] * PROF_CODE;
        });
        lds_int *p = stage + (c & (STAGE_COLS - 1));
        p[0 * STAGE_COLS] = (int)__builtin_amdgcn_perm(sv[1].x, sv[0].x, 0x05040100u);
        p[1 * STAGE_COLS] = (int)__builtin_amdgcn_perm(sv[1].x, sv[0].x, 0x07060302u);
        p[2 * STAGE_COLS] = (int)__builtin_amdgcn_perm(sv[1].y, sv[0].y, 0x05040100u);
        p[3 * STAGE_COLS] = (int)__builtin_amdgcn_perm(sv[1].y, sv[0].y, 0x07060302u);
        p[4 * STAGE_COLS] = off[0]; p[5 * STAGE_COLS] = off[1];
    }
    // IO 1: the substitution scores of this lane's R rows against every residue code of the launch: one 8-byte entry per
    // (job, code, lane), four 16-bit scores.  Rows below a job's last one (the rest of the last strip; nothing above reads
    // them, nothing of them is dumped or reported) score DEAD_ROW against everything: their match state then never leaves the 0
    // START gives it, and their END never beats a best score -- without that the lanes that hold such rows would send their
    // wave through the end-cell bookkeeping in every step (counters of round 4: 131 instructions, a quarter of the waves).
    static constexpr int DEAD_ROW = -16000;
    __device__ __forceinline__ void build_profile(int i0) {
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            int qr[R];
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                const int i = i0 + RR;
                qr[RR] = 24 * ((i >= 1 && i <= Q[H]) ? (int)qc[H][q0[H] + i - 1] : 0);
            });
            for (int d = 0; d < NCODE; d++) {
                const int code = tdense[24 + d];
                int v[R];
                static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                    v[RR] = (i0 + RR > Q[H]) ? DEAD_ROW : clamp16(kp->submat[qr[RR] + code]);
                });
                lds_int *p = lds_at(prof_a[H] + d * PROF_CODE);
                static_for<NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; p[K] = pk_pack(v[2 * K], v[2 * K + 1 < R ? 2 * K + 1 : 2 * K]); });
            }
        });
    }

    __device__ __forceinline__ void prefetch_column(int j) {
        if constexpr (IO == 1) { prefetch_staged(); return; }
        constexpr int mat = F::match_at();
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            int ti = t0[H] + j - mat;
            ti = ti < 0 ? 0 : (ti > tlast[H] ? tlast[H] : ti);
            nx_tcode[H] = tc[H][(unsigned)ti];
            if constexpr (F::has_splice()) {
                int tp = t0[H] + j - 2;
                tp = tp < 0 ? 0 : (tp > tlast[H] ? tlast[H] : tp);
                if constexpr (VAR >= 1) {
                    nx_sp16[H] = ss16[H][(unsigned)tp];
                } else {
                    static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                        nx_sp[H][K] = ss[H][(long long)K * ss_stride + tp];
                    });
                }
            }
        });
    }

    template <int RR, int PH, bool JINT>
    __device__ __forceinline__ void eval_cell(int j, int ms, const int (&sp)[4]) {
        C16 &c = col[PH][RR];
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
            constexpr TrDesc t = M::tr[K];
            constexpr int PD = (PH - t.at + NCOL) % NCOL;
            const C16 &src = (t.aq == 0) ? col[PD][RR] : (RR > 0 ? col[PD][RR > 0 ? RR - 1 : 0] : nbr[PD]);
            int cand;
            if constexpr (t.in == M::START) cand = 0;
            else cand = src.sc[t.in];
            if constexpr (t.calc >= 0) {
                constexpr CalcDesc cd = M::calc[t.calc];
                if constexpr (cd.kind == CALC_CONST) cand = pk_add<VAR>(cand, cv_pk[t.calc]);
                else if constexpr (cd.kind >= CALC_MATCH_DNA && cd.kind <= CALC_MATCH_P2D) cand = pk_add<VAR>(cand, ms);
                else if constexpr (cd.kind == CALC_SPLICE_PRE) cand = pk_add<VAR>(cand, sp[cd.param]);
                else if constexpr (cd.kind == CALC_SPLICE_POST) {
                    // intron length = length so far + this advance + 2 (c4_viterbi_kernel.h: (t0 + j - at) - shadow + 2); too
                    // short: the transition scores -987654321 (intron.c:150-160); too long cannot happen (T + 4 <= max_intron)
                    static_assert(live(t.in), "post-splice calc without a length");
                    // length so far < min - at - 2: the counter starts at -(min - at - 2) (open_il_pk), so that is its sign -- one
                    // instruction where a comparison with the limit is two
                    const int bad = pk_neg_mask(src.il[t.in], fifteen);
                    const int sv = (bad & NEG16) | (~bad & sp[cd.param]);
                    cand = pk_add<VAR>(cand, sv);
                }
            }
            if constexpr (!JINT && t.at > 0) cand = (j >= t.at) ? cand : NEG16;
            int ilc = 0;
            if constexpr (live(t.out)) {
                if constexpr (F::owns_shadow(t.in, 0)) ilc = open_il_pk;
                else if constexpr (live(t.in)) ilc = pk_add<VAR>(src.il[t.in], at_pk[t.at]);
            }
            if constexpr (F::code(K) == 1) {                     // the first transition into this state
                c.sc[t.out] = cand;
                if constexpr (live(t.out)) c.il[t.out] = ilc;
            } else {
                if constexpr (live(t.out)) {
                    const int win = pk_lt_mask<VAR>(c.sc[t.out], cand, fifteen);          // strict <: the newcomer wins
                    c.il[t.out] = (win & ilc) | (~win & c.il[t.out]);
                }
                c.sc[t.out] = pk_max<VAR>(c.sc[t.out], cand);
            }
        });
    }

    template <bool JINT, int PH>
    __device__ __forceinline__ void step(int s, int i0, bool last_strip, const int *bnd_in, int *bnd_out) {
        const int j = s - lane;
        int ms[R];
        if constexpr (IO == 1) {
            static_for<NP>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                ms[2 * K] = (int)__builtin_amdgcn_perm((unsigned)nx_prof[3 + K], (unsigned)nx_prof[K], 0x05040100u);
                if constexpr (2 * K + 1 < R) ms[2 * K + 1] = (int)__builtin_amdgcn_perm((unsigned)nx_prof[3 + K], (unsigned)nx_prof[K], 0x07060302u);
            });
        } else {
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                ms[RR] = pk_pack(kp->submat[qrow[0][RR] + nx_tcode[0]], kp->submat[qrow[1][RR] + nx_tcode[1]]);
            });
        }
        int sp[4] = {0, 0, 0, 0};
        if constexpr (IO == 1) {
            static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; sp[K] = nx_sp4[K]; });
        } else if constexpr (F::has_splice() && VAR >= 1) {
            // the values arrive clamped, with the calc constant of a pre-splice transition folded in (ss16_kernel): job A's
            // four in the halves of nx_sp16[0], job B's in nx_sp16[1]; one v_perm each puts a value of both into one register
            sp[0] = (int)__builtin_amdgcn_perm(nx_sp16[1].x, nx_sp16[0].x, 0x05040100u);
            sp[1] = (int)__builtin_amdgcn_perm(nx_sp16[1].x, nx_sp16[0].x, 0x07060302u);
            sp[2] = (int)__builtin_amdgcn_perm(nx_sp16[1].y, nx_sp16[0].y, 0x05040100u);
            sp[3] = (int)__builtin_amdgcn_perm(nx_sp16[1].y, nx_sp16[0].y, 0x07060302u);
        } else if constexpr (F::has_splice()) {
            static_for<M::NC>([&](auto CI_) __attribute__((always_inline)) { constexpr int CI = CI_;
                constexpr CalcDesc cd = M::calc[CI];
                if constexpr (cd.kind == CALC_SPLICE_PRE)
                    sp[cd.param] = pk_pack(clamp16(kp->calc_value[CI] + nx_sp[0][cd.param]), clamp16(kp->calc_value[CI] + nx_sp[1][cd.param]));
                if constexpr (cd.kind == CALC_SPLICE_POST)
                    sp[cd.param] = pk_pack(clamp16(nx_sp[0][cd.param]), clamp16(nx_sp[1][cd.param]));
            });
        }
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            nbr[PH].sc[S] = dpp_shr1(nx_carry.sc[S], expo.sc[S]);
            if constexpr (live(S)) nbr[PH].il[S] = dpp_shr1(nx_carry.il[S], expo.il[S]);
        });
        prefetch_carry(s + 1, bnd_in);
        prefetch_column(j + 1);
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            eval_cell<RR, PH, JINT>(j, ms[RR], sp);
            // IO 1: half-way through the step the stage entry read above has arrived; the profile entries it points to are then
            // there when the next step starts
            if constexpr (IO == 1 && RR == R / 2 - 1) prefetch_profile();
        });
        // end cell (viterbi.c:778-791): a new maximum is rare; one packed maximum over the lane's cells decides
        {
            int m = col[PH][0].sc[M::END];
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_; if constexpr (RR > 0) m = pk_max<VAR>(m, col[PH][RR].sc[M::END]); });
            const bool cand = (pk_sub(best_pk, m) & NEG16) != 0;
            if (__builtin_amdgcn_ballot_w64(cand)) {
                static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                    const bool jact = (j >= 0) & (j <= T[H]);
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int tsc = pk_half(col[PH][RR].sc[M::END], H);
                        const bool upd = jact & (i0 + RR <= Q[H]) & (!best_set[H] | (best[H] < tsc));
                        // which transition into END holds it: the first assigns, later ones replace on strict < (viterbi.c:766-775)
                        int from = -1, fsc = 0;
                        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                            constexpr TrDesc t = M::tr[K];
                            if constexpr (t.out == M::END && t.in != M::START) {
                                const int v = pk_half(col[PH][RR].sc[t.in], H);
                                const bool win = (from < 0) | (fsc < v);
                                fsc = win ? v : fsc;
                                from = win ? t.in : from;
                            }
                        });
                        best[H] = upd ? tsc : best[H];
                        best_i[H] = upd ? (((i0 + RR) << 4) | from) : best_i[H];
                        best_j[H] = upd ? j : best_j[H];
                        best_set[H] = best_set[H] | upd;
                    });
                });
                best_pk = pk_pack(best_set[0] ? best[0] : pk_half(unset_pk, 0), best_set[1] ? best[1] : pk_half(unset_pk, 1));
            }
        }
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            expo.sc[S] = col[PH][R - 1].sc[S];
            if constexpr (live(S)) expo.il[S] = col[PH][R - 1].il[S];
        });
        if constexpr (IO == 1) {
            // the last wave's ring is one column nobody reads; inside the rectangle's columns (JINT) lane 63 is always on a column
            if (lane == 63 && (JINT || (j >= 0 && j <= Tm))) {
                bool to_memory = false;
                if constexpr (MEMC) to_memory = mem_out;
                if (to_memory) {
                    int *p = bnd_out + (long long)j * BND;
                    for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                        p[slot] = expo.sc[S];
                        if constexpr (live(S)) p[slot + 1] = expo.il[S];
                    });
                } else {
                    lds_int *p = ring_out + (j & ring_out_mask) * BND;
                    for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                        p[slot] = expo.sc[S];
                        if constexpr (live(S)) p[slot + 1] = expo.il[S];
                    });
                }
            }
        } else if (!last_strip && lane == 63 && j >= 0 && j <= Tm) {
            if (use_ring_out) {
                for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                    lds_int *p = ring_out + (j & (RING - 1)) * BND + slot;
                    p[0] = expo.sc[S];
                    if constexpr (live(S)) p[1] = expo.il[S];
                });
            } else {
                for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                    int *p = bnd_out + (long long)j * BND + slot;
                    p[0] = expo.sc[S];
                    if constexpr (live(S)) p[1] = expo.il[S];
                });
            }
        }
        // the DC columns that end in d*K go to each job's dumps, in the 32-bit pass's format
        if (((unsigned)(s + DC - 1) & (unsigned)((1 << seed_kshift) - 1)) <= (unsigned)(DC + 62)) {
            const int d = (j + DC - 1) >> seed_kshift;
            const unsigned which = (unsigned)(j - ((d << seed_kshift) - (DC - 1)));
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                if ((j >= 0) & (j <= T[H]) & (d >= 1) & (which < (unsigned)DC) & ((d << seed_kshift) <= T[H])) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0 + RR;
                        if (i <= Q[H]) {
                            int *p = seed_wr[H] + (((long long)(d - 1) * DC + which) * seed_rows[H] + i) * SEEDW;
                            if constexpr (DUMP16) {
                                // this job's halves of two values per word (Dump16)
                                auto half_reg = [&](auto HI_) __attribute__((always_inline)) -> int { constexpr int HI = HI_;
                                    constexpr int S = D16::state_of_half(HI);
                                    if constexpr (S < 0) return 0;
                                    else if constexpr (HI < D16::n_inner()) return col[PH][RR].sc[S];
                                    else return pk_add<VAR>(col[PH][RR].il[S], lim_pk);
                                };
                                static_for<D16::SEEDW16>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
                                    const int va = half_reg(IC<2 * K>{}), vb = half_reg(IC<2 * K + 1>{});
                                    store_dword<K * 4>(p, (int)__builtin_amdgcn_perm((unsigned)vb, (unsigned)va, H ? 0x07060302u : 0x05040100u));
                                });
                            } else {
                                static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                    store_dword<S * 4>(p, pk_half(col[PH][RR].sc[S], H));
                                    if constexpr (live(S))
                                        store_dword<W32::dump_pos(S, 0) * 4>(p, t0[H] + j - pk_half(pk_add<VAR>(col[PH][RR].il[S], lim_pk), H) - 2);
                                });
                            }
                        }
                    });
                }
            });
        }
    }

    __device__ __forceinline__ void strip_begin() {
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            sbest[H] = best[H]; sbest_i[H] = best_i[H]; sbest_j[H] = best_j[H]; sbest_set[H] = best_set[H];
            best_set[H] = false; best[H] = LOW;
        });
        best_pk = unset_pk;
    }
    __device__ __forceinline__ void strip_end() {
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            const bool keep_old = sbest_set[H] & (!best_set[H] | (sbest[H] > best[H]) |
                                                  ((sbest[H] == best[H]) & ((sbest_j[H] < best_j[H]) | ((sbest_j[H] == best_j[H]) & (sbest_i[H] < best_i[H])))));
            best[H] = keep_old ? sbest[H] : best[H]; best_i[H] = keep_old ? sbest_i[H] : best_i[H]; best_j[H] = keep_old ? sbest_j[H] : best_j[H];
            best_set[H] = best_set[H] | sbest_set[H];
        });
    }
    __device__ __forceinline__ void reduce_best() {
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            for (int off = 32; off > 0; off >>= 1) {
                const int o_best = __shfl_xor(best[H], off), o_i = __shfl_xor(best_i[H], off), o_j = __shfl_xor(best_j[H], off);
                const bool o_set = __shfl_xor((int)best_set[H], off) != 0;
                const bool take = o_set & (!best_set[H] | (o_best > best[H]) |
                                           ((o_best == best[H]) & ((o_j < best_j[H]) | ((o_j == best_j[H]) & (o_i < best_i[H])))));
                best[H] = take ? o_best : best[H]; best_i[H] = take ? o_i : best_i[H]; best_j[H] = take ? o_j : best_j[H];
                best_set[H] = best_set[H] | o_set;
            }
        });
    }

    // VAR 2: the cooperating waves keep their distance through progress counters in LDS instead of meeting at a barrier
    // after every chunk: wave w starts chunk k once wave w-1 has finished chunk k+1 (the row above is there, as with the
    // barriers) and wave w+1 has finished chunk k-4 (the ring slots it is about to overwrite have been read: a ring holds
    // 256 columns, a chunk 66, lane 63 writes 63 columns behind the step) -- up to two chunks of slack per neighbour
    // instead of lock-step.
    // (IO 1: `stage` is this wave's column stage, `edge` the two constant columns -- the empty one the first wave reads and the
    // one the last wave writes; prof_a and tdense are set by the kernel)
    template <int NW>
    __device__ __forceinline__ void run_mw(const DevJob &ja, const DevJob &jb, const DevSeqs &seqs, int *bnd, lds_int *rings, int wid,
                                           lds_int *prog = nullptr, lds_int *stage = nullptr, lds_int *edge = nullptr) {
        // chunks a wave runs behind the wave above it: the row above must be there one column ahead of the step that reads it,
        // i.e. SKEW * CH - 1 - 63 >= CH
        constexpr int SKEW = (CH >= 64) ? 2 : 3;
        const DevJob *jp[2] = {&ja, &jb};
        static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
            const DevJob &jx = *jp[H];
            Q[H] = jx.Q; T[H] = jx.T; q0[H] = jx.q0; t0[H] = jx.t0;
            tlast[H] = seqs.tlen[jx.pair] > 0 ? seqs.tlen[jx.pair] - 1 : 0;
            qc[H] = seqs.qcode + seqs.qoff[jx.pair];
            tc[H] = seqs.tcode + seqs.toff[jx.pair];
            ss[H] = F::has_splice() ? seqs.ss + seqs.toff[jx.pair] : nullptr;
            ss16[H] = (F::has_splice() && VAR >= 1) ? seqs.ss16 + seqs.toff[jx.pair] : nullptr;
            seed_wr[H] = seqs.seed + jx.seed_off; seed_rows[H] = jx.seed_rows;
            best[H] = LOW; best_i[H] = best_j[H] = 0; best_set[H] = false;
        });
        ss_stride = seqs.ss_stride;
        seed_kshift = ja.seed_kshift;
        Qm = Q[0] > Q[1] ? Q[0] : Q[1]; Tm = T[0] > T[1] ? T[0] : T[1];
        best_pk = NEG16;
        fifteen = 0x000f000f;
        static_for<M::NC>([&](auto CI_) __attribute__((always_inline)) { constexpr int CI = CI_;
            const int v = clamp16(kp->calc_value[CI]);
            cv_pk[CI] = pk_pack(v, v);
        });
        static_for<4>([&](auto A_) __attribute__((always_inline)) { constexpr int A = A_; at_pk[A] = pk_pack(A, A); });
        {
            // "length so far < min_intron - at - 2" for the post-splice transitions (all advance the target by 2)
            const int lim = clamp16(kp->min_intron - 4);
            lim_pk = pk_pack(lim, lim);
            open_il_pk = pk_pack(-lim, -lim);
        }
        const int nstrips = (Qm + 1 + W - 1) / W;
        const int nsuper = (nstrips + NW - 1) / NW;
        const int nsteps = Tm + 64;
        const int nchunks = (nsteps + CH - 1) / CH;
        const int main_lo = 63 + M::MAXAT, main_hi = Tm;
        for (int sb = 0; sb < nsuper; sb++) {
            const int b = sb * NW + wid;
            const int i0 = b * W + lane * R;
            unset_pk = NEG16;
            if constexpr (IO == 1) {
                build_profile(i0);
                unset_pk = pk_pack(i0 > Q[0] ? 32767 : -32768, i0 > Q[1] ? 32767 : -32768);
            } else {
                static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0 + RR;
                        qrow[H][RR] = 24 * ((i >= 1 && i <= Q[H]) ? (int)qc[H][q0[H] + i - 1] : 0);
                    });
                });
            }
            static_for<NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                expo.sc[S] = NEG16; expo.il[S] = 0;
                static_for<NCOL>([&](auto D_) __attribute__((always_inline)) { constexpr int D = D_;
                    nbr[D].sc[S] = NEG16; nbr[D].il[S] = 0;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_; col[D][RR].sc[S] = NEG16; col[D][RR].il[S] = 0; });
                });
            });
            strip_begin();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            carry_cols = carry_ok & (sb > 0);
            const int *bnd_in = (sb == 0) ? bnd : bnd + BND + (carry_ok ? (long long)((sb + 1) & 1) * (Tm + 1) * BND : 0);
            int *bnd_out = bnd + BND + (carry_ok ? (long long)(sb & 1) * (Tm + 1) * BND : 0);
            use_ring_in = wid > 0; use_ring_out = wid < NW - 1;
            ring_in = rings + (wid > 0 ? wid - 1 : 0) * RING * BND;
            ring_out = rings + (wid < NW - 1 ? wid : 0) * RING * BND;
            if constexpr (IO == 1) {
                if (wid == 0) ring_in = edge;
                if (wid == NW - 1) ring_out = edge + BND;
                ring_in_mask = wid > 0 ? RING - 1 : 0;
                ring_out_mask = wid < NW - 1 ? RING - 1 : 0;
                mem_in = MEMC && wid == 0 && sb > 0;
                mem_out = MEMC && wid == NW - 1 && sb < nsuper - 1;
                stage_base = lds_addr(stage);
                stage_a = lds_addr(stage) + ((0 - lane) & (STAGE_COLS - 1)) * 4;
            }
            const bool last = (b >= nstrips - 1), idle = (b >= nstrips);
            auto group = [&](auto JI_, int s0) __attribute__((always_inline)) {
                constexpr bool JI = decltype(JI_)::value != 0;
                static_for<NCOL>([&](auto P_) __attribute__((always_inline)) { constexpr int P = P_;
                    step<JI, P>(s0 + P, i0, last, bnd_in, bnd_out);
                });
            };
            constexpr bool FLAGS = (VAR == 2);
            auto publish = [&](int done) __attribute__((always_inline)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(prog + wid, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            auto wait_for = [&](int w, int done) __attribute__((always_inline)) {
                while (__hip_atomic_load(prog + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < done) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            };
            auto before = [&](int k) __attribute__((always_inline)) {
                if constexpr (IO == 1) fill_stage(stage, k * CH + 1);           // columns k CH + 1 ... k CH + 64: what this chunk's steps read ahead
                if constexpr (FLAGS) {
                    // the row above: its chunk k + SKEW - 1 done; the ring slots this chunk overwrites (columns up to (k + 1) CH - 64 - RING):
                    // read by the wave below, i.e. its chunk floor(((k + 1) CH - 65 - RING) / CH) = k + 1 - BACK done
                    constexpr int BACK = (RING + 65 + CH - 1) / CH;
                    if (wid > 0) wait_for(wid - 1, k + SKEW < nchunks ? k + SKEW : nchunks);
                    if (wid < NW - 1 && k + 1 >= BACK) wait_for(wid + 1, k + 2 - BACK);
                }
            };
            auto after = [&](int k) __attribute__((always_inline)) {
                if constexpr (FLAGS) publish(k + 1);
                else __syncthreads();
            };
            if constexpr (FLAGS) {
                if (lane == 0) __hip_atomic_store(prog + wid, idle ? 0x40000000 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __syncthreads();
            } else {
                for (int t = 0; t < SKEW * wid; t++) __syncthreads();
            }
            if (idle) {
                if constexpr (!FLAGS) for (int k = 0; k < nchunks; k++) __syncthreads();
            } else {
                if constexpr (IO == 1) {
                    fill_stage(stage, -63);          // columns -63 ... 0: what the first steps of the lanes read
                    if constexpr (FLAGS) { if (wid > 0) wait_for(wid - 1, SKEW < nchunks ? SKEW : nchunks); }
                } else before(0);                    // the first carry column is read ahead of the first step
                prefetch_column(0 - lane);
                if constexpr (IO == 1) prefetch_profile();
                prefetch_carry(0, bnd_in);
                int k = 0;
                for (; k < nchunks && k * CH < main_lo; k++) {
                    before(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<0>{}, s);
                    after(k);
                }
                for (; k < nchunks && k * CH + CH - 1 <= main_hi; k++) {
                    before(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<1>{}, s);
                    after(k);
                }
                for (; k < nchunks; k++) {
                    before(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<0>{}, s);
                    after(k);
                }
            }
            if constexpr (!FLAGS) for (int t = 0; t < SKEW * (NW - 1 - wid); t++) __syncthreads();
            strip_end();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
        }
    }
};

// The four splice-site values of every target position as the packed pass adds them: clamped to 16 bits, the calc constant
// of a pre-splice transition folded in (exactly the value step() builds per column in the VAR 0 form: the same loop over the
// calcs, the same clamp), interleaved so that a column is one 8-byte load: x = value 0 | value 1 << 16, y = value 2 | value 3 << 16.
template <class M>
__global__ void ss16_kernel(const KParams *kp, const int *ss, long long ss_stride, long long n, uint2 *out) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int v[4] = {0, 0, 0, 0};
        static_for<M::NC>([&](auto CI_) __attribute__((always_inline)) { constexpr int CI = CI_;
            constexpr CalcDesc cd = M::calc[CI];
            if constexpr (cd.kind == CALC_SPLICE_PRE) v[cd.param] = clamp16(kp->calc_value[CI] + ss[cd.param * ss_stride + p]);
            if constexpr (cd.kind == CALC_SPLICE_POST) v[cd.param] = clamp16(ss[cd.param * ss_stride + p]);
        });
        uint2 o;
        o.x = ((unsigned)v[0] & 0xffffu) | ((unsigned)v[1] << 16);
        o.y = ((unsigned)v[2] & 0xffffu) | ((unsigned)v[3] << 16);
        out[p] = o;
    }
}

// NW cooperating waves per PAIR of jobs: workgroup p of the queue runs jobs 2p and 2p + 1 (the last one alone when the
// launch holds an odd number: its high half repeats it)
// (IO 1: `tdense` is the launch's residue-code table -- [0, 24) code -> dense index, [24, 24 + NCODE) dense index -> code; the
// host takes this kernel only when every query fits NW strips and the targets hold at most NCODE codes)
template <class M, int R, int NW, int WPE, int VAR = 0, bool DUMP16 = false, int IO = 0, int NCODE_ = 6, bool MEMC = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void viterbi16_kernel_mw(const KParams *kparams, DevSeqs seqs, const DevJob *jobs, int n_jobs, DevResult *results,
                         DevScratch scratch, int *queue, const uint8_t *tdense = nullptr) {
    using DP = WaveDP16<M, R, VAR, DUMP16, IO, NCODE_, MEMC>;
    // IO 1 leaves the launch constants in memory (a strip reads them once) and spends the LDS on the column stages (4 KB per
    // wave, 4 KB-aligned: the running stage address wraps with one v_and_or), the query profiles and the rings: 52.2 KB per
    // workgroup, three workgroups per CU
    __shared__ __attribute__((aligned(4096))) int stage_mem[IO ? NW * DP::STAGE_COLS * DP::STAGE_INTS : 1];
    __shared__ __attribute__((aligned(16))) int prof_mem[IO ? NW * 2 * DP::PROF_INTS : 1];
    __shared__ __attribute__((aligned(16))) int edge_cols[IO ? 2 * DP::BND : 1];
    __shared__ __attribute__((aligned(16))) int kp_stage[IO ? 1 : (sizeof(KParams) + 3) / 4];
    __shared__ int next_job;
    __shared__ __attribute__((aligned(16))) int rings[(NW > 1 ? NW - 1 : 1) * DP::RING * DP::BND];
    __shared__ int wave_best[NW][2][4];
    __shared__ int progress[NW];
    if constexpr (IO == 0) {
        const int *src = reinterpret_cast<const int *>(kparams);
        int *dst = kp_stage;
        for (int x = threadIdx.x; x < (int)(sizeof(KParams) / sizeof(int)); x += 64 * NW) dst[x] = src[x];
    } else {
        if (threadIdx.x == 0) DP::write_empty_column((typename DP::lds_int *)edge_cols);
    }
    __syncthreads();
    // wave-uniform by construction: said to the compiler, so that what follows from them (job descriptions, sequence and ring
    // pointers, strip bounds) lives in scalar registers
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *bnd = scratch.bnd + (long long)blockIdx.x * scratch.bnd_stride;
    if (threadIdx.x == 0) DP::write_empty_column(bnd);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int n_pairs = (n_jobs + 1) / 2;
    for (;;) {
        if (threadIdx.x == 0) next_job = atomicAdd(queue, 1);
        __syncthreads();
        const int pid = __builtin_amdgcn_readfirstlane(next_job);
        __syncthreads();
        if (pid >= n_pairs) break;
        const int ia = 2 * pid, ib = (2 * pid + 1 < n_jobs) ? 2 * pid + 1 : 2 * pid;
        DP dp{};                 // every member starts defined (c4_viterbi_kernel.h, viterbi_kernel)
        dp.lane = threadIdx.x & 63;
        dp.carry_ok = scratch.carry != 0;
        if constexpr (IO == 1) {
            dp.kp = kparams;
            dp.tdense = tdense;
            static_for<2>([&](auto H_) __attribute__((always_inline)) { constexpr int H = H_;
                dp.prof_a[H] = DP::lds_addr((typename DP::lds_int *)prof_mem + (wid * 2 + H) * DP::PROF_INTS) + dp.lane * DP::PROF_EB;
            });
            dp.template run_mw<NW>(jobs[ia], jobs[ib], seqs, bnd, (typename DP::lds_int *)rings, wid, (typename DP::lds_int *)progress,
                                   (typename DP::lds_int *)stage_mem + wid * DP::STAGE_COLS * DP::STAGE_INTS, (typename DP::lds_int *)edge_cols);
        } else {
            dp.kp = reinterpret_cast<const KParams *>(kp_stage);
            dp.template run_mw<NW>(jobs[ia], jobs[ib], seqs, bnd, (typename DP::lds_int *)rings, wid, (typename DP::lds_int *)progress);
        }
        dp.reduce_best();
        if (dp.lane == 0)
            for (int h = 0; h < 2; h++) {
                wave_best[wid][h][0] = dp.best[h]; wave_best[wid][h][1] = dp.best_i[h]; wave_best[wid][h][2] = dp.best_j[h];
                wave_best[wid][h][3] = dp.best_set[h];
            }
        __syncthreads();
        if (threadIdx.x < 2 && (threadIdx.x == 0 || ib != ia)) {
            const int h = threadIdx.x;
            int b = LOW, bi = 0, bj = 0; bool bs = false;
            for (int w = 0; w < NW; w++) {               // row-major-first merge (viterbi.c:778-791)
                const int ob = wave_best[w][h][0], oi = wave_best[w][h][1], oj = wave_best[w][h][2];
                const bool os = wave_best[w][h][3] != 0;
                const bool take = os && (!bs || ob > b || (ob == b && (oj < bj || (oj == bj && oi < bi))));
                if (take) { b = ob; bi = oi; bj = oj; }
                bs = bs || os;
            }
            DevResult res;
            res.flags = bs ? 0 : FLAG_NO_END; res.n_ops = 0; res.n_vsa = 0; res.pad = 0;
            res.last_srp = bs ? (bi & 15) : -1;          // the state END was entered from in the best end cell (-1: no end cell)
            res.cell_size = 1 + M::NDES; res.ops_off = 0;
            for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
            res.score = b; res.end_set = bs; res.qe = bi >> 4; res.te = bj; res.qs = 0; res.ts = 0;
            results[h ? ib : ia] = res;
        }
        __syncthreads();
    }
}

}  // namespace c4k
