// c4_viterbi_kernel.h — the C4 Viterbi recurrence + traceback as a hand-written CDNA4 (gfx950) kernel.
//
// What it computes: exactly the cell recurrence of Viterbi_interpreted (exonerate src/c4/viterbi.c:655-837)
// in all four modes (viterbi.h:104-109) and with/without continuation (viterbi.c:68-76,705-714), for a
// model family whose closed transition table is a compile-time descriptor (c4_device_models.inc): the
// same transition id order, "first valid transition assigns unconditionally, later ones replace on
// strict <" (viterbi.c:766-775), shadow end / start / transport (viterbi.c:396-462), row-major-first end
// cell (viterbi.c:778-791), checkpoint rows + SRP slot (viterbi.c:605-631) and the traceback walk of
// Viterbi_Data_create_Alignment (viterbi.c:342-392) / Viterbi_Checkpoint_traceback (viterbi.c:537-601).
//
// How it maps to CDNA4:
//   * one 64-lane wavefront per (query x target) job; persistent waves pull jobs from an atomic queue
//     (jobs are pre-sorted longest first), so a launch of N >> 256*k jobs keeps every SIMD busy;
//   * lanes tile the QUERY axis: lane l owns R consecutive query rows of a 64*R-row strip and walks the
//     target axis; at wave step s lane l is at column j = s - l (anti-diagonal wavefront).  All live DP
//     state of the anti-diagonal (columns j-1 .. j-max_target_advance of R rows) lives in VGPRs;
//   * the (i-1, .) dependency crosses lanes once per step through DPP `wave_shr:1` register shifts (no LDS
//     round trip); strip-to-strip carry rows go through an L2-resident scratch row, read back 64 columns
//     at a time (coalesced) and broadcast with v_readlane;
//   * the substitution matrix lives in LDS (one ds_read per cell); residues and splice-site score arrays
//     are read coalesced (adjacent lanes = adjacent columns);
//   * integer max-plus only: no MFMA.  The algorithmic HBM traffic is a few bytes per COLUMN, so the
//     kernel is VALU-bound by construction (roofline section).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace c4k {

struct TrDesc { int in, out, aq, at, calc, label; unsigned dst_shadow_mask; };
struct CalcDesc { int kind, param, protect; };
struct ShDesc { int designation, on_target; unsigned src_state_mask; unsigned long long dst_transition_mask; };

#include "c4_device_models.inc"

constexpr int LOW = -987654321;
constexpr int HIGH = 987654321;
constexpr int CELL_MAX = 8;
enum { MODE_SCORE = 0, MODE_PATH = 1, MODE_REGION = 2, MODE_CKPT = 3 };
enum { SCOPE_ANYWHERE = 0, SCOPE_EDGE, SCOPE_QUERY, SCOPE_TARGET, SCOPE_CORNER };
enum { CALC_CONST = 0, CALC_MATCH_DNA, CALC_MATCH_PROTEIN, CALC_MATCH_P2D, CALC_SPLICE_PRE, CALC_SPLICE_POST,
       CALC_PHASE_PRE, CALC_PHASE_POST };
enum { FLAG_OPS_OVERFLOW = 1, FLAG_NO_END = 2 };
enum { LABEL_NONE = 0, LABEL_MATCH = 1 };            // C4_Label, c4.h:78-89

struct KParams {                 // uniform per launch (device memory, staged to LDS)
    int calc_value[16];
    int min_intron, max_intron;
    int start_scope, end_scope;
    int submat[25 * 24];                 // row 24: what a DNA query position inside an annotated CDS scores against every target code
                                         // (MATCH_IMPOSSIBLY_LOW_SCORE, match.c:276-281; Engine::annotated, annotate_qcode_kernel)
    uint8_t codon_row[4096];             // split-codon calcs: 3 x 4-bit base masks -> Submat row of the residue
    // loop_tr[s] >= 0: state s has a self-loop over one target column that adds nothing (an intron), and NO other way from s
    // back to s without a query row can reach the loop's score (Engine::init_host proves it from the parameters): a path
    // sub-alignment of no query rows from s to s is that loop, T times -- answered without running it (viterbi_kernel)
    int loop_tr[16];
};
struct DevSeqs {
    const uint8_t *qcode, *tcode;        // residue -> submat row codes (tcode: codon codes for 1:3 match)
    const long long *qoff, *toff;        // per pair offsets into the concatenated arrays
    const int *tlen;                     // per pair target length (prefetch clamps)
    const int *ss;                       // [4][ss_stride] splice-site scores, same offsets as tcode
    const uint2 *ss16;                   // the packed score pass: per target position the four splice values as clamped 16-bit halves
                                         // with their calc constants folded in (c4_viterbi16_kernel.h, ss16_kernel); NULL: not built
    const uint16_t *tn4;                 // per target position: 4-bit base masks of positions p, p-1, p-2, p-3
    long long ss_stride;
    // sub-optimal blocking (SUB kernels): per job T+2 column entries {first blocked row, 2 * index into sub_rows +
    // more-rows flag} (two ints each), sub_rows holds the blocked
    // query rows (region coordinates) of each column, ascending
    const int *sub_colptr, *sub_rows;
    // span models: start cells in / END cells out, [(i * (T+1)) + j][1 + designations] per job (DevJob::span_off)
    const int *span_in;
    int *span_out;
    // SEED kernels: column dumps of the score pass, read back by the windowed region pass (see WaveDP)
    int *seed;
};
struct DevJob {
    int pair, q0, t0, Q, T;
    int first_state, final_state, cp_count;
    int tshift, root;                    // packed region start: (query_start << tshift) | target_start; root: the state the path's END
                                         // is entered from where a pass restricted to that state's component wants it (Roots; 0: not known)
    int first_cell[CELL_MAX];
    long long ops_off;                   // into the ops byte array (PATH)
    int ops_cap, vsa_off;                // vsa_off: into the DevVsa array (CKPT)
    long long ckpt_off;                  // >= 0: also dump checkpoint cells there (tests); -1: wave slab only
    long long sub_off;                   // SUB kernels: this job's column pointers start at sub_colptr + sub_off
    int sub_pt_off, sub_pt_n;            // its points in the launch's point arrays (colptr construction)
    long long span_off;                  // SPAN kernels: this job's matrix starts at span_in/span_out + span_off
    long long seed_off;                  // SEED 1: this job's dumps start at seed + seed_off; SEED 2: the dump to start from (-1: none)
    int seed_kshift, seed_rows;          // dump spacing = 1 << seed_kshift columns; rows per dumped column (Q + 1 of the score pass)
    // SEED 2, windows chained on the device: a window whose corner payload is the identity of a dumped cell goes on, in the
    // same workgroup, with the window one dump interval further left (what the host did between launches in round 2)
    long long seed_base;                 // the pair's first dump (seed_off of dump d = seed_base + (d - 1) * DC * seed_rows * SEEDW)
    int win_d, win_t0w;                  // the dump this window starts from (0: column 0, no dump); lattice column of window column 0
    int win_t0_base, win_hops;           // target_start of the whole-rectangle pass; windows one job may run (0: no chaining)
};
struct DevResult {
    int score, qs, ts, qe, te, end_set, last_srp, n_ops, flags, n_vsa, cell_size, pad;
    long long ops_off;                   // PATH: first run of this job in the compact run array
    int final_cell[CELL_MAX];
};
struct DevVsa { int qs, ts, ql, tl, first_state, pad[3]; int final_cell[CELL_MAX]; };
struct DevScratch {                      // per persistent wave slabs
    int *bnd;       long long bnd_stride;
    int carry;                                    // 0: bnd holds one dummy column per workgroup
    uint32_t *tb;   long long tb_stride;
    int *ckpt;      long long ckpt_stride;
    int *ckpt_dump;
    uint32_t *runs; long long runs_stride;        // per wave: run-length encoded path being walked
    uint32_t *runs_out; long long runs_capacity;  // all jobs: compacted (transition << 24 | length) runs
    unsigned long long *runs_used;
};

// compile-time index object: the conversion is always-inlined so that every array index is a literal
// before the first SROA run (a std::integral_constant conversion is an ordinary call at that point and
// leaves the per-lane DP state in scratch memory)
template <int V>
struct IC {
    static constexpr int value = V;
    __device__ __forceinline__ constexpr operator int() const { return V; }
};
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// ---- compile-time facts about a model descriptor ---------------------------------------------------------
template <class M>
struct Facts {
    static constexpr int fanin(int s) { int n = 0; for (int k = 0; k < M::NT; k++) n += (M::tr[k].out == s); return n; }
    static constexpr int bits(int s) { int f = fanin(s) + 1, b = 0; while ((1 << b) < f) b++; return b; }
    static constexpr int shift(int s) { int n = 0; for (int x = 0; x < s; x++) n += bits(x); return n; }
    static constexpr int code(int k) { int n = 1; for (int x = 0; x < k; x++) n += (M::tr[x].out == M::tr[k].out); return n; }
    static constexpr int total_bits = shift(M::NS);
    // does state s have an outgoing transition with advance_query > 0 (its value crosses lanes)?
    static constexpr bool exported(int s) { for (int k = 0; k < M::NT; k++) if (M::tr[k].in == s && M::tr[k].aq > 0) return true; return false; }
    static constexpr int n_exported() { int n = 0; for (int s = 0; s < M::NS; s++) n += exported(s); return n; }
    static constexpr bool owns_shadow(int s, int d) { for (int h = 0; h < M::NSH; h++) if (M::sh[h].designation == d && (M::sh[h].src_state_mask >> s & 1)) return true; return false; }
    static constexpr int consumed_designation(int k) { for (int h = 0; h < M::NSH; h++) if (M::tr[k].dst_shadow_mask >> h & 1) return M::sh[h].designation; return -1; }
    static constexpr int match_at() { for (int k = 0; k < M::NT; k++) if (M::tr[k].calc >= 0 && M::calc[M::tr[k].calc].kind >= CALC_MATCH_DNA && M::calc[M::tr[k].calc].kind <= CALC_MATCH_P2D) return M::tr[k].at; return 1; }
    static constexpr bool has_phase() { for (int c = 0; c < M::NC; c++) if (M::calc[c].kind == CALC_PHASE_POST) return true; return false; }
    static constexpr bool has_splice() { for (int c = 0; c < M::NC; c++) if (M::calc[c].kind == CALC_SPLICE_PRE || M::calc[c].kind == CALC_SPLICE_POST) return true; return false; }
    // every state a MATCH-labelled transition enters also has a silent, calc-free transition from START (score 0
    // wherever START is valid): a blocked match transition may then compete with the unset score instead of
    // being skipped
    static constexpr bool match_states_have_start() {
        for (int k = 0; k < M::NT; k++) {
            if (M::tr[k].label != 1) continue;
            bool found = false;
            for (int x = 0; x < M::NT; x++)
                if (M::tr[x].in == M::START && M::tr[x].out == M::tr[k].out && M::tr[x].aq == 0 && M::tr[x].at == 0 &&
                    M::tr[x].calc < 0) found = true;
            if (!found) return false;
        }
        return true;
    }
    static_assert(M::MAXAQ == 1, "lanes exchange exactly one query row per step");
    static_assert(total_bits <= 32, "traceback word");
};

template <class M, int X>
struct Cell {
    int sc[M::NS];
    int ex[M::NS][X > 0 ? X : 1];
};

__device__ __forceinline__ int dpp_shr1(int old, int v) {      // lane l <- lane l-1 ; lane 0 keeps `old`
    return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// one global_store_dword at base + OFF bytes (an immediate): the compiler neither merges it with its neighbours nor
// waits for it; for data no later instruction of the kernel reads
template <int OFF>
__device__ __forceinline__ void store_dword(int *base, int v) {
    static_assert(OFF >= 0 && OFF < 4096, "global_store immediate offset");
    asm volatile("global_store_dword %0, %1, off offset:%2" : : "v"(base), "v"(v), "n"(OFF) : "memory");
}

__device__ __forceinline__ bool scope_ok(int scope, bool at_q, bool at_t) {
    switch (scope) {
        case SCOPE_ANYWHERE: return true;
        case SCOPE_EDGE: return at_q || at_t;
        case SCOPE_QUERY: return at_q;
        case SCOPE_TARGET: return at_t;
        default: return at_q && at_t;
    }
}

// -------------------------------------------------------------------------------------------------------------
// One DP job on one wavefront.
// -------------------------------------------------------------------------------------------------------------
// PACK (FIND_REGION only): the two region-start slots (viterbi.c:403-412) share one int,
// (query_start << tshift) | target_start; the host picks PACK when bits(Q) + bits(T) <= 31.
// SUB: sub-optimal blocking (SubOpt_Index, src/c4/subopt.c): MATCH transitions skip the blocked cells listed
// per column in DevSeqs::sub_colptr / sub_rows.
// SPAN: BSDP's span models exchange cells with the host through the model's cell_start_func / cell_end_func
// (viterbi.c:728-741,793-799): 1 = transitions out of START read the start cell of their position from a
// matrix (score and shadow slots), 2 = every cell that reaches END is copied out to a matrix.
// SEED: FIND_REGION in two passes for long targets.  The payload that makes FIND_REGION dearer than FIND_SCORE (the
// region start carried beside every score) only matters along the one path that ends in the best cell, which spans a
// small part of a long target.  SEED = 1 (FIND_SCORE): besides the score and the end cell, every lane copies out the
// cells of its rows at the max_target_advance columns that end in d*K (d = 1, 2, ..): all a later column can read.  SEED = 2
// (FIND_REGION): the pass starts from such a dump instead of from column 0 — cells of those first columns are
// loaded, not computed, so every later cell, winner and tie-break is that of the whole-rectangle pass — and ends in
// the corner cell, whose region-start payload is reported.  A payload that entered through the dump carries the
// identity of its entry cell (row, state, column) instead of a start: the host then walks one dump further left
// (find_path_batch), so the payload work is done over the alignment's own extent only.
// The components of a model as seen from END.  A path ends with a transition into END from one state (its "root"); every
// state of the path can reach that root, so a pass that only has to reproduce ONE path whose root is known (the region
// windows, the checkpoint pass: the whole-rectangle score pass has already decided which transition into END won, and a
// pass restricted to the path's region can only confirm it: c4_win16_kernel.h, c4_ckpt16_kernel.h) needs the states from
// which the root can be reached and no others.  est2genome: the forward-strand states {2, 3, 4, 8} and the reverse-strand
// states {5, 6, 7, 9} never feed each other (they only share START and END), so such a pass computes half the model.
// ROOT = -1 stands for "every inner state" (a model with one component, or a root that is not known).
template <class M>
struct Roots {
    static constexpr bool inner(int s) { return s != M::START && s != M::END; }
    static constexpr int count() {
        int n = 0;
        for (int k = 0; k < M::NT; k++) {
            if (M::tr[k].out != M::END || !inner(M::tr[k].in)) continue;
            bool seen = false;
            for (int x = 0; x < k; x++) if (M::tr[x].out == M::END && M::tr[x].in == M::tr[k].in) seen = true;
            n += seen ? 0 : 1;
        }
        return n;
    }
    static constexpr int root(int idx) {           // the idx-th distinct source state of END, in transition order
        int n = 0;
        for (int k = 0; k < M::NT; k++) {
            if (M::tr[k].out != M::END || !inner(M::tr[k].in)) continue;
            bool seen = false;
            for (int x = 0; x < k; x++) if (M::tr[x].out == M::END && M::tr[x].in == M::tr[k].in) seen = true;
            if (seen) continue;
            if (n == idx) return M::tr[k].in;
            n++;
        }
        return -1;
    }
    static constexpr bool member(int rt, int s) {  // can inner state s reach rt (rt itself included)?  rt < 0: every inner state
        if (!inner(s)) return false;
        if (rt < 0) return true;
        bool in[M::NS] = {};
        in[rt] = true;
        for (int it = 0; it < M::NS; it++)
            for (int k = 0; k < M::NT; k++)
                if (in[M::tr[k].out] && inner(M::tr[k].in)) in[M::tr[k].in] = true;
        return in[s];
    }
    // the first root whose component holds s (-1: none)
    static constexpr int root_of(int s) { for (int r = 0; r < count(); r++) if (member(root(r), s)) return root(r); return -1; }
    // do the components of the roots overlap?  (then restricting a pass to one of them saves nothing worth a kernel)
    static constexpr bool disjoint() {
        for (int s = 0; s < M::NS; s++) { int n = 0; for (int r = 0; r < count(); r++) n += member(root(r), s); if (n > 1) return false; }
        return count() >= 2;
    }
};

template <class M, int R, int MODE, bool CONT, bool LOCAL, bool PACK = false, bool SUB = false, int SPAN = 0, int SEED = 0, int COMP = 0>
struct WaveDP {
    using F = Facts<M>;
    static constexpr int NDES = M::NDES;
    static constexpr int NRS = (MODE == MODE_REGION) ? (PACK ? 1 : 2) : 0;
    // split-codon models keep, next to the intron-start shadow, the two bases in front of the intron (an
    // internal slot: the phase calcs of phase.c:188-208 re-read exactly those bases through the shadow)
    static constexpr int NAUX = F::has_phase() ? 1 : 0;
    static constexpr int AUX = NDES;
    static constexpr int X = NDES + NAUX + NRS + (MODE == MODE_CKPT ? 1 : 0);      // register slots per state
    static constexpr int XS = X > 0 ? X : 1;
    static constexpr int RSQ = NDES + NAUX, RST = NDES + NAUX + 1, SRP = NDES + NAUX;
    static_assert(NAUX == 0 || NDES == 1, "the base slot rides on designation 0");
    // reference cell layout (viterbi.c:154-173): score, designations, [region q, region t], [checkpoint]
    static constexpr int CS = 1 + NDES + (MODE == MODE_REGION ? 2 : 0) + (MODE == MODE_CKPT ? 1 : 0);
    static constexpr int W = 64 * R;                    // query rows per strip
    static constexpr int NCOL = M::MAXAT + 1;           // live columns, kept as a ring (no register rotation)
    static constexpr int NEXP = F::n_exported();
    // sub-optimal blocking in the local score / region passes: see eval_cell
    static constexpr bool BLOCK_AS_LOW = LOCAL && (MODE == MODE_SCORE || MODE == MODE_REGION) && F::match_states_have_start();
    static constexpr int BND = NEXP * (1 + XS);         // ints per column in the strip carry row
    static constexpr int XD = NDES + NAUX;              // shadow-like slots of a dumped cell
    static constexpr int DC = M::MAXAT;                 // columns per dump: d*K - (DC - 1) .. d*K
    static_assert(SEED == 0 || (SEED == 1 && MODE == MODE_SCORE) || (SEED == 2 && MODE == MODE_REGION && PACK),
                  "dumps are written by the score pass and read by the packed region pass");
    static_assert(SEED == 0 || (!CONT && !SUB && SPAN == 0), "seeded passes: plain whole-rectangle kernels only");
    using C = Cell<M, X>;

    // COMP > 0: only the states of ONE component of the model are computed -- the inner states that can reach Roots<M>::root(COMP - 1),
    // the state the path's END is entered from (est2genome: one strand's four states and the transitions between them: half the
    // model).  Exact for every call whose path is known to lie in that component: a sub-alignment between two checkpoint cells of an
    // alignment with that root starts in a state of the component (or in START) and nothing outside the component can be reached
    // from there -- the two strands only share START and END -- so the states left out hold -987654321 in the full kernel too.
    // Same traceback word layout, same carry row layout (slots of the states left out are neither written nor read).
    using RTc = Roots<M>;
    static constexpr bool alive(int s) {
        if (COMP == 0 || s == M::START || s == M::END) return true;
        return RTc::member(RTc::root(COMP - 1), s);
    }
    static constexpr unsigned alive_mask() { unsigned m = 0; for (int s = 0; s < M::NS; s++) m |= alive(s) ? (1u << s) : 0u; return m; }
    static constexpr bool tr_alive(int k) {
        if (COMP == 0) return true;
        if (M::tr[k].out == M::END) return M::tr[k].in == M::START || alive(M::tr[k].in);
        return alive(M::tr[k].out) && alive(M::tr[k].in);
    }

    // Is slot e of state s ever read?  A designation slot only matters while a path to a consuming
    // transition exists that does not pass through a state that re-starts the shadow (e.g. est2genome:
    // only the two intron states).  Dead slots are neither stored nor transported; cells leaving the
    // kernel carry 0 there.
    static constexpr bool slot_live(int s, int e) {
        if (NAUX && e == AUX) e = 0;
        if (e >= NDES) return true;
        bool live[M::NS] = {};
        if (SPAN == 2) live[M::END] = true;          // the END cell leaves the kernel (cell_end_func)
        for (int it = 0; it < M::NS; it++)
            for (int k = 0; k < M::NT; k++) {
                const int in = M::tr[k].in, out = M::tr[k].out;
                if (F::consumed_designation(k) == e) live[in] = true;
                else if (!F::owns_shadow(in, e) && live[out]) live[in] = true;
            }
        return live[s];
    }

    // a dumped cell row: the NS scores, then the shadow-like slots something can still read (slot_live), in (state, slot) order
    static constexpr int dump_pos(int s_, int e_) {
        int n = M::NS;
        for (int s = 0; s < M::NS; s++)
            for (int e = 0; e < NDES + NAUX; e++) {
                if (s == s_ && e == e_) return n;
                if (slot_live(s, e)) n++;
            }
        return n;
    }
    static constexpr int SEEDW = dump_pos(M::NS, 0);    // ints per row of a dumped column

    // job / launch constants
    const KParams *kp;      // in LDS
    const uint8_t *qc, *tc;
    const int *ss0, *ss1, *ss2, *ss3;
    int Q, T, q0, t0, lane, tshift;
    int first_state, final_state, min_intron, max_intron, seed_aux;
    unsigned intron_span;       // max_intron - min_intron (>= 0: c4gpu_params are validated by the host)
    const int *first_cell;
    int start_scope, end_scope;

    // per-lane DP state: col[p] = the R cells evaluated at the step with phase p = s % NCOL,
    // nbr[p] = the cell above them (row i0-1 of that column), expo = our bottom row for the lane below
    C col[NCOL][R], nbr[NCOL], expo;
    int qcode[R];
    int best, best_i, best_j, best_qs, best_ts;
    bool best_set;
    int corner[CELL_MAX];
    bool corner_set;

    // uniform base pointer + 32-bit byte offset: the form the global_load "saddr" encoding takes (base in
    // SGPRs, one VGPR offset shared by the four splice arrays) instead of a 64-bit address per load
    __device__ __forceinline__ int splice(int k, unsigned byte_off) const {
        const int *p = k == 0 ? ss0 : k == 1 ? ss1 : k == 2 ? ss2 : ss3;
        return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(p) + byte_off);
    }

    // cell slots in the reference layout (for cells that leave the kernel)
    template <int S>
    __device__ __forceinline__ void export_cell(const C &c, int *out) const {
        out[0] = c.sc[S];
        static_for<NDES>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
            out[1 + E] = slot_live(S, E) ? c.ex[S][E] : 0;
        });
        if constexpr (MODE == MODE_REGION) {
            if constexpr (PACK) {
                out[1 + NDES] = c.ex[S][RSQ] >> tshift;
                out[2 + NDES] = c.ex[S][RSQ] & ((1 << tshift) - 1);
            } else {
                out[1 + NDES] = c.ex[S][RSQ];
                out[2 + NDES] = c.ex[S][RST];
            }
        }
        if constexpr (MODE == MODE_CKPT) out[1 + NDES] = c.ex[S][SRP];
    }

    // ---- one cell -----------------------------------------------------------------------------------------
    // RR: row inside the lane, PH: ring phase of this step (both compile time).
    // JINT: every lane is at max_at <= j <= T (main loop).
    // NOTE on style: every select works on scalars that were loaded first and uses the non-short-circuit
    // operators (& |): a `cond ? x : mem[..]` arm is a load under control flow, which InstCombine (run
    // BEFORE the always-inliner) turns into a load of a selected address, and that keeps the whole per-lane
    // state in scratch memory instead of VGPRs.
    template <int RR, int PH, bool JINT>
    __device__ __forceinline__ void eval_cell(int i, int j, bool active, int mscore, const int (&pre)[4],
                                              int qrow, int tn4col, bool blocked, uint32_t &tbword, bool &end_ok) {
        C &c = col[PH][RR];
        bool set[M::NS];
        static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            set[S] = false;
            c.sc[S] = LOW;
        });
        if constexpr (X > 0) {
            // the START cell's slots are zero unless seeded by a continuation (calloc'd, never written)
            static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; c.ex[M::START][E] = 0; });
        }
        // Row 0 has no row above it: transitions that advance the query are invalid there (layout.c:122-154).
        // In the local score / region passes (START reachable with score 0 in every cell) the mask is not
        // needed: the phantom row above reads as unset (-987654321), what such a transition then proposes stays
        // within a few thousand of that value, and every state that can lie on a path to END has a candidate
        // that is hundreds of millions higher in the same cell — the same reason the reference's own "unset
        // states still propagate" never shows in a result.  States that are unset in row 0 hold a different
        // near-minimum number; nothing reads them on the way to a reported score, end cell or region start.
        // CONT && LOCAL (continuations: FIND_PATH between checkpoints, FIND_CHECKPOINTS) is the same shortcut for the one
        // row that has no row above: a continuation's scopes stay CORNER (START only in the origin cell, END only in the far
        // corner, below), but the transitions that advance the query are evaluated in row 0 as everywhere else.  What they
        // read there is the empty column (-987654321, slots 0), so a state the reference leaves unset in row 0 holds a
        // number within the launch's (Q + T) x largest calc of that value instead; every cell a path from the seeded origin
        // to the corner can visit has a candidate of real magnitude and takes it (first valid transition assigns, later
        // ones replace on strict <: both forms pick the same transition among the real candidates, in the same order).
        // The host only picks these kernels while that bound keeps the two ranges apart (Engine::cont_free_ok).
        const bool i_ok = (RR > 0) | (i > 0) | (LOCAL && (CONT || MODE == MODE_SCORE || MODE == MODE_REGION));
        uint32_t tbw = 0;
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_;
            constexpr int k = K;
            constexpr TrDesc t = M::tr[k];
            // a transition of the other component (COMP) is skipped -- except that a transition out of START still does the
            // continuation's seeding below at ITS place in the order: the reference seeds the first cell inside its transition
            // loop, at the first valid transition out of START (viterbi.c:705-714), i.e. after the silent transitions with smaller
            // ids have been evaluated in the origin cell and before the others (est2genome: 6 -> 5 and 7 -> 5 before, 3 -> 2 and
            // 4 -> 2 after), whichever component that transition belongs to
            if constexpr (!tr_alive(k) && !(CONT && t.in == M::START)) return;
            // Layout_transition_is_valid (layout.c:122-154)
            bool valid = true;
            if constexpr (t.aq > 0) valid = valid & i_ok;
            if constexpr (t.at > 0 && !JINT) valid = valid & (j >= t.at);
            if constexpr (t.in == M::START && (!LOCAL || CONT))
                valid = valid & (CONT ? ((i - t.aq == 0) & (j - t.at == 0))
                                      : scope_ok(start_scope, i - t.aq == 0, j - t.at == 0));
            if constexpr (t.out == M::END && (!LOCAL || CONT))
                valid = valid & (CONT ? ((i == Q) & (j == T)) : scope_ok(end_scope, i == Q, j == T));
            // sub-optimal blocking: MATCH transitions do not enter a blocked cell (viterbi.c:701-704)
            // In the local score / region passes the match state always has START's candidate (score 0) in the
            // same cell, so a blocked match transition may as well compete with the unset score and lose: the
            // validity of everything stays a compile-time fact there (see the row-0 note above).
            if constexpr (SUB && t.label == LABEL_MATCH && !BLOCK_AS_LOW)
                valid = valid & !blocked;
            // A continuation runs with CORNER scopes (viterbi.c:68-76): a transition out of START is valid in
            // the origin cell only.  Instantiations that cannot hold the origin (rows below the lane's first,
            // steps where every lane is past column 0) drop those transitions at compile time.
            if constexpr (CONT && t.in == M::START && ((RR > 0 && t.aq == 0) || (JINT && t.at == 0))) return;
            // ... and a transition into END only in the far corner (Q, T): one cell of the whole job
            if constexpr (CONT && t.out == M::END) {
                if (!__builtin_amdgcn_ballot_w64(valid)) return;
            }
            // continuation seeding (viterbi.c:705-714): first cell into the first state, at the corner only;
            // behind a wave-uniform branch, it happens in one cell of the whole job
            if constexpr (CONT && t.in == M::START) {
                if (__builtin_amdgcn_ballot_w64(valid)) {
                    static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                        if constexpr (!alive(S)) return;             // (COMP: the first state is one of the component's)
                        const bool seed = valid & (first_state == S);
                        const int old_sc = c.sc[S], fc0 = first_cell[0];
                        c.sc[S] = seed ? fc0 : old_sc;
                        static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                            if constexpr (X > 0) {
                                if constexpr (slot_live(S, E)) {
                                    // register slot -> reference cell slot (the base slot is re-derived)
                                    const int old_ex = c.ex[S][E];
                                    const int fce = (NAUX && E == AUX) ? seed_aux : first_cell[1 + E - (E > AUX ? NAUX : 0)];
                                    c.ex[S][E] = seed ? fce : old_ex;
                                }
                            }
                        });
                        set[S] = set[S] | seed;
                    });
                }
            }
            if constexpr (!tr_alive(k)) return;                  // (COMP: seeded above where it is a transition out of START)
            // source cell: same cell (silent), row above (lane-local or the neighbour's), earlier columns
            constexpr int PD = (PH - t.at + NCOL) % NCOL;
            const C &cell_src = (t.aq == 0) ? col[PD][RR] : (RR > 0 ? col[PD][RR > 0 ? RR - 1 : 0] : nbr[PD]);
            // cell_start_func (viterbi.c:728-741): the START cell of position (i - aq, j - at) comes from the
            // job's matrix, score and shadow slots (clamped, unconditional loads; invalid transitions ignore them)
            C start_cell;
            if constexpr (SPAN == 1 && t.in == M::START && !CONT) {
                const int si = i - t.aq < 0 ? 0 : i - t.aq, sj = j - t.at < 0 ? 0 : (j - t.at > T ? T : j - t.at);
                const int *sc = span_in_p + ((long long)(si > Q ? Q : si) * (T + 1) + sj) * (1 + NDES);
                start_cell.sc[M::START] = sc[0];
                static_for<NDES>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; start_cell.ex[M::START][E] = sc[1 + E]; });
                if constexpr (NAUX == 1) {
                    // the base slot is not part of the reference's cell: re-derive it from the shadow (the two
                    // bases in front of the intron that is open in the start cell), as a continuation does
                    const int cis = sc[1];
                    const int cq = cis < 1 ? 0 : (cis - 1 > tlast ? tlast : cis - 1);
                    const int b = tn4p[cq] & 0xff;
                    start_cell.ex[M::START][AUX] = ((cis >= 1) & (cis - 1 <= tlast)) ? b : 0;
                }
            }
            const C &src = (SPAN == 1 && t.in == M::START && !CONT) ? start_cell : cell_src;
            int tscore;
            if constexpr (t.in == M::START) tscore = CONT ? src.sc[M::START] : (SPAN == 1 ? src.sc[M::START] : 0);
            else tscore = src.sc[t.in];
            // calc (C4_Calc_score, c4.c:1700)
            if constexpr (t.calc >= 0) {
                constexpr CalcDesc cd = M::calc[t.calc];
                if constexpr (cd.kind == CALC_CONST) {
                    tscore += kp->calc_value[t.calc];
                } else if constexpr (cd.kind >= CALC_MATCH_DNA && cd.kind <= CALC_MATCH_P2D) {
                    tscore += mscore;
                } else if constexpr (cd.kind == CALC_SPLICE_PRE) {
                    tscore += pre[cd.param];             // open penalty + ss[param][tpos], hoisted per column
                } else if constexpr (cd.kind == CALC_SPLICE_POST) {
                    // Intron_calc_*: the shadow end func has just loaded curr_intron_start (intron.c:468)
                    constexpr int des = F::consumed_designation(k);
                    static_assert(des >= 0, "post-splice calc without a shadow");
                    static_assert(slot_live(t.in, des), "consumed slot must be live");
                    // length < min || length > max as one unsigned comparison of (length - min)
                    const int intron_length = (t0 + j - t.at) - src.ex[t.in][des] + 2;
                    const bool bad = (unsigned)(intron_length - min_intron) > intron_span;
                    const int ssv = pre[cd.param];
                    tscore += bad ? LOW : ssv;
                } else if constexpr (cd.kind == CALC_PHASE_POST) {
                    // Phase_{1,2}_PROTEIN2DNA_FALSE_TRUE_calc_func (phase.c:188-213): the codon split by the
                    // intron = base(s) kept from in front of the intron + base(s) of this column
                    constexpr int des = F::consumed_designation(k);
                    static_assert(des == 0 && NAUX == 1, "phase calc needs the shadow and its base slot");
                    const int cis = src.ex[t.in][des], aux = src.ex[t.in][AUX];
                    constexpr int nib = t.at - 1;                 // base at target_pos = t0 + j - at
                    int codon;
                    if constexpr (cd.param == 1)
                        codon = (aux & 15) | (((tn4col >> (4 * nib)) & 15) << 4) | (((tn4col >> (4 * (nib - 1))) & 15) << 8);
                    else
                        codon = ((aux >> 4) & 15) | ((aux & 15) << 4) | (((tn4col >> (4 * nib)) & 15) << 8);
                    const int row = kp->codon_row[codon];
                    const int psc = kp->submat[qrow * 24 + row];
                    tscore += (cis < cd.param) ? LOW : psc;
                }
                if constexpr (cd.protect & 2) tscore = tscore < LOW ? LOW : tscore;
                if constexpr (cd.protect & 1) tscore = tscore > HIGH ? HIGH : tscore;
            }
            if constexpr (SUB && t.label == LABEL_MATCH && BLOCK_AS_LOW)
                tscore = blocked ? LOW : tscore;
            const bool was_set = set[t.out];
            const int old_sc = c.sc[t.out];
            const bool win = valid & (!was_set | (old_sc < tscore));
            // Viterbi_Data_assign (viterbi.c:445-462)
            c.sc[t.out] = win ? tscore : old_sc;
            if constexpr (X > 0) {
                static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                    constexpr int e = E;
                    if constexpr (slot_live(t.out, e)) {
                        int v = 0;
                        if constexpr (e < NDES) {
                            if constexpr (F::owns_shadow(t.in, e)) v = t0 + j - t.at;      // intron.c:454-458
                            else if constexpr (slot_live(t.in, e)) v = src.ex[t.in][e];
                        } else if constexpr (NAUX == 1 && e == AUX) {
                            // bases at (shadow - 1) and (shadow - 2), shadow = t0 + j - at
                            if constexpr (F::owns_shadow(t.in, 0)) v = (tn4col >> (4 * t.at)) & 0xff;
                            else if constexpr (slot_live(t.in, e)) v = src.ex[t.in][e];
                        } else if constexpr (MODE == MODE_REGION && t.in == M::START) {     // viterbi.c:403-412
                            if constexpr (PACK) v = ((i - t.aq) << tshift) | (j - t.at);
                            else v = (e == RSQ) ? (i - t.aq) : (j - t.at);
                        } else {
                            v = src.ex[t.in][e];
                        }
                        const int old_ex = c.ex[t.out][e];
                        c.ex[t.out][e] = win ? v : old_ex;
                    }
                });
            }
            if constexpr (MODE == MODE_PATH) {
                constexpr uint32_t mask = ((1u << F::bits(t.out)) - 1u) << F::shift(t.out);
                constexpr uint32_t code = (uint32_t)F::code(k) << F::shift(t.out);
                tbw = win ? ((tbw & ~mask) | code) : tbw;
            }
            set[t.out] = was_set | valid;
        });
        tbword = tbw;
        // whether this cell can be the end cell (viterbi.c:778-791): the comparison with the best so far is
        // done once per step over the lane's R cells (step, below)
        end_ok = active & set[M::END];
        // SEED 2: the cells of the window's first two columns are the whole-rectangle pass's, read from its dump; their
        // region-start payload is the cell's own identity.  Only in the steps that hold those columns (wave-uniform branch).
        if constexpr (SEED == 2 && !JINT) {
            const bool sd = seeded & active & ((unsigned)j < (unsigned)DC);
            if (__builtin_amdgcn_ballot_w64(sd)) {
                const int ic = i < 0 ? 0 : (i > Q ? Q : i), jc = j < 0 ? 0 : (j > DC - 1 ? DC - 1 : j);
                const int *p = seed_rd + ((long long)jc * seed_rows + ic) * SEEDW;
                static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                    const int v = p[S], old_sc = c.sc[S];
                    c.sc[S] = sd ? v : old_sc;
                    static_for<XD>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                        if constexpr (slot_live(S, E)) {
                            const int ve = p[dump_pos(S, E)], old_ex = c.ex[S][E];
                            c.ex[S][E] = sd ? ve : old_ex;
                        }
                    });
                    const int ident = -(1 + (((ic * M::NS) + S) * DC + jc)), old_rs = c.ex[S][RSQ];
                    c.ex[S][RSQ] = sd ? ident : old_rs;
                });
                end_ok = end_ok & !sd;
            }
        }
    }

    // ---- cross-lane / cross-strip exchange ----------------------------------------------------------------
    template <class Fn>
    __device__ __forceinline__ static void for_exported(Fn &&fn) {
        int slot = 0;
        static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (F::exported(S)) {
                if constexpr (alive(S)) fn(S_, slot);            // (COMP: the slots of the other component's states stay where they are)
                slot += 1 + XS;
            }
        });
    }

    // lane 0's neighbour row comes from the carry row the previous strip wrote.  Lane 0 is at column
    // j = s, so every lane requests the same (clamped) column one step ahead: an unconditional, uniform
    // load with no dependent ALU op, so nothing waits for it until the next step's DPP exchange.
    C nx_carry;
    static constexpr int RING = 256;                    // columns of an LDS carry ring (multi-wave kernels)
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int *ring_in, *ring_out;
    bool use_ring_in, use_ring_out;                     // the row above / below lives in LDS (LDS offset 0 is a
                                                        // valid address, so a null test cannot tell)
    int cp_next_j, cp_next_i;   // FIND_CHECKPOINTS: column / index of the next checkpoint this lane will cross
    bool carry_ok;      // the launch allocated HBM carry rows (some job has more strips than waves per job)
    bool carry_cols;    // this strip reads real carry columns from HBM (else: one column, see empty_column)
    // The first strip has no row above it: its lane 0 reads an "empty" column (every state unset: -987654321,
    // slots 0) through the same prefetch as a real carry row, so the step needs no first-strip selects.
    __device__ __forceinline__ static void write_empty_column(int *col) {
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) {
            col[slot] = LOW;
            static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; col[slot + 1 + E] = 0; });
        });
    }
    __device__ __forceinline__ void prefetch_carry(int s_next, const int *bnd_in) {
        const int jx = s_next < 0 ? 0 : (s_next > T ? T : s_next);
        const int jc = (carry_cols | use_ring_in) ? jx : 0;    // no row above / no carry rows: every load hits column 0
        if (use_ring_in) {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                const lds_int *p = ring_in + (jc & (RING - 1)) * BND + slot;
                nx_carry.sc[S] = p[0];
                static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                    if constexpr (X > 0) if constexpr (slot_live(S, E)) nx_carry.ex[S][E] = p[1 + E];
                });
            });
        } else {
            for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                const int *p = bnd_in + (long long)jc * BND + slot;
                nx_carry.sc[S] = p[0];
                static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                    if constexpr (X > 0) if constexpr (slot_live(S, E)) nx_carry.ex[S][E] = p[1 + E];
                });
            });
        }
    }

    // per-column inputs of column j: target residue code for the match transitions and, for spliced
    // models, the splice-site scores at the column the (0,2) transitions leave from.  Requested one step
    // ahead from clamped (always valid) addresses; a lane outside the rectangle gets values it never
    // uses, because every transition that would read them is masked invalid.
    int *seed_wr;               // SEED 1: this job's dumps
    const int *seed_rd;         // SEED 2: the two dumped columns this job starts from
    bool seeded;
    int seed_rows, seed_kshift;
    int nx_tcode, nx_sp[4], nx_tn4, tlast;
    const uint16_t *tn4p;
    const int *span_in_p;               // SPAN == 1: this job's start cells
    int *span_out_p;                    // SPAN == 2: this job's END cell matrix
    const int *sub_cp, *sub_rows;       // SUB: per-column entries of this job, blocked rows of the launch
    // per column one 8-byte entry {first blocked row (or SUB_NONE), 2 * index of that row in sub_rows + "more
    // rows follow"}: one load, one step ahead, no load that depends on another
    static constexpr int SUB_NONE = -0x40000000;
    int nx_sub_row0, nx_sub_lo2;                // column j+1 (consumed by the next step)
    __device__ __forceinline__ void sub_load(int j, int &row0, int &lo2) const {
        const int jc = j < 0 ? 0 : (j > T ? T : j);
        const char *p = reinterpret_cast<const char *>(sub_cp) + ((unsigned)jc << 3);      // uniform base + 32-bit offset
        row0 = reinterpret_cast<const int *>(p)[0];
        lo2 = reinterpret_cast<const int *>(p)[1];
    }
    __device__ __forceinline__ void prefetch_column(int j) {
        constexpr int mat = F::match_at();
        if constexpr (SUB) {
            sub_load(j, nx_sub_row0, nx_sub_lo2);
        }
        int ti = t0 + j - mat;
        ti = ti < 0 ? 0 : (ti > tlast ? tlast : ti);
        nx_tcode = tc[(unsigned)ti];
        if constexpr (F::has_phase()) {
            int tq = t0 + j - 1;
            tq = tq < 0 ? 0 : (tq > tlast ? tlast : tq);
            nx_tn4 = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tn4p) + ((unsigned)tq << 1));
        }
        if constexpr (F::has_splice()) {
            int tp = t0 + j - 2;
            tp = tp < 0 ? 0 : (tp > tlast ? tlast : tp);
            const unsigned sp_off = (unsigned)tp << 2;         // targets are below 2^30 residues
            static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_sp[K] = splice(K, sp_off); });
        }
    }

    // The cooperating-wave kernels (run_mw, round 6) read their column inputs from a per-wave STAGE in LDS, as the packed passes do
    // (c4_viterbi16_kernel.h IO 1, c4_ckpt16_kernel.h Stage16): one plane per input -- the residue / codon code, the base masks of
    // the split-codon calcs, the four splice values -- of STG_COLS columns; the wave fetches the columns a chunk reads ahead with
    // one coalesced, clamped load per array and lane once per chunk (fill_stage: what prefetch_column loaded per step), and a step
    // reads its column from LDS.  Where a launch has one wave per SIMD (256 proteins against one chromosome: five working waves
    // of a job, one row per lane) nothing hid the latency of six global loads per step behind a step of ~90 instructions.
    static constexpr int STG_CH = (64 + NCOL - 1) / NCOL * NCOL;           // run_mw's chunk (CH below)
    static constexpr int STG_COLS = STG_CH > 64 ? 256 : 128;              // a chunk reads 63 + CH columns
    static constexpr int STG_PLANES = 1 + (F::has_phase() ? 1 : 0) + (F::has_splice() ? 4 : 0);
    static constexpr int STG_INTS = STG_COLS * STG_PLANES;
    static constexpr int STG_P_TN4 = 1, STG_P_SP = 1 + (F::has_phase() ? 1 : 0);
    int stage_a, stage_base;                            // LDS byte address of the next column's entry; of the wave's stage
    __device__ __forceinline__ void prefetch_staged() {
        const lds_int *p = (const lds_int *)(size_t)(unsigned)stage_a;
        nx_tcode = p[0];
        if constexpr (F::has_phase()) nx_tn4 = p[STG_P_TN4 * STG_COLS];
        if constexpr (F::has_splice())
            static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; nx_sp[K] = p[(STG_P_SP + K) * STG_COLS]; });
        stage_a = ((stage_a + 4) & (STG_COLS * 4 - 1)) | stage_base;
    }
    // columns c0 + lane (lanes below `count`) into the stage
    __device__ __forceinline__ void fill_stage(lds_int *stage, int c0, int count) {
        if (lane >= count) return;
        constexpr int mat = F::match_at();
        const int c = c0 + lane;
        lds_int *p = stage + (c & (STG_COLS - 1));
        int ti = t0 + c - mat;
        ti = ti < 0 ? 0 : (ti > tlast ? tlast : ti);
        p[0] = tc[(unsigned)ti];
        if constexpr (F::has_phase()) {
            int tq = t0 + c - 1;
            tq = tq < 0 ? 0 : (tq > tlast ? tlast : tq);
            p[STG_P_TN4 * STG_COLS] = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(tn4p) + ((unsigned)tq << 1));
        }
        if constexpr (F::has_splice()) {
            int tp = t0 + c - 2;
            tp = tp < 0 ? 0 : (tp > tlast ? tlast : tp);
            const unsigned sp_off = (unsigned)tp << 2;
            static_for<4>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; p[(STG_P_SP + K) * STG_COLS] = splice(K, sp_off); });
        }
    }

    // ---- one wave step: every lane evaluates its R rows of column j = s - lane ---------------------------
    template <bool JINT, int PH, bool STG = false>
    __device__ __forceinline__ void step(int s, int i0, bool first_strip, bool last_strip, const int *bnd_in,
                                         int *bnd_out, uint32_t *tb_slab, long long tb_base, int *ckpt,
                                         int section_length, int cp_count) {
        const int j = s - lane;
        const bool jact = JINT || (j >= 0 && j <= T);
        // (0) substitution scores of this column for our R query rows: LDS reads issued first so that
        // their latency overlaps the lane exchange below
        const int tcode = nx_tcode;
        const int tn4col = F::has_phase() ? nx_tn4 : 0;
        // blocked rows of this column among our R rows: columns that hold any blocked cell are rare, so
        // the list walk sits behind a wave-uniform branch
        unsigned blk = 0;
        if constexpr (SUB) {
            // an earlier alignment has one match cell per column it crosses, so a column's list is almost
            // always empty or one row long: that row arrives with the prefetch; longer lists (several
            // earlier alignments through one column) are walked behind a wave-uniform branch
            const int lo2 = nx_sub_lo2;
            const unsigned r0 = (unsigned)(nx_sub_row0 - i0);          // SUB_NONE - i0 is far outside [0, R)
            blk = r0 < (unsigned)R ? (1u << r0) : 0u;
            if (__builtin_amdgcn_ballot_w64((lo2 & 1) != 0)) {
                if (lo2 & 1) {
                    const int jc = j < 0 ? 0 : (j > T ? T : j);
                    const int hi = sub_cp[2 * (jc + 1) + 1] >> 1;      // the next column's first row
                    for (int p = (lo2 >> 1) + 1; p < hi; p++) {
                        const unsigned r = (unsigned)(sub_rows[p] - i0);
                        blk |= r < (unsigned)R ? (1u << r) : 0u;
                    }
                }
            }
        }
        int ms[R];
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            ms[RR] = kp->submat[qcode[RR] * 24 + tcode];
        });
        int sp[4] = {0, 0, 0, 0};
        if constexpr (F::has_splice()) {
            static_for<M::NC>([&](auto CI_) __attribute__((always_inline)) { constexpr int CI = CI_;
                constexpr CalcDesc cd = M::calc[CI];
                if constexpr (cd.kind == CALC_SPLICE_PRE) sp[cd.param] = kp->calc_value[CI] + nx_sp[cd.param];
                if constexpr (cd.kind == CALC_SPLICE_POST) sp[cd.param] = nx_sp[cd.param];
            });
        }
        // (1) row i0-1 of this column: from lane-1 (DPP) or, for lane 0, from the previous strip's carry row
        // (requested one step ago)
        for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
            const int c_sc = nx_carry.sc[S];
            nbr[PH].sc[S] = dpp_shr1(c_sc, expo.sc[S]);        // first strip: the carry source is the empty column
            static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                if constexpr (X > 0) if constexpr (slot_live(S, E)) {
                    const int c_ex = nx_carry.ex[S][E];
                    nbr[PH].ex[S][E] = dpp_shr1(c_ex, expo.ex[S][E]);
                }
            });
        });
        // (2) request the next step's inputs
        prefetch_carry(s + 1, bnd_in);
        if constexpr (STG) prefetch_staged(); else prefetch_column(j + 1);
        // (3) the R cells of this lane, top to bottom
        uint32_t tbw[R];
        bool end_ok[R];
        static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
            const int i = i0 + RR;
            eval_cell<RR, PH, JINT>(i, j, jact && i <= Q, ms[RR], sp, qcode[RR], tn4col, (blk >> RR) & 1u, tbw[RR],
                                    end_ok[RR]);
        });
        // cell_end_func (viterbi.c:793-799): every cell in which END is set is copied out
        if constexpr (SPAN == 2) {
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                if (end_ok[RR]) {
                    int *ec = span_out_p + ((long long)(i0 + RR) * (T + 1) + j) * (1 + NDES);
                    ec[0] = col[PH][RR].sc[M::END];
                    static_for<NDES>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                        ec[1 + E] = slot_live(M::END, E) ? col[PH][RR].ex[M::END][E] : 0;
                    });
                }
            });
        }
        // (3b) end cell (viterbi.c:778-791): strict improvement in row-major order.  In continuation mode the
        // score is read off the corner cell later.  A new maximum is rare (it only grows along the
        // alignment): one masked maximum over the lane's cells per step decides whether any of them can
        // improve, and the bookkeeping itself sits behind a wave-uniform branch.
        if constexpr (!CONT) {
            constexpr int NEVER = (-2147483647 - 1);         // below every score the DP can produce
            int m = NEVER;
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                const int tsc = col[PH][RR].sc[M::END];
                const int mk = end_ok[RR] ? tsc : NEVER;
                m = m > mk ? m : mk;
            });
            const bool cand = (m != NEVER) & (!best_set | (best < m));
            if (__builtin_amdgcn_ballot_w64(cand)) {
                static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                    const C &c = col[PH][RR];
                    const int tsc = c.sc[M::END], b = best, bi = best_i, bj = best_j;
                    const bool b_set = best_set;
                    const bool upd = end_ok[RR] & (!b_set | (b < tsc));
                    best = upd ? tsc : b;
                    best_i = upd ? i0 + RR : bi;
                    best_j = upd ? j : bj;
                    if constexpr (MODE == MODE_REGION) {
                        const int nqs = c.ex[M::END][RSQ], oqs = best_qs;
                        best_qs = upd ? nqs : oqs;
                        if constexpr (!PACK) {
                            const int nts = c.ex[M::END][RST], ots = best_ts;
                            best_ts = upd ? nts : ots;
                        }
                    }
                    best_set = b_set | upd;
                });
            }
        }
        // (4) traceback words, step-major (fully coalesced)
        if constexpr (MODE == MODE_PATH) {
            uint32_t *p = tb_slab + tb_base + ((long long)s * 64 + lane) * R;
            if (jact) static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_; p[RR] = tbw[RR]; });
        }
        // (5) export the bottom row BEFORE any checkpoint edit (the next lane still needs column j as it was)
        for_exported([&](auto S_, int) __attribute__((always_inline)) { constexpr int S = S_;
            expo.sc[S] = col[PH][R - 1].sc[S];
            static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                if constexpr (X > 0) if constexpr (slot_live(S, E)) expo.ex[S][E] = col[PH][R - 1].ex[S][E];
            });
        });
        if (!last_strip && lane == 63 && jact) {
            if (use_ring_out) {
                for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                    lds_int *p = ring_out + (j & (RING - 1)) * BND + slot;
                    p[0] = expo.sc[S];
                    static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                        if constexpr (X > 0) if constexpr (slot_live(S, E)) p[1 + E] = expo.ex[S][E];
                    });
                });
            } else {
                for_exported([&](auto S_, int slot) __attribute__((always_inline)) { constexpr int S = S_;
                    int *p = bnd_out + (long long)j * BND + slot;
                    p[0] = expo.sc[S];
                    static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                        if constexpr (X > 0) if constexpr (slot_live(S, E)) p[1 + E] = expo.ex[S][E];
                    });
                });
            }
        }
        // (6) the corner cell (Q, T): final cell of a continuation / last SRP (viterbi.c:813-832)
        if constexpr (CONT) {
            if (jact && j == T) {
                static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                    if (i0 + RR == Q) {
                        static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                            if (final_state == S) {
                                export_cell<S>(col[PH][RR], corner);
                                corner_set = true;
                            }
                        });
                    }
                });
            }
        }
        // SEED 1: the DC columns that end in d*K go to the job's dumps.  Lane l is at column s - l, so only the steps
        // with (s + DC - 1) mod K <= DC + 62 can hold such a column: a scalar test keeps every other step free of it.
        if constexpr (SEED == 1) {
            if (((unsigned)(s + DC - 1) & (unsigned)((1 << seed_kshift) - 1)) <= (unsigned)(DC + 62)) {
                const int d = (j + DC - 1) >> seed_kshift;                       // 1-based dump index of column j
                const unsigned which = (unsigned)(j - ((d << seed_kshift) - (DC - 1)));
                if (jact & (d >= 1) & (which < (unsigned)DC) & ((d << seed_kshift) <= T)) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0 + RR;
                        if (i <= Q) {
                            // one dword per store, issued as such: merged x3 / x4 stores would want their values in
                            // consecutive registers and cost the hot loop a third of its register budget (a volatile
                            // store does not merge either, but waits for each one).  Nothing in the kernel reads the dumps.
                            int *p = seed_wr + (((long long)(d - 1) * DC + which) * seed_rows + i) * SEEDW;
                            static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                store_dword<S * 4>(p, col[PH][RR].sc[S]);
                                static_for<XD>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_;
                                    if constexpr (slot_live(S, E)) store_dword<dump_pos(S, E) * 4>(p, col[PH][RR].ex[S][E]);
                                });
                            });
                        }
                    });
                }
            }
        }
        // SEED 2: the corner cell (Q, T) of the window: score and region-start payload of the requested state (lane l
        // reaches column T at step T + l: a scalar test skips the steps before)
        if constexpr (SEED == 2) {
            if (s >= T) {
                if (jact && j == T) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        if (i0 + RR == Q) {
                            static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                if (final_state == S) {
                                    corner[0] = col[PH][RR].sc[S];
                                    corner[1] = col[PH][RR].ex[S][RSQ];
                                    corner_set = true;
                                }
                            });
                        }
                    });
                }
            }
        }
        // (7) checkpoint rows (Viterbi_Checkpoint_process, viterbi.c:605-631).  At checkpoint column c the
        // reference copies rows c, c-1, .. c-(MAXAT-1) and then stamps their SRP slots.  A column never
        // changes after it has been computed, so each of those rows is copied out at the step that
        // computes it (while its cells are in registers anyway): older columns then only keep the
        // states later transitions still read, exactly as in the other modes.  The SRP stamp stays at
        // column c: until then transitions must read the previous checkpoint's SRP.
        if constexpr (MODE == MODE_CKPT) {
            const bool cp_live = jact & (cp_next_i < cp_count);
            static_for<M::MAXAT>([&](auto ROW_) __attribute__((always_inline)) { constexpr int ROW = ROW_;
                if (cp_live & (j + ROW == cp_next_j)) {
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0 + RR;
                        if (i <= Q) {
                            // (measured: unmerged dword stores here cut 9 % of the kernel's instructions and most of
                            // its spills, and cost 20 % in time: 90 store issues per row instead of 23)
                            int *p = ckpt + ((((long long)cp_next_i * M::MAXAT + ROW) * (Q + 1) + i) * M::NS) * CS;
                            static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                                export_cell<S>(col[PH][RR], p + S * CS);
                            });
                        }
                    });
                }
            });
            if (cp_live & (j == cp_next_j)) {
                // the stamps only depend on the lane's rows: keep the compiler from hoisting ~100 of them
                // out of the column loop into registers that would then be live through every step
                int i0v = i0;
                asm volatile("" : "+v"(i0v));
                static_for<M::MAXAT>([&](auto ROW_) __attribute__((always_inline)) { constexpr int ROW = ROW_;
                    constexpr int PR = (PH - ROW + NCOL) % NCOL;
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        const int i = i0v + RR;
                        static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                            col[PR][RR].ex[S][SRP] = ((i * M::NS) + S) * M::MAXAT + ROW;     // viterbi.c:515-522
                        });
                    });
                    // our copies of row i0-1 at these columns get the same edit
                    static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                        nbr[PR].ex[S][SRP] = (((i0v - 1) * M::NS) + S) * M::MAXAT + ROW;
                    });
                });
                cp_next_i += 1; cp_next_j += section_length;
            }
        }
    }

    // ---- the whole rectangle ------------------------------------------------------------------------------
    __device__ __forceinline__ void run(const DevJob &job, const DevSeqs &seqs, int *bnd, uint32_t *tb, int *ckpt) {
        Q = job.Q; T = job.T; q0 = job.q0; t0 = job.t0;
        tshift = job.tshift;
        tlast = seqs.tlen[job.pair] > 0 ? seqs.tlen[job.pair] - 1 : 0;
        seed_aux = 0;
        if constexpr (F::has_phase()) {
            tn4p = seqs.tn4 + seqs.toff[job.pair];
            if constexpr (CONT) {                     // bases in front of an intron that is open at the seam
                const int cis = job.first_cell[1];
                seed_aux = (cis >= 1 && cis - 1 <= tlast) ? (tn4p[cis - 1] & 0xff) : 0;
            }
        }
        if constexpr (SUB) { sub_cp = seqs.sub_colptr + 2 * job.sub_off; sub_rows = seqs.sub_rows; }
        if constexpr (SPAN == 1) span_in_p = seqs.span_in + job.span_off;
        if constexpr (SPAN == 2) span_out_p = seqs.span_out + job.span_off;
        first_state = job.first_state; final_state = CONT ? job.final_state : M::END;
        first_cell = job.first_cell;
        min_intron = kp->min_intron; max_intron = kp->max_intron;
        intron_span = (unsigned)(max_intron - min_intron);
        start_scope = CONT ? SCOPE_CORNER : kp->start_scope;
        end_scope = CONT ? SCOPE_CORNER : kp->end_scope;
        qc = seqs.qcode + seqs.qoff[job.pair];
        tc = seqs.tcode + seqs.toff[job.pair];
        if constexpr (F::has_splice()) {
            const int *base = seqs.ss + seqs.toff[job.pair];
            ss0 = base; ss1 = base + seqs.ss_stride; ss2 = base + 2 * seqs.ss_stride; ss3 = base + 3 * seqs.ss_stride;
        }
        best = LOW; best_i = best_j = best_qs = best_ts = 0; best_set = false;
        corner_set = false;
        ring_in = nullptr; ring_out = nullptr; use_ring_in = false; use_ring_out = false;
        const int section_length = (MODE == MODE_CKPT) ? T / (job.cp_count + 1) : 1;
        const int nstrips = (Q + 1 + W - 1) / W;
        const long long strip_tb = (long long)(T + 64) * 64 * R;
        for (int b = 0; b < nstrips; b++) {
            const int i0 = b * W + lane * R;
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                const int qpos = q0 + i0 + RR - 1;            // residue consumed by an advance_query=1 move into row i
                qcode[RR] = (i0 + RR >= 1 && i0 + RR <= Q) ? qc[qpos] : 0;
            });
            // registers start empty (row -1 does not exist; validity masks keep it unread)
            static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                expo.sc[S] = LOW;
                static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; expo.ex[S][E] = 0; });
                static_for<NCOL>([&](auto D_) __attribute__((always_inline)) { constexpr int D = D_;
                    nbr[D].sc[S] = LOW;
                    static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; nbr[D].ex[S][E] = 0; });
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        col[D][RR].sc[S] = LOW;
                        static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; col[D][RR].ex[S][E] = 0; });
                    });
                });
            });
            if constexpr (!CONT) strip_begin();
            cp_next_j = section_length > 0 ? section_length : 0x7fffffff; cp_next_i = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const bool first = (b == 0), last = (b == nstrips - 1);
            // slab layout: [empty column][carry row A (T+1 columns)][carry row B]
            carry_cols = carry_ok & !first;
            const int *bnd_in = first ? bnd : bnd + BND + (carry_ok ? (long long)((b + 1) & 1) * (T + 1) * BND : 0);
            int *bnd_out = bnd + BND + (carry_ok ? (long long)(b & 1) * (T + 1) * BND : 0);
            const int nsteps = T + 64;
            const int main_lo = 63 + M::MAXAT, main_hi = T;          // steps where every lane is interior in j
            // steps run in groups of NCOL with compile-time ring phases (s % NCOL); the padding steps past
            // nsteps have every lane outside the rectangle and do nothing
            const int nsteps_r = (nsteps + NCOL - 1) / NCOL * NCOL;
            const int main_lo_r = (main_lo + NCOL - 1) / NCOL * NCOL;
            auto group = [&](auto JI_, int s0) __attribute__((always_inline)) {
                constexpr bool JI = decltype(JI_)::value != 0;
                static_for<NCOL>([&](auto P_) __attribute__((always_inline)) { constexpr int P = P_;
                    step<JI, P>(s0 + P, i0, first, last, bnd_in, bnd_out, tb, b * strip_tb, ckpt, section_length,
                                job.cp_count);
                });
            };
            prefetch_column(0 - lane);
            prefetch_carry(0, bnd_in);
            int s = 0;
            for (; s < main_lo_r && s < nsteps_r; s += NCOL) group(IC<0>{}, s);
            for (; s + NCOL - 1 <= main_hi; s += NCOL) group(IC<1>{}, s);
            for (; s < nsteps_r; s += NCOL) group(IC<0>{}, s);
            if constexpr (!CONT) strip_end();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // carry row / traceback visible to the next strip
        }
    }

    // ---- the whole rectangle on NW cooperating waves (FIND_SCORE / FIND_REGION) ----------------------------
    // Wave w of the workgroup owns strip w of each "super-strip" of NW*64*R query rows and runs 2 chunks
    // of CH steps behind wave w-1; bottom rows go from wave to wave through LDS rings (no HBM carry row
    // unless Q+1 exceeds one super-strip), all waves meet at a barrier after every chunk.
    // Wave w at chunk c' reads columns [CH*c', CH*c'+CH] of the row above; wave w-1 has then finished
    // chunk c'+1, i.e. columns up to CH*(c'+2)-64: enough for CH >= 64.  Ring span <= 3*CH-64 < RING.
    static constexpr int CH = (64 + NCOL - 1) / NCOL * NCOL;
    template <int NW, bool STG_ = false>
    __device__ __forceinline__ void run_mw(const DevJob &job, const DevSeqs &seqs, int *bnd, lds_int *rings,
                                           int wid, lds_int *stage) {
        // STG_ (the kernel decides): not the blocking kernels -- the blocked-row entries of a column travel with their per-step
        // loads -- and not where the rings of an unpacked region pass on eight waves leave no room in the CU's 160 KB
        constexpr bool STG = STG_;
        static_assert(STG_CH == CH, "the stage is sized for run_mw's chunk");
        static_assert(!CONT && (MODE == MODE_SCORE || MODE == MODE_REGION), "multi-wave: full-rectangle passes");
        Q = job.Q; T = job.T; q0 = job.q0; t0 = job.t0;
        tshift = job.tshift;
        tlast = seqs.tlen[job.pair] > 0 ? seqs.tlen[job.pair] - 1 : 0;
        seed_aux = 0;
        if constexpr (F::has_phase()) {
            tn4p = seqs.tn4 + seqs.toff[job.pair];
            if constexpr (CONT) {                     // bases in front of an intron that is open at the seam
                const int cis = job.first_cell[1];
                seed_aux = (cis >= 1 && cis - 1 <= tlast) ? (tn4p[cis - 1] & 0xff) : 0;
            }
        }
        if constexpr (SUB) { sub_cp = seqs.sub_colptr + 2 * job.sub_off; sub_rows = seqs.sub_rows; }
        first_state = job.first_state; final_state = (SEED == 2) ? job.final_state : M::END;
        first_cell = job.first_cell;
        seeded = false; seed_rows = job.seed_rows; seed_kshift = job.seed_kshift;
        if constexpr (SEED == 1) seed_wr = seqs.seed + job.seed_off;
        if constexpr (SEED == 2) { seeded = job.seed_off >= 0; seed_rd = seqs.seed + (seeded ? job.seed_off : 0); }
        min_intron = kp->min_intron; max_intron = kp->max_intron;
        intron_span = (unsigned)(max_intron - min_intron);
        start_scope = kp->start_scope; end_scope = kp->end_scope;
        qc = seqs.qcode + seqs.qoff[job.pair];
        tc = seqs.tcode + seqs.toff[job.pair];
        if constexpr (F::has_splice()) {
            const int *base = seqs.ss + seqs.toff[job.pair];
            ss0 = base; ss1 = base + seqs.ss_stride; ss2 = base + 2 * seqs.ss_stride; ss3 = base + 3 * seqs.ss_stride;
        }
        best = LOW; best_i = best_j = best_qs = best_ts = 0; best_set = false;
        corner_set = false;
        const int nstrips = (Q + 1 + W - 1) / W;
        const int nsuper = (nstrips + NW - 1) / NW;
        const int nsteps = T + 64;
        const int nchunks = (nsteps + CH - 1) / CH;
        const int main_lo = 63 + M::MAXAT, main_hi = T;
        for (int sb = 0; sb < nsuper; sb++) {
            const int b = sb * NW + wid;
            const int i0 = b * W + lane * R;
            static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                const int qpos = q0 + i0 + RR - 1;
                qcode[RR] = (i0 + RR >= 1 && i0 + RR <= Q) ? qc[qpos] : 0;
            });
            static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
                expo.sc[S] = LOW;
                static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; expo.ex[S][E] = 0; });
                static_for<NCOL>([&](auto D_) __attribute__((always_inline)) { constexpr int D = D_;
                    nbr[D].sc[S] = LOW;
                    static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; nbr[D].ex[S][E] = 0; });
                    static_for<R>([&](auto RR_) __attribute__((always_inline)) { constexpr int RR = RR_;
                        col[D][RR].sc[S] = LOW;
                        static_for<XS>([&](auto E_) __attribute__((always_inline)) { constexpr int E = E_; col[D][RR].ex[S][E] = 0; });
                    });
                });
            });
            strip_begin();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            carry_cols = carry_ok & (sb > 0);
            const int *bnd_in = (sb == 0) ? bnd : bnd + BND + (carry_ok ? (long long)((sb + 1) & 1) * (T + 1) * BND : 0);
            int *bnd_out = bnd + BND + (carry_ok ? (long long)(sb & 1) * (T + 1) * BND : 0);
            use_ring_in = wid > 0;  use_ring_out = wid < NW - 1;
            ring_in = rings + (wid > 0 ? wid - 1 : 0) * RING * BND;
            ring_out = rings + (wid < NW - 1 ? wid : 0) * RING * BND;
            const bool first = (b == 0);                       // no row above at all
            const bool last = (b >= nstrips - 1);              // nobody below needs our bottom row
            const bool idle = (b >= nstrips);                  // strip entirely below the rectangle
            auto group = [&](auto JI_, int s0) __attribute__((always_inline)) {
                constexpr bool JI = decltype(JI_)::value != 0;
                static_for<NCOL>([&](auto P_) __attribute__((always_inline)) { constexpr int P = P_;
                    step<JI, P, STG>(s0 + P, i0, first, last, bnd_in, bnd_out, nullptr, 0, nullptr, 1, 0);
                });
            };
            // the columns chunk k's steps read ahead: k CH + 1 ... k CH + CH (the lanes' columns of earlier chunks are in the ring)
            auto refill = [&](int k) __attribute__((always_inline)) {
                if constexpr (STG) {
                    fill_stage(stage, k * CH + 1, 64);
                    if constexpr (CH > 64) fill_stage(stage, k * CH + 65, CH - 64);
                }
            };
            // every wave executes exactly nticks barriers: 2*wid before its first chunk, one after each
            // of its nchunks chunks, 2*(NW-1-wid) after its last
            for (int t = 0; t < 2 * wid; t++) __syncthreads();
            if (idle) {
                for (int k = 0; k < nchunks; k++) __syncthreads();
            } else {
                if constexpr (STG) {
                    stage_base = (int)(unsigned)(size_t)stage;
                    stage_a = stage_base + ((0 - lane) & (STG_COLS - 1)) * 4;
                    fill_stage(stage, -63, 64);          // columns -63 ... 0: what the first steps of the lanes read
                    prefetch_staged();
                } else {
                    prefetch_column(0 - lane);
                }
                prefetch_carry(0, bnd_in);
                int k = 0;
                for (; k < nchunks && k * CH < main_lo; k++) {
                    refill(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<0>{}, s);
                    __syncthreads();
                }
                for (; k < nchunks && k * CH + CH - 1 <= main_hi; k++) {
                    refill(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<1>{}, s);
                    __syncthreads();
                }
                for (; k < nchunks; k++) {
                    refill(k);
                    for (int s = k * CH; s < k * CH + CH; s += NCOL) group(IC<0>{}, s);
                    __syncthreads();
                }
            }
            for (int t = 0; t < 2 * (NW - 1 - wid); t++) __syncthreads();
            strip_end();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
        }
    }

    // ---- epilogue: wave-wide end cell, traceback walk, checkpoint traceback -----------------------------
    // Row-major-first maximum (viterbi.c:778-791): highest score, then smallest j, then smallest i.
    __device__ __forceinline__ void reduce_best() {
        for (int off = 32; off > 0; off >>= 1) {
            const int o_best = __shfl_xor(best, off), o_i = __shfl_xor(best_i, off), o_j = __shfl_xor(best_j, off);
            const int o_qs = __shfl_xor(best_qs, off), o_ts = __shfl_xor(best_ts, off);
            const bool o_set = __shfl_xor((int)best_set, off) != 0;
            const int b = best, bi = best_i, bj = best_j, bqs = best_qs, bts = best_ts;
            const bool bs = best_set;
            const bool take = o_set & (!bs | (o_best > b) | ((o_best == b) & ((o_j < bj) | ((o_j == bj) & (o_i < bi)))));
            best = take ? o_best : b;  best_i = take ? o_i : bi;  best_j = take ? o_j : bj;
            best_qs = take ? o_qs : bqs;  best_ts = take ? o_ts : bts;
            best_set = bs | o_set;
        }
    }

    // strict-greater per strip is row-major-first only inside a strip: merge strips with the full order
    int sbest, sbest_i, sbest_j, sbest_qs, sbest_ts;
    bool sbest_set;
    __device__ __forceinline__ void strip_begin() {
        sbest = best; sbest_i = best_i; sbest_j = best_j; sbest_qs = best_qs; sbest_ts = best_ts; sbest_set = best_set;
        best_set = false; best = LOW;
    }
    __device__ __forceinline__ void strip_end() {
        const int b = best, bi = best_i, bj = best_j, bqs = best_qs, bts = best_ts;
        const int ob = sbest, oi = sbest_i, oj = sbest_j, oqs = sbest_qs, ots = sbest_ts;
        const bool bs = best_set, os = sbest_set;
        const bool keep_old = os & (!bs | (ob > b) | ((ob == b) & ((oj < bj) | ((oj == bj) & (oi < bi)))));
        best = keep_old ? ob : b;  best_i = keep_old ? oi : bi;  best_j = keep_old ? oj : bj;
        best_qs = keep_old ? oqs : bqs;  best_ts = keep_old ? ots : bts;
        best_set = bs | os;
    }

    // traceback word of cell (i, j) in the step-major slab
    __device__ static uint32_t tb_at(const uint32_t *tb, int i, int j, int T) {
        const int b = i / W, l = (i - b * W) / R, r = (i - b * W) - l * R;
        const long long strip_tb = (long long)(T + 64) * 64 * R;
        return tb[b * strip_tb + ((long long)(j + l) * 64 + l) * R + r];
    }
    __device__ __forceinline__ static int tb_transition(uint32_t word, int state) {      // -1: never assigned (NULL)
        int code = 0;
        static_for<M::NS>([&](auto S_) __attribute__((always_inline)) { constexpr int S = S_;
            if constexpr (F::bits(S) > 0)
                if (state == S) code = (word >> F::shift(S)) & ((1u << F::bits(S)) - 1u);
        });
        int tr = -1;
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; if (M::tr[K].out == state && F::code(K) == code) tr = K; });
        return tr;
    }
    __device__ __forceinline__ static void tr_info(int k, int &in, int &out, int &aq, int &at) {
        in = out = aq = at = 0;
        static_for<M::NT>([&](auto K_) __attribute__((always_inline)) { constexpr int K = K_; if (k == K) { in = M::tr[K].in; out = M::tr[K].out; aq = M::tr[K].aq; at = M::tr[K].at; } });
    }

    // Viterbi_Data_create_Alignment's walk (viterbi.c:342-379), by lane 0.  The path is emitted END -> START
    // as runs (transition << 24 | length): the run-length merge of Alignment_add (alignment.c:75-102).
    __device__ __noinline__ static void walk(const uint32_t *tb, int T, int first_state, int final_state, int qe,
                                             int te, uint32_t *runs, int cap, DevResult &res) {
        int i = qe, j = te, n = 0, in, out, aq, at;
        int tr = tb_transition(tb_at(tb, i, j, T), final_state);
        bool overflow = false;
        int run_tr = -1, run_len = 0;
        auto emit = [&](int t) {
            if (t == run_tr && run_len < 0xffffff) { run_len++; return; }
            if (run_tr >= 0) { if (n < cap) runs[n] = ((uint32_t)run_tr << 24) | (uint32_t)run_len; else overflow = true; n++; }
            run_tr = t; run_len = 1;
        };
        while (tr >= 0) {
            emit(tr);
            tr_info(tr, in, out, aq, at);
            i -= aq; j -= at;
            tr = tb_transition(tb_at(tb, i, j, T), in);
            if (tr < 0) break;
            tr_info(tr, in, out, aq, at);
            if (in == M::START) {
                emit(tr);
                i -= aq; j -= at;
                break;
            }
            if (CONT && !(i | j) && out == first_state) break;
        }
        if (run_tr >= 0) { if (n < cap) runs[n] = ((uint32_t)run_tr << 24) | (uint32_t)run_len; else overflow = true; n++; }
        res.qs = i; res.ts = j; res.n_ops = n;
        if (overflow) res.flags |= FLAG_OPS_OVERFLOW;
    }

    // Viterbi_Checkpoint_traceback (viterbi.c:537-601), by lane 0; list is written last section first.
    __device__ __noinline__ static void checkpoint_traceback(const int *ckpt, const DevJob &job, DevVsa *vsa,
                                                             DevResult &res) {
        const int Q = job.Q, T = job.T, q0 = job.q0, t0 = job.t0;
        const int cpn = job.cp_count, section_length = T / (cpn + 1);
        auto decode = [&](int srp, int &state, int &row, int &pos) {
            row = srp % M::MAXAT;
            const int rem = srp / M::MAXAT;
            state = rem % M::NS;
            pos = rem / M::NS;
        };
        auto cell_at = [&](int cp, int row, int qpos, int state) {
            return ckpt + ((((long long)cp * M::MAXAT + row) * (Q + 1) + qpos) * M::NS + state) * CS;
        };
        int state, row, pos, n = 0;
        decode(res.last_srp, state, row, pos);
        int query_start = q0 + pos, target_start = t0 + section_length * cpn - row;
        DevVsa v;
        v.qs = query_start; v.ts = target_start;
        v.ql = (q0 + Q) - query_start; v.tl = (t0 + T) - target_start;
        v.first_state = state;
        for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = l < CS ? res.final_cell[l] : 0;
        vsa[n++] = v;
        for (int c = cpn - 1; c >= 1; c--) {
            const DevVsa p = v;
            const int prev_row = row;
            const int *cell = cell_at(c, prev_row, p.qs - q0, p.first_state);
            decode(cell[CS - 1], state, row, pos);
            query_start = q0 + pos;
            target_start = p.ts - section_length - row + prev_row;
            v.qs = query_start; v.ts = target_start; v.ql = p.qs - query_start; v.tl = p.ts - target_start;
            v.first_state = state;
            for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = l < CS ? cell[l] : 0;
            vsa[n++] = v;
        }
        {
            const DevVsa p = v;
            const int *cell = cell_at(0, row, p.qs - q0, p.first_state);
            v.qs = q0; v.ts = t0; v.ql = query_start - q0; v.tl = target_start - t0;
            v.first_state = job.first_state;
            for (int l = 0; l < CELL_MAX; l++) v.final_cell[l] = l < CS ? cell[l] : 0;
            vsa[n++] = v;
        }
        res.n_vsa = n;
    }
};


// -------------------------------------------------------------------------------------------------------------
// Kernel: persistent waves, one job at a time per wave.
// -------------------------------------------------------------------------------------------------------------
// WPE: waves per SIMD the register allocator must leave room for (1 = no cap: 512 unified registers)
template <class T> struct TypeTag { using type = T; };

// BYROOT (continuation passes of a model whose inner states fall into two components that only share START and END: est2genome's
// strands): a job that names the state its alignment's END is entered from (DevJob::root, handed down from the region pass through
// the checkpoint pass to every sub-alignment) runs the instantiation that computes that state's component only (WaveDP, COMP);
// a job that does not (root 0), or whose first state is not of that component, runs the whole model.
template <class M, int R, int MODE, bool CONT, bool LOCAL, bool PACK, int WPE, bool SUB = false, int SPAN = 0, bool BYROOT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void viterbi_kernel(const KParams *kparams, DevSeqs seqs, const DevJob *jobs,
                                                     int n_jobs, DevResult *results, DevVsa *vsas, uint8_t *ops,
                                                     DevScratch scratch, int *queue) {
    using DP = WaveDP<M, R, MODE, CONT, LOCAL, PACK, SUB, SPAN>;
    __shared__ KParams kp_lds;
    __shared__ int next_job;
    __shared__ long long run_off;
    __shared__ int run_n;
    {
        const int *src = reinterpret_cast<const int *>(kparams);
        int *dst = reinterpret_cast<int *>(&kp_lds);
        for (int x = threadIdx.x; x < (int)(sizeof(KParams) / sizeof(int)); x += 64) dst[x] = src[x];
    }
    __syncthreads();
    const int wave = blockIdx.x;
    int *bnd = scratch.bnd + (long long)wave * scratch.bnd_stride;
    if (threadIdx.x == 0) DP::write_empty_column(bnd);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    uint32_t *tb = scratch.tb ? scratch.tb + (long long)wave * scratch.tb_stride : nullptr;
    int *ckpt_slab = scratch.ckpt ? scratch.ckpt + (long long)wave * scratch.ckpt_stride : nullptr;
    for (;;) {
        if (threadIdx.x == 0) next_job = atomicAdd(queue, 1);
        __syncthreads();
        const int jid = __builtin_amdgcn_readfirstlane(next_job);      // wave-uniform: job descriptions and pointers in scalar registers
        __syncthreads();
        if (jid >= n_jobs) break;
        const DevJob &job = jobs[jid];
        if constexpr (MODE == MODE_PATH && CONT && !SUB && SPAN == 0) {
            // A sub-alignment between two checkpoints of ONE query row that starts and ends in a state whose only way back to
            // itself that costs nothing is its own loop (an intron across a whole section: about half of the sections of a
            // chance alignment across a 100 kb window): the reference's continuation (viterbi.c:705-714) can only find the loop,
            // T times -- every detour (3' site, gaps, 5' site back in) scores strictly less at every cell, so neither order nor
            // ties matter -- and leaves the first cell's slots as they are.  One run, no DP (1 700 steps of a wave with one
            // live lane otherwise).  KParams::loop_tr holds the proof's result per state; -1: run it.
            const int fs = job.first_state;
            const int ltr = (fs >= 0 && fs < 16) ? kp_lds.loop_tr[fs] : -1;
            if (job.Q == 0 && job.T > 0 && job.T < (1 << 24) && fs == job.final_state && ltr >= 0) {
                if (threadIdx.x == 0) {
                    DevResult res;
                    res.flags = 0; res.n_ops = 1; res.n_vsa = 0; res.last_srp = 0; res.qs = res.ts = 0; res.pad = 0;
                    res.cell_size = DP::CS;
                    for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = l < DP::CS ? job.first_cell[l] : 0;
                    res.end_set = true; res.score = res.final_cell[0]; res.qe = 0; res.te = job.T;
                    res.ops_off = (long long)atomicAdd(scratch.runs_used, 1ull);
                    if (res.ops_off + 1 > scratch.runs_capacity) { res.flags |= FLAG_OPS_OVERFLOW; res.n_ops = 0; }
                    else scratch.runs_out[res.ops_off] = ((uint32_t)ltr << 24) | (uint32_t)job.T;
                    results[jid] = res;
                }
                continue;
            }
        }
        // every member starts defined.  Round 4: one set of the derived protein2genome vectors came out differently when the
        // suite's kernel-variant tests had run before it in the same process (and only then): an object left uninitialised is
        // undefined wherever a member is read before it is written, and what the registers held decided
        // (tests/test_gpu_parity.py after tests/test_gpu_kernel_variants.py; -ftrivial-auto-var-init=zero and =pattern
        // both gave the reference's answer, as does this)
      auto do_job = [&](auto tag_) __attribute__((always_inline)) {
        using DP = typename decltype(tag_)::type;
        DP dp{};
        dp.kp = &kp_lds;
        dp.lane = threadIdx.x;
        dp.carry_ok = scratch.carry != 0;
        int *ckpt = ckpt_slab;
        if constexpr (MODE == MODE_CKPT)
            if (job.ckpt_off >= 0) ckpt = scratch.ckpt_dump + job.ckpt_off;
        dp.run(job, seqs, bnd, tb, ckpt);
        DevResult res;
        res.flags = 0; res.n_ops = 0; res.n_vsa = 0; res.last_srp = 0; res.qs = res.ts = 0; res.pad = 0;
        res.cell_size = DP::CS;
        for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
        if constexpr (CONT) {
            // the lane that owns (Q, T) holds the corner cell of the final state
            const unsigned long long owners = __ballot(dp.corner_set);
            const int owner = owners ? __ffsll((long long)owners) - 1 : 0;
            for (int l = 0; l < DP::CS; l++) res.final_cell[l] = __shfl(dp.corner[l], owner);
            res.end_set = owners != 0;
            res.score = res.final_cell[0];
            res.qe = job.Q; res.te = job.T;
            if constexpr (MODE == MODE_CKPT) res.last_srp = res.final_cell[DP::CS - 1];
        } else {
            dp.reduce_best();
            res.score = dp.best; res.end_set = dp.best_set;
            res.qe = dp.best_i; res.te = dp.best_j;
            if constexpr (MODE == MODE_REGION) {
                if constexpr (PACK) { res.qs = dp.best_qs >> job.tshift; res.ts = dp.best_qs & ((1 << job.tshift) - 1); }
                else { res.qs = dp.best_qs; res.ts = dp.best_ts; }
            }
        }
        if (!res.end_set) res.flags |= FLAG_NO_END;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) {
            if constexpr (MODE == MODE_PATH) {
                uint32_t *runs = scratch.runs + (long long)wave * scratch.runs_stride;
                res.ops_off = 0;
                if (res.end_set) DP::walk(tb, job.T, job.first_state, CONT ? job.final_state : M::END, res.qe, res.te,
                                          runs, (int)scratch.runs_stride, res);
                res.ops_off = (long long)atomicAdd(scratch.runs_used, (unsigned long long)res.n_ops);
                if (res.ops_off + res.n_ops > scratch.runs_capacity) { res.flags |= FLAG_OPS_OVERFLOW; res.n_ops = 0; }
                run_off = res.ops_off; run_n = res.n_ops;
            }
            if constexpr (MODE == MODE_CKPT) {
                if (res.end_set) DP::checkpoint_traceback(ckpt, job, vsas + job.vsa_off, res);
            }
            results[jid] = res;
        }
        __syncthreads();
        if constexpr (MODE == MODE_PATH) {          // all lanes copy the runs to their compact place
            const uint32_t *runs = scratch.runs + (long long)wave * scratch.runs_stride;
            for (int x = threadIdx.x; x < run_n; x += 64) scratch.runs_out[run_off + x] = runs[x];
            __syncthreads();
        }
      };
        using RTk = Roots<M>;
        if constexpr (BYROOT && CONT && RTk::count() == 2 && RTk::disjoint()) {
            using DP1 = WaveDP<M, R, MODE, CONT, LOCAL, PACK, SUB, SPAN, 0, 1>;
            using DP2 = WaveDP<M, R, MODE, CONT, LOCAL, PACK, SUB, SPAN, 0, 2>;
            const int root = job.root, fs = job.first_state;            // wave-uniform
            constexpr unsigned m1 = DP1::alive_mask(), m2 = DP2::alive_mask();     // (START and END are in both)
            const int ls = job.final_state;
            const bool in1 = root == RTk::root(0) && ((m1 >> fs) & 1u) && ((m1 >> ls) & 1u);
            const bool in2 = root == RTk::root(1) && ((m2 >> fs) & 1u) && ((m2 >> ls) & 1u);
            if (in1) do_job(TypeTag<DP1>{});
            else if (in2) do_job(TypeTag<DP2>{});
            else do_job(TypeTag<DP>{});
        } else {
            do_job(TypeTag<DP>{});
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// Kernel: NW cooperating waves per job (FIND_SCORE / FIND_REGION over whole rectangles).
// -------------------------------------------------------------------------------------------------------------
template <class M, int R, int MODE, bool LOCAL, bool PACK, int NW, int WPE, bool SUB = false, int SEED = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
void viterbi_kernel_mw(const KParams *kparams, DevSeqs seqs, const DevJob *jobs, int n_jobs, DevResult *results,
                       DevScratch scratch, int *queue) {
    using DP = WaveDP<M, R, MODE, false, LOCAL, PACK, SUB, 0, SEED>;
    __shared__ int corner_lds[3];
    __shared__ DevJob job_lds;           // SEED 2: the window being run (rewritten between the hops of one job)
    __shared__ int hop_more;
    __shared__ KParams kp_lds;
    __shared__ int next_job;
    __shared__ int rings[(NW > 1 ? NW - 1 : 1) * DP::RING * DP::BND];
    __shared__ int wave_best[NW][8];
    // the column stages (WaveDP::fill_stage): a plane is STG_COLS ints, aligned to its own size for the running address
    constexpr bool STG = !SUB && (sizeof(int) * ((NW > 1 ? NW - 1 : 1) * DP::RING * DP::BND + NW * DP::STG_INTS) + sizeof(KParams) + sizeof(DevJob) + 1024 <= 160 * 1024);
    __shared__ __attribute__((aligned(1024))) int stage_mem[STG ? NW * DP::STG_INTS : 1];
    {
        const int *src = reinterpret_cast<const int *>(kparams);
        int *dst = reinterpret_cast<int *>(&kp_lds);
        for (int x = threadIdx.x; x < (int)(sizeof(KParams) / sizeof(int)); x += 64 * NW) dst[x] = src[x];
    }
    __syncthreads();
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int *bnd = scratch.bnd + (long long)blockIdx.x * scratch.bnd_stride;
    if (threadIdx.x == 0) DP::write_empty_column(bnd);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) next_job = atomicAdd(queue, 1);
        __syncthreads();
        const int jid = __builtin_amdgcn_readfirstlane(next_job);      // wave-uniform: job descriptions and pointers in scalar registers
        __syncthreads();
        if (jid >= n_jobs) break;
        if constexpr (SEED == 2) {
            if (threadIdx.x == 0) job_lds = jobs[jid];
            __syncthreads();
        }
        const DevJob &job = (SEED == 2) ? job_lds : jobs[jid];
        int hop = 0, first_score = 0;
      next_hop:
        DP dp{};                 // every member starts defined (see viterbi_kernel)
        dp.kp = &kp_lds;
        dp.lane = threadIdx.x & 63;
        dp.carry_ok = scratch.carry != 0;
        if constexpr (SEED == 2) {
            if (threadIdx.x == 0) { corner_lds[2] = 0; hop_more = 0; }
            __syncthreads();
        }
        dp.template run_mw<NW, STG>(job, seqs, bnd, (typename DP::lds_int *)rings, wid, (typename DP::lds_int *)stage_mem + (STG ? wid * DP::STG_INTS : 0));
        if constexpr (SEED == 2) {
            if (dp.corner_set) { corner_lds[0] = dp.corner[0]; corner_lds[1] = dp.corner[1]; corner_lds[2] = 1; }
        }
        dp.reduce_best();
        if (dp.lane == 0) {
            wave_best[wid][0] = dp.best; wave_best[wid][1] = dp.best_i; wave_best[wid][2] = dp.best_j;
            wave_best[wid][3] = dp.best_qs; wave_best[wid][4] = dp.best_ts; wave_best[wid][5] = dp.best_set;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int b = LOW, bi = 0, bj = 0, bqs = 0, bts = 0; bool bs = false;
            for (int w = 0; w < NW; w++) {           // row-major-first merge (viterbi.c:778-791)
                const int ob = wave_best[w][0], oi = wave_best[w][1], oj = wave_best[w][2];
                const bool os = wave_best[w][5] != 0;
                const bool take = os && (!bs || ob > b || (ob == b && (oj < bj || (oj == bj && oi < bi))));
                if (take) { b = ob; bi = oi; bj = oj; bqs = wave_best[w][3]; bts = wave_best[w][4]; }
                bs = bs || os;
            }
            DevResult res;
            res.flags = bs ? 0 : FLAG_NO_END; res.n_ops = 0; res.n_vsa = 0; res.last_srp = 0; res.pad = 0;
            res.cell_size = DP::CS; res.ops_off = 0;
            for (int l = 0; l < CELL_MAX; l++) res.final_cell[l] = 0;
            res.score = b; res.end_set = bs; res.qe = bi; res.te = bj; res.qs = 0; res.ts = 0;
            if constexpr (MODE == MODE_REGION) {
                if constexpr (PACK) { res.qs = bqs >> job.tshift; res.ts = bqs & ((1 << job.tshift) - 1); }
                else { res.qs = bqs; res.ts = bts; }
            }
            if constexpr (SEED == 2) {          // the window's corner cell: score, raw payload (start or entry-cell identity)
                const int payload = corner_lds[1];
                if (hop == 0) first_score = corner_lds[0];
                res.end_set = corner_lds[2]; res.score = first_score; res.pad = payload;
                res.qe = job.Q; res.te = job.T;
                res.flags = corner_lds[2] ? 0 : FLAG_NO_END;
                res.n_vsa = hop + 1;                                       // windows this job ran
                if (corner_lds[2] && payload >= 0) {                       // a real region start (window coordinates)
                    res.qs = payload >> job.tshift;
                    res.ts = (payload & ((1 << job.tshift) - 1)) + job.win_t0w;
                } else if (corner_lds[2] && job.win_d >= 1 && hop + 1 < job.win_hops) {
                    // entered through the dump: the identity of the cell; the next window ends in that cell and state
                    constexpr int DC = DP::DC;
                    const int v = -payload - 1, jc = v % DC, rest = v / DC;
                    const int d2 = job.win_d - 1, t0w2 = d2 >= 1 ? (d2 << job.seed_kshift) - (DC - 1) : 0;
                    const int endcol = job.win_t0w + jc;
                    job_lds.Q = rest / M::NS; job_lds.final_state = rest % M::NS;
                    job_lds.T = endcol - t0w2; job_lds.t0 = job.win_t0_base + t0w2;
                    job_lds.seed_off = d2 >= 1 ? job.seed_base + (long long)(d2 - 1) * DC * job.seed_rows * DP::SEEDW : -1;
                    int tb = 0;
                    while ((1LL << tb) <= job_lds.T) tb++;
                    job_lds.tshift = tb; job_lds.win_d = d2; job_lds.win_t0w = t0w2;
                    hop_more = 1;
                }
            }
            if (!(SEED == 2 && hop_more)) results[jid] = res;
        }
        __syncthreads();
        if constexpr (SEED == 2) {
            if (hop_more) { hop++; goto next_hop; }
        }
    }
}

}  // namespace c4k
