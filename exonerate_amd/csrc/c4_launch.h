// c4_launch.h — host-visible table of the compiled Viterbi kernels (the device-side counterpart of the
// reference's name-keyed Bootstrapper_lookup table, src/model/bootstrapper.c:85-90).
#pragma once
#include <hip/hip_runtime.h>
#include "c4_viterbi_kernel.h"

namespace c4k {

enum Family { FAM_UNGAPPED = 0, FAM_AFFINE, FAM_EST2GENOME, FAM_UNGAPPED_P2D, FAM_PROTEIN2DNA, FAM_PROTEIN2GENOME,
              // BSDP's derived (Segment) models of those: start terminal, end terminal, join per match state
              FAM_AFFINE_START, FAM_AFFINE_END, FAM_AFFINE_JOIN,
              FAM_EST2GENOME_FWD_START, FAM_EST2GENOME_FWD_END, FAM_EST2GENOME_FWD_JOIN,
              FAM_EST2GENOME_REV_START, FAM_EST2GENOME_REV_END, FAM_EST2GENOME_REV_JOIN,
              FAM_PROTEIN2DNA_START, FAM_PROTEIN2DNA_END, FAM_PROTEIN2DNA_JOIN,
              FAM_PROTEIN2GENOME_START, FAM_PROTEIN2GENOME_END, FAM_PROTEIN2GENOME_JOIN,
              // est2genome's span models (heuristic.c:461-472): match state -> intron state, intron state -> match state
              FAM_EST2GENOME_FWD_SPAN_SRC, FAM_EST2GENOME_FWD_SPAN_DST, FAM_EST2GENOME_REV_SPAN_SRC, FAM_EST2GENOME_REV_SPAN_DST,
              // protein2genome's: one span per intron phase (span states 10, 11, 12)
              FAM_PROTEIN2GENOME_PHASE0_SPAN_SRC, FAM_PROTEIN2GENOME_PHASE0_SPAN_DST,
              FAM_PROTEIN2GENOME_PHASE1_SPAN_SRC, FAM_PROTEIN2GENOME_PHASE1_SPAN_DST,
              FAM_PROTEIN2GENOME_PHASE2_SPAN_SRC, FAM_PROTEIN2GENOME_PHASE2_SPAN_DST,
              FAM_COUNT };

struct LaunchArgs {
    const KParams *kp;
    DevSeqs seqs;
    const DevJob *jobs;
    int n_jobs;
    DevResult *results;
    DevVsa *vsas;
    uint8_t *ops;
    DevScratch scratch;
    int *queue;
    int grid;
    hipStream_t stream;
    // the two-jobs-per-lane kernels that take their pairing from the host (c4_win16_kernel.h, c4_ckpt16_kernel.h): n_aux pairs of
    // job indices (second = -1: the job runs alone); the two jobs of a pair have the same root (Roots, c4_viterbi16_kernel.h)
    const int *aux = nullptr;
    int n_aux = 0;
};

struct KernelInfo {
    hipError_t (*launch)(const LaunchArgs &);
    const void *func;         // for occupancy queries
    const char *name;
    int R;                    // query rows per lane
    int cs;                   // ints per state cell (1 + extra slots)
    int bnd;                  // ints per column of the strip carry row
    int n_states, max_at;
    int waves;                // waves per job (workgroup = 64 * waves threads)
    int seedw;                // ints per row of a dumped column (SEED kernels)
    int ckw;                  // the packed checkpoint pass: ints per row of a checkpoint column in a job's slab (root -1; see ckw_root)
    int pairs = 0;            // 1: the kernel reads LaunchArgs::aux (pairs of jobs with a common root, DevJob::pad0)
    int ckw_root = 0;         // the packed checkpoint pass restricted to one root's component: ints per row
    int hbm_carry = 0;        // 1: cooperating waves that hand the strip carry rows on through the workgroup's slab (not LDS rings)
};

// family x mode x continuation x local-scope specialisation; NULL launch = not compiled
// sub: the variant with sub-optimal blocking (DevSeqs::sub_colptr / sub_rows must be set)
// span: 0, or BSDP's span seam (1 = start cells read from a matrix, 2 = END cells copied out to one)
const KernelInfo *get_kernel(int family, int mode, bool cont, bool local, bool pack, int wpe = 0, bool sub = false,
                             int span = 0);
// multi-wave kernels (`waves` = 4 or 8 cooperating waves per job) for FIND_SCORE / FIND_REGION without
// continuation
// seed: 0, or the two halves of the windowed region pass (1 = score pass that dumps columns, 2 = region pass that
// starts from a dump and reports its corner cell); local, 4 waves only
const KernelInfo *get_kernel_mw(int family, int mode, bool local, bool pack, int waves = 4, bool sub = false, int seed = 0);

// the packed 16-bit score pass with column dumps (c4_viterbi16_kernel.h): two jobs per lane; NULL = not compiled for the family.
// Launched over the same job / result arrays as the 32-bit kernel (workgroup p runs jobs 2p and 2p + 1).
const KernelInfo *get_kernel_pk16(int family, int variant = 0);
// the packed 16-bit checkpoint pass (c4_ckpt16_kernel.h): two jobs per lane, one wave per pair of jobs (LaunchArgs::aux);
// scratch.ckpt holds two slabs of ckpt_stride ints per wave; rooted: the form that computes the component of DevJob::root only
// (NULL where the family's components overlap); variant: rows per lane / register cap shapes kept for measurement (0 = the default)
const KernelInfo *get_kernel_ck16(int family, int variant = 0, bool rooted = false);
// the packed 16-bit region windows (c4_win16_kernel.h): two windows per lane, one wave per pair of window chains, started from
// the 16-bit dumps of get_kernel_pk16(family, 3); variant: rows per lane / register cap shapes (0 = the default)
const KernelInfo *get_kernel_win16(int family, int variant = 0);
// the staged form of the packed score pass (get_kernel_pk16(family, 4)): residue codes its query profile holds, query rows
// (Q + 1) a workgroup covers
int pk16_staged_codes();
int pk16_staged_rows();
int pk16_staged_rows6();            // ... of the six-rows-per-lane form (get_kernel_pk16 variant 7)
hipError_t pk16_build_splice(int family, const KParams *kp, const int *ss, long long ss_stride, long long n, void *out, hipStream_t s);

#define C4K_DEFINE_KERNEL_SPAN(SYMBOL, M, RVAL, MODE, CONT, LOCAL, PACK, WPE, SUBV, SPANV)                                          \
    static hipError_t SYMBOL##_launch(const LaunchArgs &a) {                                               \
        hipLaunchKernelGGL((viterbi_kernel<M, RVAL, MODE, CONT, LOCAL, PACK, WPE, SUBV, SPANV>), dim3(a.grid), dim3(64), 0,        \
                           a.stream, a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.vsas, a.ops, a.scratch,  \
                           a.queue);                                                                       \
        return hipGetLastError();                                                                          \
    }                                                                                                      \
    const KernelInfo *SYMBOL() {                                                                           \
        static const KernelInfo ki = {SYMBOL##_launch,                                                     \
                                      (const void *)viterbi_kernel<M, RVAL, MODE, CONT, LOCAL, PACK, WPE, SUBV, SPANV>,            \
                                      #SYMBOL,                                                             \
                                      RVAL,                                                                \
                                      WaveDP<M, RVAL, MODE, CONT, LOCAL, PACK>::CS,                              \
                                      WaveDP<M, RVAL, MODE, CONT, LOCAL, PACK>::BND,                             \
                                      M::NS,                                                               \
                                      M::MAXAT,                                                            \
                                      1, 0};                                                        \
        return &ki;                                                                                        \
    }

#define C4K_DEFINE_KERNEL(SYMBOL, M, RVAL, MODE, CONT, LOCAL, PACK, WPE, SUBV) \
    C4K_DEFINE_KERNEL_SPAN(SYMBOL, M, RVAL, MODE, CONT, LOCAL, PACK, WPE, SUBV, 0)

#define C4K_DEFINE_KERNEL_MW(SYMBOL, M, RVAL, MODE, LOCAL, PACK, NWV, WPE, SUBV) \
    C4K_DEFINE_KERNEL_MW_SEED(SYMBOL, M, RVAL, MODE, LOCAL, PACK, NWV, WPE, SUBV, 0)

#define C4K_DEFINE_KERNEL_MW_SEED(SYMBOL, M, RVAL, MODE, LOCAL, PACK, NWV, WPE, SUBV, SEEDV)                     \
    static hipError_t SYMBOL##_launch(const LaunchArgs &a) {                                               \
        hipLaunchKernelGGL((viterbi_kernel_mw<M, RVAL, MODE, LOCAL, PACK, NWV, WPE, SUBV, SEEDV>), dim3(a.grid),   \
                           dim3(64 * NWV), 0, a.stream, a.kp, a.seqs, a.jobs, a.n_jobs, a.results,          \
                           a.scratch, a.queue);                                                            \
        return hipGetLastError();                                                                          \
    }                                                                                                      \
    const KernelInfo *SYMBOL() {                                                                           \
        static const KernelInfo ki = {SYMBOL##_launch,                                                     \
                                      (const void *)viterbi_kernel_mw<M, RVAL, MODE, LOCAL, PACK, NWV, WPE, SUBV, SEEDV>, \
                                      #SYMBOL,                                                             \
                                      RVAL,                                                                \
                                      WaveDP<M, RVAL, MODE, false, LOCAL, PACK>::CS,                       \
                                      WaveDP<M, RVAL, MODE, false, LOCAL, PACK>::BND,                      \
                                      M::NS,                                                               \
                                      M::MAXAT,                                                            \
                                      NWV, WaveDP<M, RVAL, MODE, false, LOCAL, PACK>::SEEDW};              \
        return &ki;                                                                                        \
    }

}  // namespace c4k
