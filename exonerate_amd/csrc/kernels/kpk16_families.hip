// the packed 16-bit score pass (c4_viterbi16_kernel.h): two jobs per lane, 4 rows per lane, 4 waves per pair of jobs, held to
// 3 waves per SIMD — the best of the shapes measured (profiles/r03_pk16.md).  est2genome only: for protein2dna (no shadow
// payload to shed, a ring of four columns) the packed pass is 38 % SLOWER than the 32-bit one on config 3's shape
// (2 356 against 1 706 ms per pass), and affine runs no whole-rectangle score pass at config 2's size.
#include "../c4_launch.h"
#include "../c4_viterbi16_kernel.h"
namespace c4k {
#define PK16_KERNEL(NAME, M, RV, NWV, WPEV, VARV, D16V)                                                                \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((viterbi16_kernel_mw<M, RV, NWV, WPEV, VARV, D16V>), dim3(a.grid), dim3(64 * NWV), 0, a.stream,   \
                           a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue);                             \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)viterbi16_kernel_mw<M, RV, NWV, WPEV, VARV, D16V>, #NAME, RV, 2,  \
                                    WaveDP16<M, RV, VARV, D16V>::BND, M::NS, M::MAXAT, NWV, WaveDP16<M, RV, VARV, D16V>::SEEDW};
PK16_KERNEL(kpk16_est2genome, Est2GenomeDesc, 4, 4, 3, 0, false)
PK16_KERNEL(kpk16b_est2genome, Est2GenomeDesc, 4, 4, 3, 1, false)
PK16_KERNEL(kpk16c_est2genome, Est2GenomeDesc, 4, 4, 3, 2, false)
// variant 3: variant 1 with its column dumps in 16-bit form (Dump16), what the packed region windows (c4_win16_kernel.h) read
PK16_KERNEL(kpk16d_est2genome, Est2GenomeDesc, 4, 4, 3, 1, true)
// variant 4: variant 3 with its column loop fed from LDS only (c4_viterbi16_kernel.h, IO 1: column stage, query profile, every strip
// boundary a ring); for launches whose queries fit the four strips of a workgroup and whose targets hold at most six residue
// codes; the launch's code table arrives in LaunchArgs::aux
static hipError_t kpk16e_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 3, 1, true, 1>), dim3(a.grid), dim3(64 * 4), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16e_est2genome = {kpk16e_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 3, 1, true, 1>,
                                             "kpk16e_est2genome", 4, 2, WaveDP16<Est2GenomeDesc, 4, 1, true, 1>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 4, WaveDP16<Est2GenomeDesc, 4, 1, true, 1>::SEEDW};
// variant 5 (the default of the staged form): variant 4 with progress counters between the cooperating waves instead of a barrier per
// chunk -- a wave starts chunk k once the wave above has finished chunk k + 2 and the wave below chunk k - 5 (the ring slots it
// overwrites have been read): 303 -> 297 ms per launch of 4 096 pairs now that the four waves do the same work (round 3, with the
// fourth wave 1.7 x slower: 0.7 %)
static hipError_t kpk16f_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 3, 2, true, 1>), dim3(a.grid), dim3(64 * 4), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16f_est2genome = {kpk16f_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 3, 2, true, 1>,
                                             "kpk16f_est2genome", 4, 2, WaveDP16<Est2GenomeDesc, 4, 2, true, 1>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 4, WaveDP16<Est2GenomeDesc, 4, 2, true, 1>::SEEDW};
// variant 6: variant 5 on EIGHT cooperating waves of two rows per lane -- the same 1 024 query rows per workgroup on twice the
// waves, for launches with at most one pair of jobs per compute unit (the 512-pair shard of a strong-scaled run: 256 pairs of
// jobs on 256 CUs would otherwise be one wave per SIMD, which issues an instruction every ~5 cycles whatever else is free)
static hipError_t kpk16g_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 2, 8, 2, 2, true, 1>), dim3(a.grid), dim3(64 * 8), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16g_est2genome = {kpk16g_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 2, 8, 2, 2, true, 1>,
                                             "kpk16g_est2genome", 2, 2, WaveDP16<Est2GenomeDesc, 2, 2, true, 1>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 8, WaveDP16<Est2GenomeDesc, 2, 2, true, 1>::SEEDW};
// variant 7: the staged form with SIX rows per lane (strips of 384 rows: queries of up to 1 535 nt in the four strips of one
// workgroup, where four rows per lane need a second pass over the target for rows 1 024 ..: cDNAs of 1.1 kb ran at half the rate
// of 1 kb ones, bench.py configs.c4_query_1100), two waves per SIMD
static hipError_t kpk16h_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 6, 4, 2, 2, true, 1>), dim3(a.grid), dim3(64 * 4), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16h_est2genome = {kpk16h_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 6, 4, 2, 2, true, 1>,
                                             "kpk16h_est2genome", 6, 2, WaveDP16<Est2GenomeDesc, 6, 2, true, 1>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 4, WaveDP16<Est2GenomeDesc, 6, 2, true, 1>::SEEDW};
int pk16_staged_rows6() { return 6 * 64 * 4; }
// variant 9: variant 7 for queries of ANY length the packed guard lets through (up to ~3 190 rows): super-strips of 1 536 rows one
// after the other, the row between two of them through the workgroup's slab in memory (MEMC) -- cDNAs of 1.6 - 3 kb stay on the
// staged form (two or three passes over the target) instead of the form that loads per step
static hipError_t kpk16j_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 6, 4, 2, 2, true, 1, 6, true>), dim3(a.grid), dim3(64 * 4), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16j_est2genome = {kpk16j_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 6, 4, 2, 2, true, 1, 6, true>,
                                             "kpk16j_est2genome", 6, 2, WaveDP16<Est2GenomeDesc, 6, 2, true, 1, 6, true>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 4, WaveDP16<Est2GenomeDesc, 6, 2, true, 1, 6, true>::SEEDW};
// variant 8: variant 5 with a query profile for EIGHT residue codes (targets with IUPAC ambiguity codes beside A C G T N: the six-code
// form sent such batches to the form that loads per step, 0.77 of the rate, bench.py configs.c4_eight_codes): 61.6 KB of LDS, two
// workgroups per CU, compiled for two waves per SIMD (256 registers: nothing in scratch)
static hipError_t kpk16i_est2genome_launch(const LaunchArgs &a) {
    hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 2, 2, true, 1, 8>), dim3(a.grid), dim3(64 * 4), 0, a.stream,
                       a.kp, a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue, reinterpret_cast<const uint8_t *>(a.aux));
    return hipGetLastError();
}
static const KernelInfo kpk16i_est2genome = {kpk16i_est2genome_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, 4, 4, 2, 2, true, 1, 8>,
                                             "kpk16i_est2genome", 4, 2, WaveDP16<Est2GenomeDesc, 4, 2, true, 1, 8>::BND, Est2GenomeDesc::NS,
                                             Est2GenomeDesc::MAXAT, 4, WaveDP16<Est2GenomeDesc, 4, 2, true, 1, 8>::SEEDW};
int pk16_staged_codes() { return WaveDP16<Est2GenomeDesc, 4, 1, true, 1>::NCODE; }
int pk16_staged_rows() { return 4 * 64 * 4; }
// the packed splice array of variant 1 (ss16_kernel): n positions of the batch's concatenated targets
hipError_t pk16_build_splice(int family, const KParams *kp, const int *ss, long long ss_stride, long long n, void *out, hipStream_t s) {
    if (family != FAM_EST2GENOME) return hipErrorInvalidValue;
    hipLaunchKernelGGL((ss16_kernel<Est2GenomeDesc>), dim3(4096), dim3(256), 0, s, kp, ss, ss_stride, n, (uint2 *)out);
    return hipGetLastError();
}
const KernelInfo *get_kernel_pk16(int family, int variant) {
    if (family != FAM_EST2GENOME) return nullptr;
    return variant == 9 ? &kpk16j_est2genome : variant == 8 ? &kpk16i_est2genome : variant == 7 ? &kpk16h_est2genome : variant == 6 ? &kpk16g_est2genome : variant == 5 ? &kpk16f_est2genome : variant == 4 ? &kpk16e_est2genome : variant == 3 ? &kpk16d_est2genome : variant == 2 ? &kpk16c_est2genome : variant == 1 ? &kpk16b_est2genome : &kpk16_est2genome;
}
}
