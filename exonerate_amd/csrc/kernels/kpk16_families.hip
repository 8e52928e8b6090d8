// the packed 16-bit score pass (c4_viterbi16_kernel.h): two jobs per lane, 4 rows per lane, 4 waves per pair of jobs, held to
// 3 waves per SIMD — the best of the shapes measured (profiles/r03_pk16.md).  est2genome only: for protein2dna (no shadow
// payload to shed, a ring of four columns) the packed pass is 38 % SLOWER than the 32-bit one on config 3's shape
// (2 356 against 1 706 ms per pass), and affine runs no whole-rectangle score pass at config 2's size.
#include "../c4_launch.h"
#include "../c4_viterbi16_kernel.h"
namespace c4k {
#define PK16_KERNEL(NAME, M, RV, NWV, WPEV)                                                                            \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((viterbi16_kernel_mw<M, RV, NWV, WPEV>), dim3(a.grid), dim3(64 * NWV), 0, a.stream, a.kp,   \
                           a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue);                                   \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)viterbi16_kernel_mw<M, RV, NWV, WPEV>, #NAME, RV, 2,  \
                                    WaveDP16<M, RV>::BND, M::NS, M::MAXAT, NWV, WaveDP16<M, RV>::SEEDW};
PK16_KERNEL(kpk16_est2genome, Est2GenomeDesc, 4, 4, 3)
const KernelInfo *get_kernel_pk16(int family) { return family == FAM_EST2GENOME ? &kpk16_est2genome : nullptr; }
}
