// the packed 16-bit score pass (c4_viterbi16_kernel.h) of the est2genome family: two jobs per lane, NW waves per pair of jobs
#include "../c4_launch.h"
#include "../c4_viterbi16_kernel.h"
namespace c4k {
#define PK16_VARIANT(NAME, RV, NWV, WPEV)                                                                               \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((viterbi16_kernel_mw<Est2GenomeDesc, RV, NWV, WPEV>), dim3(a.grid), dim3(64 * NWV), 0, a.stream, a.kp,    \
                           a.seqs, a.jobs, a.n_jobs, a.results, a.scratch, a.queue);                                   \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)viterbi16_kernel_mw<Est2GenomeDesc, RV, NWV, WPEV>, #NAME, RV, 2, \
                                    WaveDP16<Est2GenomeDesc, RV>::BND, Est2GenomeDesc::NS, Est2GenomeDesc::MAXAT, NWV,   \
                                    WaveDP16<Est2GenomeDesc, RV>::SEEDW};
// 4 rows per lane, 4 waves per pair of jobs, held to 3 waves per SIMD: the best of the shapes measured (profiles/r03_pk16.md)
PK16_VARIANT(kpk16_e2g_r4w4c3, 4, 4, 3)
const KernelInfo *get_kernel_pk16(int family) { return family == FAM_EST2GENOME ? &kpk16_e2g_r4w4c3 : nullptr; }
}
