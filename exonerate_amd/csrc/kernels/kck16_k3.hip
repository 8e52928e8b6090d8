// one shape of the packed 16-bit checkpoint pass (kernels/kck16_families.hip holds the table): a translation unit each, so that
// the build compiles them side by side (the six together were the longest single compile of the library: 10 minutes)
#include "../c4_launch.h"
#include "../c4_ckpt16_kernel.h"
namespace c4k {
#define CK16_KERNEL(NAME, M, RV, WPEV, ROOTEDV)                                                                         \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((ckpt16_kernel<M, RV, WPEV, ROOTEDV>), dim3(a.grid), dim3(64), 0, a.stream, a.kp, a.seqs, a.jobs,    \
                           a.aux, a.n_aux, a.results, a.vsas, a.scratch, a.queue);                                     \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    const KernelInfo *NAME##_info() {                                                                                  \
    static const KernelInfo ki = {NAME##_launch, (const void *)ckpt16_kernel<M, RV, WPEV, ROOTEDV>, #NAME, RV, WaveCK16<M, RV>::CS, \
                                    WaveCK16<M, RV>::BND, M::NS, M::MAXAT, 1, 0, WaveCK16<M, RV>::CKW, 1,              \
                                    WaveCK16<M, RV, Roots<M>::root(0)>::CKW};                                          \
    return &ki; }
CK16_KERNEL(kck16r_est2genome_r6w2, Est2GenomeDesc, 6, 2, true)
}
