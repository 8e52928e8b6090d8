// one shape of the packed 16-bit checkpoint pass with the strips of a pair of jobs on cooperating waves (WaveCK16::run<NW>;
// kernels/kck16_families.hip holds the table)
#include "../c4_launch.h"
#include "../c4_ckpt16_kernel.h"
namespace c4k {
#define CK16_KERNEL_NW(NAME, M, RV, WPEV, ROOTEDV, NWV)                                                                 \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((ckpt16_kernel<M, RV, WPEV, ROOTEDV, NWV>), dim3(a.grid), dim3(64 * NWV), 0, a.stream, a.kp, a.seqs, \
                           a.jobs, a.aux, a.n_aux, a.results, a.vsas, a.scratch, a.queue);                             \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    const KernelInfo *NAME##_info() {                                                                                  \
    static const KernelInfo ki = {NAME##_launch, (const void *)ckpt16_kernel<M, RV, WPEV, ROOTEDV, NWV>, #NAME, RV, WaveCK16<M, RV>::CS, \
                                    WaveCK16<M, RV>::BND, M::NS, M::MAXAT, NWV, 0, WaveCK16<M, RV>::CKW, 1,            \
                                    WaveCK16<M, RV, Roots<M>::root(0)>::CKW, 1};                                       \
    return &ki; }
CK16_KERNEL_NW(kck16r_est2genome_r4w3n4, Est2GenomeDesc, 4, 3, true, 4)
}
