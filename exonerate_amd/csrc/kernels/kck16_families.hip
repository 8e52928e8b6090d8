// the packed 16-bit checkpoint pass (c4_ckpt16_kernel.h): two jobs per lane, one wave per pair of jobs.  est2genome only (the
// family whose reduced-space passes dominate a step: introns make the aligned regions tens of thousands of columns wide).
#include "../c4_launch.h"
#include "../c4_ckpt16_kernel.h"
namespace c4k {
#define CK16_KERNEL(NAME, M, RV, WPEV)                                                                                  \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((ckpt16_kernel<M, RV, WPEV>), dim3(a.grid), dim3(64), 0, a.stream, a.kp, a.seqs, a.jobs,    \
                           a.n_jobs, a.results, a.vsas, a.scratch, a.queue);                                           \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)ckpt16_kernel<M, RV, WPEV>, #NAME, RV, WaveCK16<M, RV>::CS, \
                                    WaveCK16<M, RV>::BND, M::NS, M::MAXAT, 1, 0, WaveCK16<M, RV>::CKW};
CK16_KERNEL(kck16_est2genome_r3w2, Est2GenomeDesc, 3, 2)
CK16_KERNEL(kck16_est2genome_r4w2, Est2GenomeDesc, 4, 2)
CK16_KERNEL(kck16_est2genome_r2w3, Est2GenomeDesc, 2, 3)
CK16_KERNEL(kck16_est2genome_r2w2, Est2GenomeDesc, 2, 2)
CK16_KERNEL(kck16_est2genome_r1w4, Est2GenomeDesc, 1, 4)
const KernelInfo *get_kernel_ck16(int family, int variant) {
    if (family != FAM_EST2GENOME) return nullptr;
    switch (variant) {
        case 1: return &kck16_est2genome_r4w2;
        case 2: return &kck16_est2genome_r2w3;
        case 3: return &kck16_est2genome_r2w2;
        case 4: return &kck16_est2genome_r1w4;
        default: return &kck16_est2genome_r3w2;
    }
}
}
