// the packed 16-bit checkpoint pass (c4_ckpt16_kernel.h): two jobs per lane, one wave per pair of jobs.  est2genome only (the
// family whose reduced-space passes dominate a step: introns make the aligned regions tens of thousands of columns wide).
// Two forms: every inner state (jobs whose root is not known), and ROOTED: the component of the state the path's END is
// entered from (one strand's four states instead of eight).
#include "../c4_launch.h"
#include "../c4_ckpt16_kernel.h"
namespace c4k {
#define CK16_KERNEL(NAME, M, RV, WPEV, ROOTEDV)                                                                         \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((ckpt16_kernel<M, RV, WPEV, ROOTEDV>), dim3(a.grid), dim3(64), 0, a.stream, a.kp, a.seqs, a.jobs,    \
                           a.aux, a.n_aux, a.results, a.vsas, a.scratch, a.queue);                                     \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)ckpt16_kernel<M, RV, WPEV, ROOTEDV>, #NAME, RV, WaveCK16<M, RV>::CS, \
                                    WaveCK16<M, RV>::BND, M::NS, M::MAXAT, 1, 0, WaveCK16<M, RV>::CKW, 1,              \
                                    WaveCK16<M, RV, Roots<M>::root(0)>::CKW};
CK16_KERNEL(kck16_est2genome_r3w2, Est2GenomeDesc, 3, 2, false)
CK16_KERNEL(kck16_est2genome_r4w2, Est2GenomeDesc, 4, 2, false)
CK16_KERNEL(kck16r_est2genome_r4w2, Est2GenomeDesc, 4, 2, true)
CK16_KERNEL(kck16r_est2genome_r6w2, Est2GenomeDesc, 6, 2, true)
CK16_KERNEL(kck16r_est2genome_r3w3, Est2GenomeDesc, 3, 3, true)
CK16_KERNEL(kck16r_est2genome_r2w4, Est2GenomeDesc, 2, 4, true)
// variant: shapes kept for measurement (0 = the default of each form)
const KernelInfo *get_kernel_ck16(int family, int variant, bool rooted) {
    if (family != FAM_EST2GENOME) return nullptr;
    if (rooted) {
        // six rows per lane at two waves per SIMD: 517 ms of checkpoint launches per four steps of the north-star batch
        // against 550 (4 x 2), 855 (3 x 3) and 939 (2 x 4): gpurun_out/r4d, profiles/r04_step.md
        switch (variant) {
            case 1: return &kck16r_est2genome_r4w2;
            case 2: return &kck16r_est2genome_r3w3;
            case 3: return &kck16r_est2genome_r2w4;
            default: return &kck16r_est2genome_r6w2;
        }
    }
    return variant == 1 ? &kck16_est2genome_r3w2 : &kck16_est2genome_r4w2;
}
}
