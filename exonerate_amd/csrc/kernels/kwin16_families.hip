// the packed 16-bit region windows (c4_win16_kernel.h): two windows per lane, one wave per pair of window chains.  est2genome
// only (the family the packed score pass serves: the windows start from its 16-bit dumps).
#include "../c4_launch.h"
#include "../c4_win16_kernel.h"
namespace c4k {
#define WIN16_KERNEL(NAME, M, RV, WPEV, NWV)                                                                            \
    static hipError_t NAME##_launch(const LaunchArgs &a) {                                                            \
        hipLaunchKernelGGL((win16_kernel<M, RV, WPEV, NWV>), dim3(a.grid), dim3(64 * NWV), 0, a.stream, a.kp, a.seqs, a.jobs, \
                           a.aux, a.n_aux, a.results, a.scratch, a.queue);                                             \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static const KernelInfo NAME = {NAME##_launch, (const void *)win16_kernel<M, RV, WPEV, NWV>, #NAME, RV, WaveWin16<M, RV, -1>::CS, \
                                    WaveWin16<M, RV, -1>::BND_ALL, M::NS, M::MAXAT, NWV, WaveWin16<M, RV, -1>::SEEDW, 0, 1, 0, 1};
WIN16_KERNEL(kwin16_est2genome_r4w2, Est2GenomeDesc, 4, 2, 1)
WIN16_KERNEL(kwin16_est2genome_r3w3, Est2GenomeDesc, 3, 3, 1)
WIN16_KERNEL(kwin16_est2genome_r2w4, Est2GenomeDesc, 2, 4, 1)
WIN16_KERNEL(kwin16_est2genome_r6w2, Est2GenomeDesc, 6, 2, 1)
// the strips of a window on cooperating waves (WaveWin16::run<NW>)
WIN16_KERNEL(kwin16_est2genome_r4w2n4, Est2GenomeDesc, 4, 2, 4)
WIN16_KERNEL(kwin16_est2genome_r2w4n8, Est2GenomeDesc, 2, 4, 8)
WIN16_KERNEL(kwin16_est2genome_r2w4n4, Est2GenomeDesc, 2, 4, 4)
WIN16_KERNEL(kwin16_est2genome_r4w2n2, Est2GenomeDesc, 4, 2, 2)
WIN16_KERNEL(kwin16_est2genome_r4w3n2, Est2GenomeDesc, 4, 3, 2)
const KernelInfo *get_kernel_win16(int family, int variant) {
    if (family != FAM_EST2GENOME) return nullptr;
    switch (variant) {
        case 1: return &kwin16_est2genome_r3w3;
        case 2: return &kwin16_est2genome_r2w4;
        case 3: return &kwin16_est2genome_r6w2;
        case 4: return &kwin16_est2genome_r4w2n4;
        case 5: return &kwin16_est2genome_r2w4n8;
        case 6: return &kwin16_est2genome_r2w4n4;
        case 7: return &kwin16_est2genome_r4w2n2;
        case 9: return &kwin16_est2genome_r4w3n2;
        default: return &kwin16_est2genome_r4w2;
    }
}
}
