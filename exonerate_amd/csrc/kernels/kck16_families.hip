// the packed 16-bit checkpoint pass (c4_ckpt16_kernel.h): two jobs per lane, one wave per pair of jobs.  est2genome only (the
// family whose reduced-space passes dominate a step: introns make the aligned regions tens of thousands of columns wide).
// Two forms: every inner state (jobs whose root is not known), and ROOTED: the component of the state the path's END is
// entered from (one strand's four states instead of eight).
#include "../c4_launch.h"
namespace c4k {
// (the kernels themselves: kernels/kck16_k0.hip ... kck16_k5.hip, one shape per translation unit)
const KernelInfo *kck16_est2genome_r3w2_info();
const KernelInfo *kck16_est2genome_r4w2_info();
const KernelInfo *kck16r_est2genome_r4w2_info();
const KernelInfo *kck16r_est2genome_r6w2_info();
const KernelInfo *kck16r_est2genome_r3w3_info();
const KernelInfo *kck16r_est2genome_r2w4_info();
const KernelInfo *kck16r_est2genome_r4w2n4_info();       // ... with the strips on 4 / 2 / 3 cooperating waves
const KernelInfo *kck16r_est2genome_r4w2n2_info();
const KernelInfo *kck16r_est2genome_r6w2n3_info();
const KernelInfo *kck16r_est2genome_r4w3n4_info();       // three waves per SIMD (168 registers, some of the state in scratch)
// variant: shapes kept for measurement (0 = the default of each form)
const KernelInfo *get_kernel_ck16(int family, int variant, bool rooted) {
    if (family != FAM_EST2GENOME) return nullptr;
    if (rooted) {
        // six rows per lane at two waves per SIMD: 517 ms of checkpoint launches per four steps of the north-star batch
        // against 550 (4 x 2), 855 (3 x 3) and 939 (2 x 4): gpurun_out/r4d, profiles/r04_step.md
        switch (variant) {
            case 1: return kck16r_est2genome_r4w2_info();
            case 2: return kck16r_est2genome_r3w3_info();
            case 3: return kck16r_est2genome_r2w4_info();
            case 4: return kck16r_est2genome_r4w2n4_info();
            case 5: return kck16r_est2genome_r4w2n2_info();
            case 6: return kck16r_est2genome_r6w2n3_info();
            case 8: return kck16r_est2genome_r4w3n4_info();
            default: return kck16r_est2genome_r6w2_info();
        }
    }
    return variant == 1 ? kck16_est2genome_r3w2_info() : kck16_est2genome_r4w2_info();
}
}
