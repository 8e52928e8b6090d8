// c4_sdp_launch.h — the compiled SDP passes (c4_sdp_wave.h) by model family: one translation unit per family under
// kernels/ (ksdp_*.hip), looked up by c4gpu_sdp_batch (c4_sdp_dev.inc).
#pragma once
#include <hip/hip_runtime.h>
#include "c4_sdp_wave.h"

namespace c4sdp {

typedef hipError_t (*SdpPassLaunch)(const SdpLaunch &, int n_jobs, hipStream_t);
struct SdpKernels { SdpPassLaunch rev, fwd; const void *rev_func, *fwd_func; };

const SdpKernels *sdp_kernels_affine();
const SdpKernels *sdp_kernels_protein2dna();
const SdpKernels *sdp_kernels_est2genome();
const SdpKernels *sdp_kernels_protein2genome();

#define C4SDP_DEFINE_KERNELS(SYMBOL, M, BND)                                                                           \
    static hipError_t SYMBOL##_rev(const SdpLaunch &a, int n, hipStream_t s) {                                         \
        hipLaunchKernelGGL((sdp_wave_kernel<M, false, BND>), dim3(n), dim3(64), 0, s, a);                              \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    static hipError_t SYMBOL##_fwd(const SdpLaunch &a, int n, hipStream_t s) {                                         \
        hipLaunchKernelGGL((sdp_wave_kernel<M, true, BND>), dim3(n), dim3(64), 0, s, a);                               \
        return hipGetLastError();                                                                                      \
    }                                                                                                                  \
    const SdpKernels *SYMBOL() {                                                                                       \
        static const SdpKernels k = {SYMBOL##_rev, SYMBOL##_fwd, (const void *)sdp_wave_kernel<M, false, BND>,          \
                                     (const void *)sdp_wave_kernel<M, true, BND>};                                     \
        return &k;                                                                                                     \
    }

}  // namespace c4sdp
