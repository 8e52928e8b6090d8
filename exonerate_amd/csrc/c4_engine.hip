// c4_engine.hip — host engine of libc4gpu.so: device context, sequence preparation kernels, batched job
// launches and the Optimal_* orchestration (exonerate src/c4/optimal.c) over batches of independent pairs.
//
// There is NO CPU fallback here: every entry point that computes needs a HIP device and fails loudly
// (c4gpu_last_error) without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <map>
#include <memory>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>

#include "c4gpu.h"
#include "c4_internal.h"
#include "c4_config.h"
#include "c4_memrule.h"
#include "c4_launch.h"
#include "c4_sdp_launch.h"
#include "c4_sdp_host.h"

using namespace c4k;

namespace c4h {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace c4h

#define HIP_OK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            c4h::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                 \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)

struct c4gpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;                      // c4gpu_ctx_own_stream: destroyed with the context
    hipDeviceProp_t prop;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // accumulated statistics of the Viterbi kernel launches (HIP events on the launch stream)
    double kernel_ms[4] = {0, 0, 0, 0};            // indexed by Viterbi mode
    int64_t kernel_launches[4] = {0, 0, 0, 0}, kernel_cells[4] = {0, 0, 0, 0};
    bool timing = false;
    // share of pairs whose region-pass score reached the threshold in recent Optimal_find_path batches
    // (-1: not known yet): decides whether a score-only pass goes first (find_path_batch, step 1);
    // kept apart for first alignments [0] and the later rounds of the sub-optimal loop [1], which mostly fail
    double hit_rate[2] = {-1.0, -1.0};
    // share of pairs whose region start the windowed region pass found within its hop budget in recent batches
    // (-1: not known yet): alignments that span most of their target make the two-pass form the dearer one
    double window_rate = -1.0;
    // SDP (c4_sdp_dev.inc): the arena the passes' step records grow in, kept between calls
    void *sdp_arena = nullptr;
    size_t sdp_arena_bytes = 0;
    bool sdp_arena_keep = false;                   // c4gpu_ctx_sdp_reserve: the arena stays with the context between batches
};

namespace {

#include "c4_engine_mem.inc"
#include "c4_engine_staging.inc"
#include "c4_engine_launch.inc"
#include "c4_engine_passes.inc"
#include "c4_engine_find_path.inc"
}  // namespace

#include "c4_engine_abi.inc"
// ---- SDP on the device (seeded flavour): its own file, same translation unit ------------------------------------
#include "c4_sdp_dev.inc"
#include "c4_seed_dev.inc"
