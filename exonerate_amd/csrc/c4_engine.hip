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

// Host loops over hundreds of thousands of independent small items (the sub-alignments between checkpoints: 778 443 per
// pass for 4 096 pairs of 1 kb x 1 kb under the reference's -D 32 rule) are split over a few threads; `fn(first, last)`
// works on its own items only.  C4GPU_HOST_THREADS=1 keeps everything on the calling thread.
template <typename F> void parallel_for(long long n, long long min_per_thread, F &&fn) {
    static const int hw = [] {
        const int v = c4cfg::num(c4cfg::HOST_THREADS, (int)std::thread::hardware_concurrency());
        return std::max(1, std::min(v, 16));
    }();
    const int t = (int)std::min<long long>(hw, n / std::max<long long>(1, min_per_thread));
    if (t <= 1) { if (n > 0) fn(0LL, n); return; }
    const long long chunk = (n + t - 1) / t;
    std::vector<std::thread> th;
    for (int k = 1; k < t; k++) {
        const long long a = k * chunk, b = std::min(n, a + chunk);
        if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); });
    }
    fn(0LL, std::min(n, chunk));
    for (auto &x : th) x.join();
}

// ---- small RAII device buffer ------------------------------------------------------------------------------
// hipFree waits for the whole device: with two launch lanes and a staging stream in flight, a launch buffer that has to grow
// in the middle of a step would stall its lane until the other lane's kernels (hundreds of ms) have finished.  A buffer that
// is outgrown is therefore retired, not freed: it goes onto this list and is freed when its owner is (batch / stage / context
// destruction: nothing is in flight then), or at once when the list holds more than its cap: 16 GB, or a tenth of the device's
// memory where that is less (set when a context opens).  An allocation that fails frees the list and tries once more (DevBuf::alloc,
// PinBuf::reserve, the SDP arena), and every sizing decision that asks the device how much is free counts the list as free
// (dev_mem_info): what waits here is memory nobody uses.
struct RetiredBuffers {
    std::mutex lock;
    std::vector<std::pair<void *, size_t>> list;
    size_t bytes = 0;
    size_t cap = (size_t)16 << 30;
    void set_cap_for(size_t device_bytes) {
        std::lock_guard<std::mutex> hold(lock);
        cap = std::min<size_t>((size_t)16 << 30, device_bytes / 10);
    }
    void retire(void *p, size_t n) {
        std::vector<std::pair<void *, size_t>> drop;
        {
            std::lock_guard<std::mutex> hold(lock);
            list.emplace_back(p, n); bytes += n;
            if (bytes > cap) { drop.swap(list); bytes = 0; }
        }
        for (auto &d : drop) (void)hipFree(d.first);
    }
    void flush() {
        std::vector<std::pair<void *, size_t>> drop;
        { std::lock_guard<std::mutex> hold(lock); drop.swap(list); bytes = 0; }
        for (auto &d : drop) (void)hipFree(d.first);
    }
};
static RetiredBuffers g_retired;
// hipMalloc that gives the retired buffers back to the device before it gives up
static hipError_t dev_malloc(void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        g_retired.flush();
        e = hipMalloc(p, bytes);
    }
    return e;
}
// free / total device memory as a sizing decision should see them: what is only waiting to be freed counts as free (it is
// NOT freed here: hipFree waits for every kernel in flight, which is why the buffers were retired; the allocation that follows
// frees them if it has to)
static hipError_t dev_mem_info(size_t *free_bytes, size_t *total_bytes) {
    const hipError_t e = hipMemGetInfo(free_bytes, total_bytes);
    if (e == hipSuccess) { std::lock_guard<std::mutex> hold(g_retired.lock); *free_bytes += g_retired.bytes; }
    return e;
}

// Small transfers between the passes go through page-locked memory.  A copy to or from pageable memory is carried out by a
// copy KERNEL of one workgroup (__amd_rocclr_copyBuffer), which needs a compute unit with room for it -- and while the other
// launch lane's persistent kernel fills the device there is none until one of its workgroups retires: the job list of the next
// pass waited 30-105 ms per pass for that (rocprofv3 kernel trace of the wide-region batch, profiles/r05_wide_trace.md).  From
// page-locked memory the same copy is a DMA transfer that needs no compute unit.  Every thread that talks to the device owns
// one arena of page-locked memory (taken from a pool, handed back when the thread ends): an upload copies its source into the
// arena first (the source is consumed when upload() returns, as with a pageable copy), a download lands in the arena and is
// copied out to its destination by c4_stream_sync(), which every wait for a stream in this library goes through.
struct PinArena {
    uint8_t *base = nullptr;
    size_t cap = 0, head = 0;
    struct Pending { void *dst; const void *src; size_t bytes; };
    std::vector<Pending> pending;
    hipStream_t stream = nullptr;
    bool stream_set = false;
};
struct PinArenaPool {
    std::mutex lock;
    std::vector<PinArena *> idle;
    PinArena *take() {
        {
            std::lock_guard<std::mutex> hold(lock);
            if (!idle.empty()) { PinArena *a = idle.back(); idle.pop_back(); return a; }
        }
        PinArena *a = new PinArena;
        const size_t cap = (size_t)64 << 20;
        if (hipHostMalloc((void **)&a->base, cap, hipHostMallocDefault) == hipSuccess) a->cap = cap;
        else { (void)hipGetLastError(); a->base = nullptr; a->cap = 0; }            // no arena: transfers go directly
        return a;
    }
    void give(PinArena *a) { std::lock_guard<std::mutex> hold(lock); idle.push_back(a); }
};
static PinArenaPool g_pin_pool;
struct PinArenaRef {
    PinArena *a = nullptr;
    ~PinArenaRef() { if (a) { a->pending.clear(); a->head = 0; a->stream_set = false; g_pin_pool.give(a); } }
    PinArena *get() { if (!a) a = g_pin_pool.take(); return a; }
};
static thread_local PinArenaRef t_pin;
// C4GPU_FREE_NOW=1: ~DevBuf frees at once (as before round 5's end); C4GPU_DL_SYNC_FIRST=1: see DevBuf::download;
// C4GPU_PIN_XFER=0: pageable copies, as before round 5
static inline bool g_free_now_q() { return c4cfg::nonzero(c4cfg::FREE_NOW); }
static inline bool g_dl_sync_first_q() { return c4cfg::nonzero(c4cfg::DL_SYNC_FIRST); }
static inline bool g_pin_off_q() { return c4cfg::is(c4cfg::PIN_XFER, 0); }

// every wait for a stream: the downloads of this thread that landed in its arena reach their destinations
static hipError_t c4_stream_sync(hipStream_t s) {
    const hipError_t e = hipStreamSynchronize(s);
    PinArena *a = t_pin.a;
    if (a && (!a->pending.empty() || a->head)) {
        if (a->stream_set && a->stream != s) (void)hipStreamSynchronize(a->stream);     // (a thread uses one stream; be safe)
        for (const PinArena::Pending &pd : a->pending) memcpy(pd.dst, pd.src, pd.bytes);
        a->pending.clear();
        a->head = 0;
        a->stream_set = false;
    }
    return e;
}
// bytes of the calling thread's arena for a transfer on stream s (nullptr: too large, or no arena -- copy directly)
static uint8_t *pin_slot(size_t bytes, hipStream_t s) {
    if (g_pin_off_q() || !bytes) return nullptr;
    PinArena *a = t_pin.get();
    if (!a->cap || bytes > a->cap / 4) return nullptr;
    if (a->stream_set && a->stream != s) { (void)c4_stream_sync(a->stream); }
    const size_t at = (a->head + 63) & ~(size_t)63;
    if (at + bytes > a->cap) {
        // full: everything in flight must land before its bytes are reused (the pending copy-outs are carried out now;
        // their destinations are not read before the caller's own wait, which then finds nothing left to do)
        (void)c4_stream_sync(s);
        return pin_slot(bytes, s);
    }
    a->head = at + bytes;
    a->stream = s; a->stream_set = true;
    return a->base + at;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    // (retired, not freed: hipFree waits for every kernel on the device, those of other threads' batches included -- the HSP
    // extension of the drop-in's main thread waited 0.35 s for the SDP passes of the flight beside it at the end of each of
    // its calls; what is retired is freed at the next flush: context / batch / stage destroy, or once 16 GB are waiting)
    ~DevBuf() { if (p) { if (g_free_now_q()) (void)hipFree(p); else g_retired.retire(p, n * sizeof(T)); } }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    int alloc(size_t count) {
        if (count <= n && p) return 0;
        // every buffer is made an eighth larger than asked (small ones twice as large): in a stream of batches of about one
        // size a launch buffer would otherwise be outgrown whenever a batch needs a little more
        if (p) { g_retired.retire(p, n * sizeof(T)); p = nullptr; n = 0; }
        count += count / 8;
        if (count * sizeof(T) < ((size_t)1 << 20)) count *= 2;
        if (!count) count = 1;
        HIP_OK(dev_malloc((void **)&p, count * sizeof(T)));
        n = count;
        // C4GPU_FILL_ALLOC=<byte>: every new device buffer starts as that byte (a test hook: nothing may depend on what a
        // buffer held before its first write)
        if (c4cfg::has(c4cfg::FILL_ALLOC)) HIP_OK(hipMemset(p, c4cfg::num(c4cfg::FILL_ALLOC, 0), count * sizeof(T)));
        return 0;
    }
    int upload(const T *src, size_t count, hipStream_t s) {
        if (alloc(count)) return -1;
        if (!count) return 0;
        if (uint8_t *slot = pin_slot(count * sizeof(T), s)) {
            memcpy(slot, src, count * sizeof(T));
            HIP_OK(hipMemcpyAsync(p, slot, count * sizeof(T), hipMemcpyHostToDevice, s));
        } else {
            HIP_OK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
        }
        return 0;
    }
    // the data is at `dst` after the next c4_stream_sync(s) of the calling thread
    int download(T *dst, size_t count, hipStream_t s) const {
        if (!count) return 0;
        // C4GPU_DL_SYNC_FIRST=1: a read-back is queued only once everything in front of it in its stream is over.  A DMA copy that
        // waits for a kernel waits in its engine's queue, and that queue is shared by all streams of the process: the read-backs
        // of another thread's batch then sit behind it until THIS stream's kernel ends (seen: a word scan's 8-byte read-back held
        // up for the 0.36 s of another thread's SDP passes).  The one place where that happened -- a batch of SDP passes beside
        // the drop-in's main thread -- waits for its passes itself before it queues anything (sdp_run_device); doing it for
        // every read-back costs nothing in a warm run (425.8 ms either way) but serialises the host's work between a launch and
        // its wait with the kernel, and the first steps of a run, which still allocate their launch buffers there, took
        // 805 ms instead of 417 (`bench.py` with its default three steps).  So: off by default.
        if (g_dl_sync_first_q()) (void)hipStreamSynchronize(s);
        if (uint8_t *slot = pin_slot(count * sizeof(T), s)) {
            HIP_OK(hipMemcpyAsync(slot, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
            t_pin.a->pending.push_back(PinArena::Pending{dst, slot, count * sizeof(T)});
        } else {
            HIP_OK(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
        }
        return 0;
    }
    // zeroes the first `count` elements without a fill kernel (which, like a copy kernel, waits for a compute unit)
    int zero(size_t count, hipStream_t s) {
        if (alloc(count)) return -1;
        if (!count) return 0;
        if (uint8_t *slot = pin_slot(count * sizeof(T), s)) {
            memset(slot, 0, count * sizeof(T));
            HIP_OK(hipMemcpyAsync(p, slot, count * sizeof(T), hipMemcpyHostToDevice, s));
        } else {
            HIP_OK(hipMemsetAsync(p, 0, count * sizeof(T), s));
        }
        return 0;
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
};

// Page-locked host memory that is kept between uses (c4gpu_stage: the gathered residues of the next batch): the DMA engine
// reads it at PCIe speed, hipMemcpyAsync returns at once, and nothing is page-faulted in after the first use.
struct PinBuf {
    uint8_t *p = nullptr;
    size_t n = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    int reserve(size_t bytes) {
        if (bytes <= n && p) return 0;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        bytes += bytes / 8;                      // the next batch of a stream of batches is about this size, rarely the same
        hipError_t e = hipHostMalloc((void **)&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); g_retired.flush(); e = hipHostMalloc((void **)&p, bytes, hipHostMallocDefault); }
        HIP_OK(e);
        n = bytes;
        return 0;
    }
};

// ---- which compiled family does a flattened model belong to? ----------------------------------------------
template <class M>
bool model_matches(const c4gpu_model &m) {
    if (m.n_states != M::NS || m.n_transitions != M::NT || m.n_calcs != M::NC || m.n_shadows != M::NSH) return false;
    if (m.start_state != M::START || m.end_state != M::END || m.total_shadow_designations != M::NDES) return false;
    for (int k = 0; k < M::NT; k++) {
        const c4gpu_transition &t = m.transitions[k];
        const TrDesc &d = M::tr[k];
        if (t.input != d.in || t.output != d.out || t.advance_query != d.aq || t.advance_target != d.at ||
            t.calc != d.calc || t.label != d.label || t.dst_shadow_mask != d.dst_shadow_mask) return false;
    }
    for (int c = 0; c < M::NC; c++) {
        const int kind = m.calcs[c].kind, dk = M::calc[c].kind;
        const bool both_11 = (kind == C4GPU_CALC_MATCH_DNA || kind == C4GPU_CALC_MATCH_PROTEIN) &&
                             (dk == C4GPU_CALC_MATCH_DNA || dk == C4GPU_CALC_MATCH_PROTEIN);
        if (!both_11 && kind != dk) return false;
        if (kind >= C4GPU_CALC_SPLICE_PRE && m.calcs[c].param != M::calc[c].param) return false;
        if (m.calcs[c].protect != M::calc[c].protect) return false;
    }
    for (int s = 0; s < M::NSH; s++) {
        if (m.shadows[s].designation != M::sh[s].designation || m.shadows[s].on_target != M::sh[s].on_target ||
            m.shadows[s].src_state_mask != M::sh[s].src_state_mask ||
            m.shadows[s].dst_transition_mask != M::sh[s].dst_transition_mask) return false;
    }
    return true;
}

int model_family(const c4gpu_model &m) {
    if (model_matches<UngappedDesc>(m)) return FAM_UNGAPPED;
    if (model_matches<AffineDesc>(m)) return FAM_AFFINE;
    if (model_matches<Est2GenomeDesc>(m)) return FAM_EST2GENOME;
    if (model_matches<UngappedP2DDesc>(m)) return FAM_UNGAPPED_P2D;
    if (model_matches<Protein2DnaDesc>(m)) return FAM_PROTEIN2DNA;
    if (model_matches<Protein2GenomeDesc>(m)) return FAM_PROTEIN2GENOME;
    if (model_matches<AffineStartDesc>(m)) return FAM_AFFINE_START;
    if (model_matches<AffineEndDesc>(m)) return FAM_AFFINE_END;
    if (model_matches<AffineJoinDesc>(m)) return FAM_AFFINE_JOIN;
    if (model_matches<Est2GenomeFwdStartDesc>(m)) return FAM_EST2GENOME_FWD_START;
    if (model_matches<Est2GenomeFwdEndDesc>(m)) return FAM_EST2GENOME_FWD_END;
    if (model_matches<Est2GenomeFwdJoinDesc>(m)) return FAM_EST2GENOME_FWD_JOIN;
    if (model_matches<Est2GenomeRevStartDesc>(m)) return FAM_EST2GENOME_REV_START;
    if (model_matches<Est2GenomeRevEndDesc>(m)) return FAM_EST2GENOME_REV_END;
    if (model_matches<Est2GenomeRevJoinDesc>(m)) return FAM_EST2GENOME_REV_JOIN;
    if (model_matches<Protein2DnaStartDesc>(m)) return FAM_PROTEIN2DNA_START;
    if (model_matches<Protein2DnaEndDesc>(m)) return FAM_PROTEIN2DNA_END;
    if (model_matches<Protein2DnaJoinDesc>(m)) return FAM_PROTEIN2DNA_JOIN;
    if (model_matches<Protein2GenomeStartDesc>(m)) return FAM_PROTEIN2GENOME_START;
    if (model_matches<Protein2GenomeEndDesc>(m)) return FAM_PROTEIN2GENOME_END;
    if (model_matches<Protein2GenomeJoinDesc>(m)) return FAM_PROTEIN2GENOME_JOIN;
    if (model_matches<Est2GenomeFwdSpanSrcDesc>(m)) return FAM_EST2GENOME_FWD_SPAN_SRC;
    if (model_matches<Est2GenomeFwdSpanDstDesc>(m)) return FAM_EST2GENOME_FWD_SPAN_DST;
    if (model_matches<Est2GenomeRevSpanSrcDesc>(m)) return FAM_EST2GENOME_REV_SPAN_SRC;
    if (model_matches<Est2GenomeRevSpanDstDesc>(m)) return FAM_EST2GENOME_REV_SPAN_DST;
    if (model_matches<Protein2GenomePhase0SpanSrcDesc>(m)) return FAM_PROTEIN2GENOME_PHASE0_SPAN_SRC;
    if (model_matches<Protein2GenomePhase0SpanDstDesc>(m)) return FAM_PROTEIN2GENOME_PHASE0_SPAN_DST;
    if (model_matches<Protein2GenomePhase1SpanSrcDesc>(m)) return FAM_PROTEIN2GENOME_PHASE1_SPAN_SRC;
    if (model_matches<Protein2GenomePhase1SpanDstDesc>(m)) return FAM_PROTEIN2GENOME_PHASE1_SPAN_DST;
    if (model_matches<Protein2GenomePhase2SpanSrcDesc>(m)) return FAM_PROTEIN2GENOME_PHASE2_SPAN_SRC;
    if (model_matches<Protein2GenomePhase2SpanDstDesc>(m)) return FAM_PROTEIN2GENOME_PHASE2_SPAN_DST;
    return -1;
}

static bool family_is_p2g_span(int fam) {
    return fam >= FAM_PROTEIN2GENOME_PHASE0_SPAN_SRC && fam <= FAM_PROTEIN2GENOME_PHASE2_SPAN_DST;
}
bool family_is_p2d(int fam) {
    return fam == FAM_UNGAPPED_P2D || fam == FAM_PROTEIN2DNA || fam == FAM_PROTEIN2GENOME ||
           (fam >= FAM_PROTEIN2DNA_START && fam <= FAM_PROTEIN2GENOME_JOIN) || family_is_p2g_span(fam);
}
bool family_has_splice(int fam) {
    return fam == FAM_EST2GENOME || fam == FAM_PROTEIN2GENOME || (fam >= FAM_EST2GENOME_FWD_START && fam <= FAM_EST2GENOME_REV_JOIN) ||
           (fam >= FAM_PROTEIN2GENOME_START && fam <= FAM_PROTEIN2GENOME_JOIN) ||
           (fam >= FAM_EST2GENOME_FWD_SPAN_SRC && fam <= FAM_EST2GENOME_REV_SPAN_DST) || family_is_p2g_span(fam);
}
bool family_has_phase(int fam) {
    return fam == FAM_PROTEIN2GENOME || (fam >= FAM_PROTEIN2GENOME_START && fam <= FAM_PROTEIN2GENOME_JOIN) ||
           family_is_p2g_span(fam);
}

// ---- sequence preparation kernels -------------------------------------------------------------------------
struct PrepTables {
    uint8_t submat_index[256];
    uint8_t nt2d[256];
    uint8_t trans[4096];
    uint8_t aa[40];
};

// residue bytes -> substitution matrix row (Submat_lookup's index step, submat.h:54-56)
// (codes, when given: the set of rows that occur, one bit each -- what the staged packed score pass sizes its query profile by)
__global__ void encode_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long n,
                              const PrepTables *__restrict__ tab, int *bad, int *codes = nullptr) {
    int seen = 0;
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const uint8_t c = tab->submat_index[in[x]];
        if (c >= 24) atomicExch(bad, 1);
        out[x] = c >= 24 ? 0 : c;
        seen |= 1 << (c >= 24 ? 0 : c);
    }
    if (codes) {
        for (int off = 32; off > 0; off >>= 1) seen |= __shfl_xor(seen, off);
        if ((threadIdx.x & 63) == 0 && seen) atomicOr(codes, seen);
    }
}

// the targets' row codes as DENSE indices (0 .. 7) into the batch's code table (tab[code] -> index): what the packed checkpoint
// pass and region windows index their query profiles by (Prof16, c4_ckpt16_kernel.h)
__global__ void dense_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, long long n, const uint8_t *__restrict__ tab) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const uint8_t d = tab[in[x]];
        out[x] = d < 8 ? d : 0;
    }
}

// protein2dna target: row of the residue encoded by the codon starting at each position
// (Translate_base, translate.h:73-76, then the submat index)
__global__ void codon_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const long long *off,
                             const int *len, int n_seqs, const PrepTables *__restrict__ tab, int *bad) {
  for (int pair = blockIdx.y; pair < n_seqs; pair += gridDim.y) {
    const uint8_t *s = in + off[pair];
    uint8_t *o = out + off[pair];
    const int n = len[pair];
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
        uint8_t code = 0;
        if (x + 2 < n) {
            const uint8_t aa = tab->aa[tab->trans[tab->nt2d[s[x]] | (tab->nt2d[s[x + 1]] << 4) | (tab->nt2d[s[x + 2]] << 8)]];
            code = tab->submat_index[aa];
            if (code >= 24) { atomicExch(bad, 2); code = 0; }
        }
        o[x] = code;
    }
  }
}

// split-codon calcs (phase.c:188-208) re-read bases around an intron: per position the 4-bit base masks
// (Translate nt2d, translate.h:40-50) of positions p, p-1, p-2, p-3
__global__ void tn4_kernel(const uint8_t *__restrict__ in, uint16_t *__restrict__ out, const long long *off,
                           const int *len, int n_seqs, const PrepTables *__restrict__ tab) {
  for (int pair = blockIdx.y; pair < n_seqs; pair += gridDim.y) {
    const uint8_t *s = in + off[pair];
    uint16_t *o = out + off[pair];
    const int n = len[pair];
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
        unsigned v = 0;
        for (int d = 0; d < 4; d++)
            if (x - d >= 0) v |= (unsigned)tab->nt2d[s[x - d]] << (4 * d);
        o[x] = (uint16_t)v;
    }
  }
}

// SplicePredictor_predict_array_int (splice.c:383-397): float accumulation left to right over the PSSM
// window clipped to the sequence (Splice_predict_position, splice.c:320-344), rounded half away from
// zero in double (SplicePredictor_round, splice.c:379-381).  Plain adds only: no contraction possible.
__global__ void splice_kernel(const uint8_t *__restrict__ seq, const long long *off, const int *len, int n_seqs,
                              const c4gpu_splice_model *__restrict__ models, int *__restrict__ out,
                              long long stride) {
  const int type = blockIdx.z;
  const c4gpu_splice_model *sp = &models[type];
  for (int pair = blockIdx.y; pair < n_seqs; pair += gridDim.y) {
    const uint8_t *s = seq + off[pair];
    const int n = len[pair];
    int *o = out + (long long)type * stride + off[pair];
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += gridDim.x * blockDim.x) {
        int seq_start = pos - sp->splice_after, model_start = 0, calc_length = sp->model_length;
        if (seq_start < 0) { model_start = -seq_start; seq_start = 0; calc_length -= model_start; }
        if (seq_start + calc_length > n) calc_length = n - seq_start;
        float score = 0.0f;
        for (int i = 0; i < calc_length; i++) score = score + sp->data[model_start + i][sp->index[s[seq_start + i]]];
        if (sp->gtag_only) {                    // Splice_predict_is_on_GTAG, splice.c:312-318 (past the end: the NUL)
            const int b1 = s[pos], b2 = pos + 1 < n ? s[pos + 1] : 0;
            const int u1 = (b1 >= 'a' && b1 <= 'z') ? b1 - 32 : b1, u2 = (b2 >= 'a' && b2 <= 'z') ? b2 - 32 : b2;
            if (u1 != sp->expect_one || u2 != sp->expect_two) score = -987654321.0f;
        }
        const double r = score < 0 ? (double)score - 0.5 : (double)score + 0.5;
        o[pos] = (int)r;
    }
  }
}

// The same arrays, tiled: a workgroup takes 1 024 consecutive positions of one sequence and computes all four site types for
// them.  The four PSSMs, the residue -> column tables and the tile's residues (with the columns of every model looked up once)
// sit in LDS; a thread owns four consecutive positions and slides a four-byte window of columns along the model, so one LDS
// byte read serves four positions per model row; the four values of a position leave as 16-byte stores.  Every position's sum
// is the same chain of float adds in the same order as above (no reassociation), so the arrays are bit-identical; positions
// whose window is clipped by an end of the sequence take the one-by-one loop.  With `out16` (est2genome batches whose
// parameters allow the packed passes) the kernel also writes the packed passes' splice array -- the four values clamped to 16
// bits with the calc constant of a pre-splice transition folded in (fold[type]; ss16_kernel's formula) -- instead of a second
// pass that reads the 16 bytes per position back.
struct SpliceFold { int add[4]; };
constexpr int SPLICE_TILE = 1024, SPLICE_HALO = C4GPU_SPLICE_MAX_LEN;
__global__ __launch_bounds__(256) void splice_tile_kernel(const uint8_t *__restrict__ seq, const long long *off, const int *len, int n_seqs,
                                                          const c4gpu_splice_model *__restrict__ models, int *__restrict__ out,
                                                          long long stride, SpliceFold fold, uint2 *__restrict__ out16) {
    __shared__ float sdata[4][C4GPU_SPLICE_MAX_LEN * 5];
    __shared__ uint8_t sindex[4][256];
    __shared__ __attribute__((aligned(16))) uint8_t scol[4][SPLICE_TILE + 2 * SPLICE_HALO + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sraw[SPLICE_TILE + 16];
    __shared__ int s_mlen[4], s_after[4], s_gtag[4], s_e1[4], s_e2[4];
    const int tid = threadIdx.x;
    for (int x = tid; x < 4 * C4GPU_SPLICE_MAX_LEN * 5; x += 256) sdata[x / (C4GPU_SPLICE_MAX_LEN * 5)][x % (C4GPU_SPLICE_MAX_LEN * 5)] =
        models[x / (C4GPU_SPLICE_MAX_LEN * 5)].data[(x % (C4GPU_SPLICE_MAX_LEN * 5)) / 5][x % 5];
    for (int x = tid; x < 4 * 256; x += 256) sindex[x >> 8][x & 255] = models[x >> 8].index[x & 255];
    if (tid < 4) {
        s_mlen[tid] = models[tid].model_length; s_after[tid] = models[tid].splice_after; s_gtag[tid] = models[tid].gtag_only;
        s_e1[tid] = models[tid].expect_one; s_e2[tid] = models[tid].expect_two;
    }
    __syncthreads();
    for (int pair = blockIdx.y; pair < n_seqs; pair += gridDim.y) {
        const uint8_t *s = seq + off[pair];
        const int n = len[pair];
        for (int tile = blockIdx.x * SPLICE_TILE; tile < n; tile += gridDim.x * SPLICE_TILE) {
            __syncthreads();                                   // the tile before has been read
            // residues tile - HALO .. tile + TILE + HALO as model columns (outside the sequence: never read), tile .. tile + TILE + 1 raw
            const int lo = tile - SPLICE_HALO;
            for (int x = tid; x < SPLICE_TILE + 2 * SPLICE_HALO; x += 256) {
                const int p = lo + x;
                const uint8_t b = (p >= 0 && p < n) ? s[p] : 0;
                for (int k = 0; k < 4; k++) scol[k][x] = sindex[k][b];
            }
            for (int x = tid; x < SPLICE_TILE + 2; x += 256) sraw[x] = (tile + x < n) ? s[tile + x] : 0;
            __syncthreads();
            const int p0 = tile + 4 * tid;                     // this thread's positions p0 .. p0 + 3
            if (p0 >= n) continue;
            int v[4][4];                                       // [type][position]
            for (int k = 0; k < 4; k++) {
                const int mlen = s_mlen[k], after = s_after[k];
                const float *d = sdata[k];
                const uint8_t *c = scol[k] + (p0 - after - lo);     // column of residue p0 - after + i at c[i]
                float sc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (p0 - after >= 0 && p0 + 3 - after + mlen <= n) {
                    unsigned w = (unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16);
                    for (int i = 0; i < mlen; i++) {
                        w |= (unsigned)c[i + 3] << 24;
                        const float *row = d + 5 * i;
                        sc[0] = sc[0] + row[w & 0xff];
                        sc[1] = sc[1] + row[(w >> 8) & 0xff];
                        sc[2] = sc[2] + row[(w >> 16) & 0xff];
                        sc[3] = sc[3] + row[w >> 24];
                        w >>= 8;
                    }
                } else {
                    for (int q = 0; q < 4; q++) {
                        const int pos = p0 + q;
                        if (pos >= n) break;
                        int seq_start = pos - after, model_start = 0, calc_length = mlen;
                        if (seq_start < 0) { model_start = -seq_start; seq_start = 0; calc_length -= model_start; }
                        if (seq_start + calc_length > n) calc_length = n - seq_start;
                        float score = 0.0f;
                        for (int i = 0; i < calc_length; i++) score = score + d[5 * (model_start + i) + scol[k][seq_start + i - lo]];
                        sc[q] = score;
                    }
                }
                for (int q = 0; q < 4; q++) {
                    float score = sc[q];
                    if (s_gtag[k]) {                            // Splice_predict_is_on_GTAG, splice.c:312-318 (past the end: the NUL)
                        const int pos = p0 + q;
                        const int b1 = sraw[4 * tid + q], b2 = pos + 1 < n ? sraw[4 * tid + q + 1] : 0;
                        const int u1 = (b1 >= 'a' && b1 <= 'z') ? b1 - 32 : b1, u2 = (b2 >= 'a' && b2 <= 'z') ? b2 - 32 : b2;
                        if (u1 != s_e1[k] || u2 != s_e2[k]) score = -987654321.0f;
                    }
                    const double r = score < 0 ? (double)score - 0.5 : (double)score + 0.5;
                    v[k][q] = (int)r;
                }
            }
            const long long base = off[pair] + p0;               // sequences start at multiples of four: 16-byte aligned
            if (p0 + 3 < n) {
                for (int k = 0; k < 4; k++)
                    *reinterpret_cast<int4 *>(out + (long long)k * stride + base) = make_int4(v[k][0], v[k][1], v[k][2], v[k][3]);
            } else {
                for (int k = 0; k < 4; k++)
                    for (int q = 0; q < 4 && p0 + q < n; q++) out[(long long)k * stride + base + q] = v[k][q];
            }
            if (out16) {
                auto c16 = [](int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); };
                for (int q = 0; q < 4 && p0 + q < n; q++) {
                    uint2 o;
                    o.x = ((unsigned)c16(fold.add[0] + v[0][q]) & 0xffffu) | ((unsigned)c16(fold.add[1] + v[1][q]) << 16);
                    o.y = ((unsigned)c16(fold.add[2] + v[2][q]) & 0xffffu) | ((unsigned)c16(fold.add[3] + v[3][q]) << 16);
                    out16[base + q] = o;
                }
            }
        }
    }
}

// ---- HSP seeding: the ungapped X-drop extension of HSPset_seed_hsp (src/comparison/hspset.c:933-997) ----------------
// One lane per seed: HSP_trim_ends (:837-870), HSP_init (:722-741), HSP_extend without masking (:743-812: left, then
// right, best extension so far, stop below zero or `dropoff` under the best) and HSP_find_cobs (:426-441).  Residues
// are read through the batch's coded arrays (qcode / tcode = substitution matrix rows; for PROTEIN2DNA tcode holds the
// row of the codon starting at each position), the matrix sits in LDS.  Seeds of one diagonal neighbourhood read the
// same cache lines; the work per seed is a few hundred bytes, so the kernel is latency- and not bandwidth-bound.
struct HspJob { long long qoff, toff; int qlen, tlen; };
__device__ __forceinline__ c4gpu_hsp hsp_extend_one(const uint8_t *__restrict__ qcode, const uint8_t *__restrict__ tcode, const HspJob jb,
                                                     const c4gpu_hsp_seed sd, const int *sm, int aq, int at, int seedlen, int dropoff) {
        const uint8_t *q = qcode + jb.qoff, *t = tcode + jb.toff;
        auto sc = [&](int qp, int tp) { return sm[q[qp] * 24 + t[tp]]; };
        int qs = sd.query_start, ts = sd.target_start, length = seedlen, i;
        for (i = 0; i < length; i++) {                                   // HSP_trim_ends
            if (sc(qs, ts) > 0) break;
            qs += aq; ts += at;
        }
        length -= i;
        int qp = qs + length * aq - aq, tp = ts + length * at - at;
        while (length > 0) {
            if (sc(qp, tp) > 0) break;
            length--; qp -= aq; tp -= at;
        }
        int score = 0;                                                   // HSP_init
        for (i = 0, qp = qs, tp = ts; i < length; i++, qp += aq, tp += at) score += sc(qp, tp);
        int maxscore = score, extend, maxext;                            // HSP_extend: left
        qp = qs - aq; tp = ts - at;
        for (extend = 1, maxext = 0; qp >= 0 && tp >= 0; extend++) {
            score += sc(qp, tp);
            if (maxscore <= score) { maxscore = score; maxext = extend; }
            else { if (score < 0) break; if (maxscore - score >= dropoff) break; }
            qp -= aq; tp -= at;
        }
        qp = qs + length * aq; tp = ts + length * at;
        qs -= maxext * aq; ts -= maxext * at; length += maxext;
        score = maxscore;
        for (extend = 1, maxext = 0; qp + aq <= jb.qlen && tp + at <= jb.tlen; extend++) {     // right
            score += sc(qp, tp);
            if (maxscore <= score) { maxscore = score; maxext = extend; }
            else { if (score < 0) break; if (maxscore - score >= dropoff) break; }
            qp += aq; tp += at;
        }
        length += maxext;
        score = 0;                                                       // HSP_find_cobs
        for (i = 0, qp = qs, tp = ts; i < length; i++, qp += aq, tp += at) {
            score += sc(qp, tp);
            if (score >= (maxscore >> 1)) break;
        }
        return c4gpu_hsp{qs, ts, length, maxscore, i};
}

__global__ void hsp_extend_kernel(const uint8_t *__restrict__ qcode, const uint8_t *__restrict__ tcode,
                                  const HspJob *__restrict__ jobs, const c4gpu_hsp_seed *__restrict__ seeds, int n_seeds,
                                  const int *__restrict__ submat, int aq, int at, int seedlen, int dropoff,
                                  c4gpu_hsp *__restrict__ out) {
    __shared__ int sm[24 * 24];
    for (int x = threadIdx.x; x < 24 * 24; x += blockDim.x) sm[x] = submat[x];
    __syncthreads();
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_seeds; k += gridDim.x * blockDim.x)
        out[k] = hsp_extend_one(qcode, tcode, jobs[seeds[k].pair], seeds[k], sm, aq, at, seedlen, dropoff);
}

// one lane per horizon chain (c4gpu_hsp_extend_chains): its seeds in order, skipped while below the running horizon
__global__ void hsp_chain_kernel(const uint8_t *__restrict__ qcode, const uint8_t *__restrict__ tcode,
                                 const HspJob *__restrict__ jobs, const c4gpu_hsp_seed *__restrict__ seeds,
                                 const int *__restrict__ order, const int *__restrict__ chain_first, int n_chains,
                                 const int *__restrict__ horizon0, const int *__restrict__ submat, int aq, int at, int seedlen,
                                 int dropoff, c4gpu_hsp *__restrict__ out) {
    __shared__ int sm[24 * 24];
    for (int x = threadIdx.x; x < 24 * 24; x += blockDim.x) sm[x] = submat[x];
    __syncthreads();
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_chains; c += gridDim.x * blockDim.x) {
        int horizon = horizon0[c];
        for (int x = chain_first[c]; x < chain_first[c + 1]; x++) {
            const int k = order[x];
            if (seeds[k].target_start < horizon) { out[k] = c4gpu_hsp{0, 0, -1, 0, 0}; continue; }      // hspset.c:952-958
            const c4gpu_hsp h = hsp_extend_one(qcode, tcode, jobs[seeds[k].pair], seeds[k], sm, aq, at, seedlen, dropoff);
            out[k] = h;
            horizon = h.target_start + h.length * at;                                                   // HSP_target_end, :990
        }
    }
}

// ---- resident sequences of a batch --------------------------------------------------------------------------
// Column entries of the blocked-cell lists (the device form of SubOpt_Index's rows, subopt.c:250-333): for
// every job and every column 0..T+1 two ints: the first blocked row of the column (or SUB_NONE), and twice
// the index of the first point at or after that column, plus 1 when the column holds more than one point.
__global__ void subopt_colptr_kernel(const DevJob *jobs, int n_jobs, const int *pts_t, const int *pts_q, int *colent) {
    // The points are sorted by column, so point k owns the columns after its predecessor's up to its own:
    // one pass of T+2 writes per job, no searches.
    constexpr int SUB_NONE = -0x40000000;
    for (int x = blockIdx.x; x < n_jobs; x += gridDim.x) {
        const DevJob &j = jobs[x];
        const int *pt = pts_t + j.sub_pt_off;
        const int *pq = pts_q + j.sub_pt_off;
        const int n = j.sub_pt_n;
        for (int k = threadIdx.x; k <= n; k += blockDim.x) {
            const int first = k == 0 ? 0 : pt[k - 1] + 1;
            const int last = k == n ? j.T + 1 : pt[k];          // inclusive
            const int idx2 = 2 * (j.sub_pt_off + k);
            for (int c = first; c <= last; c++) {
                const bool own = k < n && c == last;            // the column of point k itself
                const bool more = own && k + 1 < n && pt[k + 1] == c;
                int *e = colent + 2 * (j.sub_off + c);
                e[0] = own ? pq[k] : SUB_NONE;
                e[1] = idx2 | (more ? 1 : 0);
            }
        }
    }
}

// ---- sub-alignments of a checkpoint pass, listed and stitched on the device -------------------------------------
// Optimal_find_path_reduced_space (optimal.c:160-230) turns the checkpoint traceback of a region into a list of
// Viterbi_SubAlignments and Optimal_compute_subalignments (optimal.c:266-313) runs one FIND_PATH continuation per entry,
// each seeded with the final cell of the one before it.  A batch of 4 096 pairs of 1 kb x 1 kb under -D 32 has 778 443 of
// them: describing them on the host, uploading the descriptions and unpacking as many results was two thirds of a pass.
// The three kernels below keep that list on the device: the checkpoint kernel's DevVsa records become DevJobs
// (expand), the path kernel runs them, and one thread per pair checks every predicted final cell against the computed
// one and concatenates the runs in path order with Alignment_add's merge rule (stitch).  Pairs the fast route cannot
// finish (a section that itself needs checkpoints, a final cell that differs from its prediction, no END) are flagged
// and take the host route of find_path_batch.
enum { FUSE_MAX_OPS_CAP = 0, FUSE_MAX_TB, FUSE_MAX_T, FUSE_MAX_STRIPS, FUSE_OPS_TOTAL, FUSE_STATS };

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one workgroup (one wave) per checkpoint job x; its children are out[first[x]] .. out[first[x + 1] - 1] in path order
__global__ __launch_bounds__(64) void fuse_expand_kernel(const DevJob *parents, const DevVsa *vsa, const int *first, int n_parents,
                                                         DevJob *out, c4h::MemRule rule, int dpmemory_mb, int path_R,
                                                         int *flags, unsigned long long *stats, int keep_root) {
    const int x = blockIdx.x;
    if (x >= n_parents) return;
    const DevJob &pj = parents[x];
    const int base = first[x], cnt = first[x + 1] - base;
    unsigned long long m_cap = 0, m_tb = 0, m_T = 0, m_strips = 0, ops = 0;
    bool nested = false;
    for (int k = threadIdx.x; k < cnt; k += 64) {
        const DevVsa &dv = vsa[pj.vsa_off + (cnt - 1 - k)];              // the list is last section first
        DevJob j;
        memset(&j, 0, sizeof j);
        j.pair = pj.pair; j.q0 = dv.qs; j.t0 = dv.ts; j.Q = dv.ql; j.T = dv.tl;
        j.root = keep_root ? pj.root : 0;    // the state the alignment's END is entered from (0: not known): BYROOT path kernels
        j.first_state = dv.first_state;
        // optimal.c:204-213,283-301: first cell = final cell of the sub-alignment before it (the parent's first cell for
        // the first one), final state = first state of the next one (the parent's final state for the last one)
        j.final_state = (k + 1 < cnt) ? vsa[pj.vsa_off + (cnt - 2 - k)].first_state : pj.final_state;
        const int *fc = k > 0 ? vsa[pj.vsa_off + (cnt - k)].final_cell : pj.first_cell;
        for (int l = 0; l < CELL_MAX; l++) j.first_cell[l] = fc[l];
        int tb = 0;
        while ((1LL << tb) <= j.T) tb++;
        j.tshift = tb;
        j.ckpt_off = -1; j.seed_off = -1;
        j.ops_cap = 3 * (j.Q + j.T) + 16;
        out[base + k] = j;
        nested |= c4h::use_reduced_space(rule, dv.ql, dv.tl, dpmemory_mb);
        const unsigned long long strips = (unsigned long long)(j.Q + 1 + 64 * path_R - 1) / (unsigned long long)(64 * path_R);
        const unsigned long long tbw = strips * (unsigned long long)(j.T + 64) * 64ull * (unsigned long long)path_R;
        m_cap = m_cap > (unsigned long long)j.ops_cap ? m_cap : (unsigned long long)j.ops_cap;
        m_tb = m_tb > tbw ? m_tb : tbw;
        m_T = m_T > (unsigned long long)j.T ? m_T : (unsigned long long)j.T;
        m_strips = m_strips > strips ? m_strips : strips;
        ops += (unsigned long long)j.ops_cap;
    }
    m_cap = wave_max_u64(m_cap); m_tb = wave_max_u64(m_tb); m_T = wave_max_u64(m_T); m_strips = wave_max_u64(m_strips);
    ops = wave_sum_u64(ops);
    const bool any_nested = __builtin_amdgcn_ballot_w64(nested) != 0;
    if (threadIdx.x == 0) {
        if (any_nested) flags[x] = 1;
        atomicMax(&stats[FUSE_MAX_OPS_CAP], m_cap); atomicMax(&stats[FUSE_MAX_TB], m_tb); atomicMax(&stats[FUSE_MAX_T], m_T);
        atomicMax(&stats[FUSE_MAX_STRIPS], m_strips); atomicAdd(&stats[FUSE_OPS_TOTAL], ops);
    }
}

// c4gpu_memrule_device: the rule on the device, one size per thread
__global__ void memrule_probe_kernel(c4h::MemRule rule, int dpmemory_mb, const int *ql, const int *tl, int n, int *reduced, int *rows) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    reduced[x] = c4h::use_reduced_space(rule, ql[x], tl[x], dpmemory_mb) ? 1 : 0;
    // the reference divides by the row size (viterbi.c:209): only defined where that size did not overflow
    rows[x] = c4h::viterbi_row_bytes(rule, ql[x], 1 + rule.total_shadow_designations + 1) ? c4h::checkpoint_rows(rule, ql[x], tl[x], dpmemory_mb) : -1;
}

struct FusePair { long long off; int count, status; };      // merged (transition, length) pairs at out[2 * off ..]; status 0 = done

// Is the final cell a sub-alignment computed the one its successor was seeded with?  Exactly, except for cells predicted by the
// packed checkpoint pass (c4_ckpt16_kernel.h), whose intron-length counter saturates: a shadow slot (the target position an
// open intron started at) that lies 32 767 columns or more behind the cell is known there only as "that far back", and two
// such positions are interchangeable — the one thing ever computed from a shadow is the intron's length at its 3' site
// (intron.c:150-160), which passes the minimum either way and cannot exceed the maximum (Engine::pk16_fits).
//
// The SCORE of the cell is not compared (strict_score = false, the default; C4GPU_CELL_STRICT=1 compares it: the form of rounds
// 1-4).  A continuation sub-DP starts from ONE cell of ONE state (viterbi.c:705-714) and every score in it is that cell's
// score plus calcs along a path from it: max-plus is translation invariant, so a first cell whose score differs by d gives the
// same winners, the same ties, the same traceback and a final cell whose score differs by d and whose shadows are the same.
// The reference threads the computed cell through (optimal.c:283,301) and never looks at the checkpoint pass's score again;
// where intron length limits break optimal substructure the two differ by a few points (pair 3 775 of the all-against-all
// batch: 74 computed, 71 predicted, same intron start) -- the paths of every later sub-alignment are the ones the batch
// computed.  What does decide later cells are the shadow slots, and those are compared.  (Unset states hold -987654321
// whatever d is; a real candidate beats them by ~10^9 either way.)
__host__ __device__ inline bool final_cell_equiv(const int *computed, const int *predicted, int n_slots, int target_end, bool packed,
                                                 bool strict_score = false) {
    if (strict_score && computed[0] != predicted[0]) return false;
    for (int l = 1; l < n_slots; l++) {
        if (computed[l] == predicted[l]) continue;
        if (!(packed && (long long)target_end - predicted[l] - 2 >= 32767 && (long long)target_end - computed[l] - 2 >= 32767)) return false;
    }
    return true;
}

inline bool cell_strict() { return c4cfg::nonzero(c4cfg::CELL_STRICT); }

// one thread per checkpoint job: verify the chain of final cells, then Alignment_add (alignment.c:75-102) over the runs
// of its sub-alignments in path order (each job's walk wrote its runs END -> START)
__global__ void fuse_stitch_kernel(const DevJob *parents, const DevVsa *vsa, const int *first, int n_parents,
                                   const DevResult *sub, const uint32_t *runs, int path_cs, const int *flags,
                                   unsigned long long *out_used, int *out, FusePair *pairs, int n_packed, int strict_score) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_parents) return;
    FusePair fp; fp.off = 0; fp.count = 0; fp.status = 1;
    const int base = first[x], cnt = first[x + 1] - base;
    if (flags[x] || cnt == 0) { pairs[x] = fp; return; }
    const DevJob &pj = parents[x];
    long long total = 0;
    bool bad = false;
    for (int k = 0; k < cnt; k++) {
        const DevResult &r = sub[base + k];
        if (r.flags & (FLAG_OPS_OVERFLOW | FLAG_NO_END)) { bad = true; break; }
        total += r.n_ops;
        // the next sub-alignment was seeded with the predicted cell: it must be the one this one produced
        if (k + 1 < cnt) {
            const DevVsa &dv = vsa[pj.vsa_off + (cnt - 1 - k)];
            bad |= !final_cell_equiv(r.final_cell, dv.final_cell, path_cs, dv.ts + dv.tl, x < n_packed, strict_score != 0);
            if (bad) break;
        }
    }
    if (bad) { pairs[x] = fp; return; }
    const long long off = (long long)atomicAdd(out_used, (unsigned long long)total);
    int *o = out + 2 * off;
    int n = 0;
    for (int k = 0; k < cnt; k++) {
        const DevResult &r = sub[base + k];
        const uint32_t *w = runs + r.ops_off;
        for (int q = r.n_ops - 1; q >= 0; q--) {
            const int tr = (int)(w[q] >> 24), len = (int)(w[q] & 0xffffff);
            if (n && o[2 * (n - 1)] == tr) {
                o[2 * (n - 1) + 1] += len;
                if (o[2 * (n - 1) + 1] == 0) n--;
            } else {
                o[2 * n] = tr; o[2 * n + 1] = len; n++;
            }
        }
    }
    fp.off = off; fp.count = n; fp.status = 0;
    pairs[x] = fp;
}

struct ResidentSeqs {
    int n_pairs = 0;
    std::vector<long long> qoff, toff;
    std::vector<int> qlen, tlen;
    long long total_q = 0, total_t = 0;
    DevBuf<uint8_t> qraw, traw, qcode, tcode;
    DevBuf<long long> d_qoff, d_toff;
    DevBuf<int> d_qlen, d_tlen, ss;
    DevBuf<uint16_t> tn4;
    mutable DevBuf<uint2> ss16;            // the packed score pass's splice values (built on its first launch over this batch)
    mutable bool ss16_built = false;
    mutable std::mutex ss16_lock;          // two lanes may reach the first packed launch together
    // the residue codes the targets hold, for the staged packed score pass (c4_viterbi16_kernel.h, IO 1): [0, 24) code -> dense
    // index (0xff: absent), [24, 32) dense index -> code; tdense_n = 0: not known (codon-coded targets)
    DevBuf<uint8_t> tdense;
    int tdense_n = 0;
    DevBuf<uint8_t> tcode_dense;          // the targets as dense indices into that table (dense_kernel); with tdense_n in 1 .. 8
    long long ss_len = 0;                 // positions per splice array
    DevBuf<PrepTables> tables;
    DevBuf<c4gpu_splice_model> splice_models;
    DevBuf<int> bad;
    DevSeqs dev;

    // Pairs that hand over the same host buffer (same pointer and length: one genomic contig against many
    // queries, all-vs-all front ends) share one device copy and one set of derived arrays.
    int n_utargets = 0;
    DevBuf<long long> d_utoff;
    DevBuf<int> d_utlen;

    // page-locked staging of the residues, kept between the batches of a c4gpu_stage (build with pin = true)
    PinBuf pin_q, pin_t;

    int build(c4gpu_ctx *ctx, int family, const c4gpu_params *params, const c4gpu_pair *pairs, int n, bool pin = false,
              const SpliceFold *fold16 = nullptr) {
        const bool trace = c4cfg::has(c4cfg::TRACE);
        const auto t_begin = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (trace) fprintf(stderr, "c4gpu trace: staging: %-24s at %.3f ms\n", what,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
        };
        n_pairs = n;
        qoff.resize(n); toff.resize(n); qlen.resize(n); tlen.resize(n);
        total_q = total_t = 0;
        std::map<std::pair<const uint8_t *, int>, long long> qseen, tseen;
        std::vector<int> uq, ut;                      // first pair that holds each unique sequence
        std::vector<long long> utoff;
        std::vector<int> utlen;
        for (int i = 0; i < n; i++) {
            qlen[i] = pairs[i].query_len; tlen[i] = pairs[i].target_len;
            if (qlen[i] < 0 || tlen[i] < 0 || tlen[i] >= (1 << 30) || qlen[i] >= (1 << 30)) {
                c4h::set_error("sequence length outside [0, 2^30): the kernels address a sequence with 32-bit byte offsets");
                return -1;
            }
            auto qk = std::make_pair(pairs[i].query, qlen[i]);
            auto qi = qseen.find(qk);
            if (qi == qseen.end()) {
                qseen[qk] = total_q; qoff[i] = total_q; uq.push_back(i);
                total_q += (qlen[i] + 3) & ~3LL;
            } else qoff[i] = qi->second;
            auto tk = std::make_pair(pairs[i].target, tlen[i]);
            auto ti = tseen.find(tk);
            if (ti == tseen.end()) {
                tseen[tk] = total_t; toff[i] = total_t; ut.push_back(i);
                utoff.push_back(total_t); utlen.push_back(tlen[i]);
                total_t += (tlen[i] + 3) & ~3LL;
            } else toff[i] = ti->second;
        }
        n_utargets = (int)ut.size();
        // one host buffer per side, every byte written exactly once (residues, 'A' in the gaps up to the next multiple
        // of four and in the 64-byte tail the kernels may read into) by several threads: a value-initialised vector of
        // this size (410 MB of targets for the north-star batch) is page-faulted in and written twice on one core.
        // A stage (c4gpu_stage: `pin` set) gathers into page-locked buffers it keeps between batches, slice by slice, each
        // slice on its way over the link while the next one is gathered.
        const size_t hq_n = (size_t)total_q + 64, ht_n = (size_t)total_t + 64;
        std::unique_ptr<uint8_t[]> own_q, own_t;
        uint8_t *hq = nullptr, *ht = nullptr;
        if (pin) {
            if (pin_q.reserve(hq_n) || pin_t.reserve(ht_n)) return -1;
            hq = pin_q.p; ht = pin_t.p;
        } else {
            own_q.reset(new uint8_t[hq_n]); own_t.reset(new uint8_t[ht_n]);
            hq = own_q.get(); ht = own_t.get();
        }
        hipStream_t s = ctx->stream;
        if (qraw.alloc(hq_n) || traw.alloc(ht_n)) return -1;
        // gathers the unique sequences [ua, ub) of one side; returns the byte range they cover
        auto gather = [&](uint8_t *dst, const std::vector<int> &uniq, bool query, size_t ua, size_t ub) {
            parallel_for((long long)(ub - ua), 16, [&](long long first, long long last) {
                for (long long x = (long long)ua + first; x < (long long)ua + last; x++) {
                    const int i = uniq[x];
                    const int len = query ? qlen[i] : tlen[i];
                    uint8_t *d = dst + (query ? qoff[i] : toff[i]);
                    if (len) memcpy(d, query ? pairs[i].query : pairs[i].target, len);
                    for (int k = len; k < ((len + 3) & ~3); k++) d[k] = 'A';
                }
            });
        };
        auto side = [&](uint8_t *host, uint8_t *devp, const std::vector<int> &uniq, bool query, long long total, size_t host_n) -> int {
            const size_t slice = pin ? (size_t)32 << 20 : ~(size_t)0;          // bytes per slice
            size_t ua = 0;
            long long sent = 0;
            while (ua < uniq.size()) {
                size_t ub = ua;
                long long end = sent;
                while (ub < uniq.size() && (size_t)(end - sent) < slice) {
                    const int i = uniq[ub++];
                    end = (query ? qoff[i] : toff[i]) + (((query ? qlen[i] : tlen[i]) + 3) & ~3LL);
                }
                gather(host, uniq, query, ua, ub);
                if (pin && end > sent) HIP_OK(hipMemcpyAsync(devp + sent, host + sent, (size_t)(end - sent), hipMemcpyHostToDevice, s));
                sent = end; ua = ub;
            }
            memset(host + total, 'A', 64);
            if (pin) HIP_OK(hipMemcpyAsync(devp + total, host + total, 64, hipMemcpyHostToDevice, s));
            else HIP_OK(hipMemcpyAsync(devp, host, host_n, hipMemcpyHostToDevice, s));
            return 0;
        };
        if (side(hq, qraw.p, uq, true, total_q, hq_n) || side(ht, traw.p, ut, false, total_t, ht_n)) return -1;
        lap("sequences gathered");
        PrepTables pt;
        memcpy(pt.submat_index, params->submat_index, 256);
        memcpy(pt.nt2d, params->nt2d, 256);
        memcpy(pt.trans, params->trans, 4096);
        memcpy(pt.aa, params->aa, 40);
        if (tables.upload(&pt, 1, s) ||
            qcode.alloc(hq_n) || tcode.alloc(ht_n) || d_qoff.upload(qoff.data(), n, s) ||
            d_toff.upload(toff.data(), n, s) || d_qlen.upload(qlen.data(), n, s) || d_tlen.upload(tlen.data(), n, s) ||
            d_utoff.upload(utoff.data(), n_utargets, s) || d_utlen.upload(utlen.data(), n_utargets, s))
            return -1;
        if (trace) { HIP_OK(c4_stream_sync(s)); lap("uploaded"); }
        int zero[2] = {0, 0};
        if (bad.upload(zero, 2, s)) return -1;
        const int blocks = 1024;
        hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), 0, s, qraw.p, qcode.p, (long long)hq_n, tables.p, bad.p);
        int max_t = 1;
        for (int i = 0; i < n; i++) max_t = std::max(max_t, tlen[i]);
        const int xb = std::min(256, (max_t + 255) / 256), yb = std::max(1, std::min(n_utargets, 32768));
        if (family_is_p2d(family)) {
            hipLaunchKernelGGL(codon_kernel, dim3(xb, yb), dim3(256), 0, s, traw.p, tcode.p, d_utoff.p, d_utlen.p, n_utargets, tables.p, bad.p);
        } else {
            hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), 0, s, traw.p, tcode.p, (long long)ht_n, tables.p, bad.p, bad.p + 1);
        }
        dev.ss = nullptr;
        dev.ss16 = nullptr;
        dev.ss_stride = 0;
        ss16_built = false;
        if (family_has_splice(family)) {
            if (splice_models.upload(params->splice, 4, s) || ss.alloc((size_t)4 * ht_n)) return -1;
            bool tiled = !(c4cfg::is(c4cfg::SPLICE_TILE, 0));     // 0: the one-position-per-thread kernel
            for (int k = 0; k < 4; k++)
                tiled = tiled && params->splice[k].splice_after >= 0 && params->splice[k].splice_after <= SPLICE_HALO &&
                        params->splice[k].model_length >= 0 && params->splice[k].model_length <= C4GPU_SPLICE_MAX_LEN;
            if (tiled) {
                if (fold16 && ss16.alloc(ht_n)) return -1;
                const int tiles = std::max(1, std::min(4096, (max_t + SPLICE_TILE - 1) / SPLICE_TILE));
                hipLaunchKernelGGL(splice_tile_kernel, dim3(tiles, yb), dim3(256), 0, s, traw.p, d_utoff.p, d_utlen.p, n_utargets,
                                   splice_models.p, ss.p, (long long)ht_n, fold16 ? *fold16 : SpliceFold{{0, 0, 0, 0}},
                                   fold16 ? ss16.p : (uint2 *)nullptr);
                if (fold16) { ss16_built = true; dev.ss16 = nullptr; }
            } else
            hipLaunchKernelGGL(splice_kernel, dim3(xb, yb, 4), dim3(256), 0, s, traw.p, d_utoff.p, d_utlen.p, n_utargets,
                               splice_models.p, ss.p, (long long)ht_n);
            dev.ss = ss.p;
            dev.ss_stride = (long long)ht_n;
            ss_len = (long long)ht_n;
        }
        dev.tn4 = nullptr;
        if (family_has_phase(family)) {
            if (tn4.alloc(ht_n)) return -1;
            hipLaunchKernelGGL(tn4_kernel, dim3(xb, yb), dim3(256), 0, s, traw.p, tn4.p, d_utoff.p, d_utlen.p, n_utargets, tables.p);
            dev.tn4 = tn4.p;
        }
        HIP_OK(hipGetLastError());
        lap("kernels queued");
        int hbad2[2] = {0, 0};
        if (bad.download(hbad2, 2, s)) return -1;
        HIP_OK(c4_stream_sync(s));
        const int hbad = hbad2[0];
        tdense_n = 0;
        if (!family_is_p2d(family) && hbad2[1]) {
            uint8_t tab[32];
            memset(tab, 0xff, 24);
            memset(tab + 24, 0, 8);
            int nd = 0;
            for (int c = 0; c < 24; c++)
                if (hbad2[1] >> c & 1) { if (nd < 8) { tab[c] = (uint8_t)nd; tab[24 + nd] = (uint8_t)c; } nd++; }
            if (nd <= 8) {
                if (tdense.upload(tab, 32, s) || tcode_dense.alloc(ht_n)) return -1;
                hipLaunchKernelGGL(dense_kernel, dim3(blocks), dim3(256), 0, s, tcode.p, tcode_dense.p, (long long)ht_n, tdense.p);
                HIP_OK(hipGetLastError());
                HIP_OK(c4_stream_sync(s));
                tdense_n = nd;
            }
        }
        lap("coded, splice arrays built");
        if (hbad) {
            c4h::set_error(hbad == 2 ? "a target codon translates outside the substitution matrix alphabet (non-IUPAC base?)"
                                     : "a residue is outside the 24-letter substitution matrix alphabet "
                                       "(exonerate's Submat index would read out of bounds, submat.c:27-61)");
            return -1;
        }
        dev.qcode = qcode.p; dev.tcode = tcode.p; dev.qoff = d_qoff.p; dev.toff = d_toff.p; dev.tlen = d_tlen.p;
        return 0;
    }

    // everything but the lock changes hands (c4gpu_batch_swap_stage: the batch takes the sequences a stage has loaded, the
    // stage takes the batch's old ones and loads the next batch into their buffers)
    void swap_with(ResidentSeqs &o) {
        std::swap(n_pairs, o.n_pairs);
        qoff.swap(o.qoff); toff.swap(o.toff); qlen.swap(o.qlen); tlen.swap(o.tlen);
        std::swap(total_q, o.total_q); std::swap(total_t, o.total_t);
        qraw.swap(o.qraw); traw.swap(o.traw); qcode.swap(o.qcode); tcode.swap(o.tcode);
        d_qoff.swap(o.d_qoff); d_toff.swap(o.d_toff); d_qlen.swap(o.d_qlen); d_tlen.swap(o.d_tlen); ss.swap(o.ss);
        tn4.swap(o.tn4); ss16.swap(o.ss16); std::swap(ss16_built, o.ss16_built);
        tdense.swap(o.tdense); std::swap(tdense_n, o.tdense_n); std::swap(ss_len, o.ss_len); tcode_dense.swap(o.tcode_dense);
        tables.swap(o.tables); splice_models.swap(o.splice_models); bad.swap(o.bad);
        std::swap(dev, o.dev);
        std::swap(n_utargets, o.n_utargets); d_utoff.swap(o.d_utoff); d_utlen.swap(o.d_utlen);
        std::swap(pin_q.p, o.pin_q.p); std::swap(pin_q.n, o.pin_q.n); std::swap(pin_t.p, o.pin_t.p); std::swap(pin_t.n, o.pin_t.n);
    }
};

// ---- launching a list of jobs ---------------------------------------------------------------------------------
struct JobSpec {                  // host description of one Viterbi call
    int pair;
    c4gpu_region region;
    int first_state = 0, final_state = 1, cp_count = 0;
    int root = 0;                        // the state the path's END is entered from, for the kernels that only compute that state's
                                         // component (Roots, c4_viterbi16_kernel.h); 0: not known
    int first_cell[CELL_MAX] = {0};
    bool dump_checkpoints = false;
    const c4gpu_subopt *sub = nullptr;   // sub-optimal blocking for this call (else the engine's per-pair table)
    const int32_t *span_in = nullptr;    // span models: start cells (host), END cells (host, updated in place)
    int32_t *span_out = nullptr;
};
typedef std::vector<std::pair<int32_t, int32_t>> RegionPoints;   // (target, query) in region coordinates, ascending
// the runs of one path: the sub-alignments between checkpoints are a handful of runs each and there are
// hundreds of thousands of them per batch, so short lists live inline (no allocation per job)
struct RunList {
    static constexpr int INLINE = 6;
    int n = 0;
    uint32_t small[INLINE];
    std::vector<uint32_t> big;
    const uint32_t *begin() const { return n <= INLINE ? small : big.data(); }
    const uint32_t *end() const { return begin() + n; }
    void assign_reversed(const uint32_t *first, int count) {
        n = count;
        uint32_t *dst = small;
        if (count > INLINE) { big.resize(count); dst = big.data(); }
        for (int k = 0; k < count; k++) dst[k] = first[count - 1 - k];
    }
};
struct JobOut {
    DevResult res;
    RunList runs;                 // PATH: (transition << 24 | length) runs, START -> END
    std::vector<DevVsa> vsa;      // CKPT: sub-alignments, last section first
    std::vector<int> checkpoints; // CKPT + dump_checkpoints
    bool packed = false;          // CKPT: the cells come from the packed 16-bit pass (final_cell_equiv)
};

// The windowed region pass (c4_viterbi_kernel.h, SEED): what a launch needs to know about the column dumps.
struct SeedPlan {
    int mode = 0;                  // 1: the score pass writes dumps; 2: the region windows start from them
    int kshift = 11;               // a dump every 1 << kshift columns
    bool fmt16 = false;            // the dumps are the packed score pass's 16-bit rows (Dump16), read by the packed windows
                                   // (c4_win16_kernel.h): set by the mode 1 run, handed on to the mode 2 run
    int seedw = 0, dc = 0;         // ints per dumped row, dumped columns per dump: of the kernel the mode 1 run took
    std::vector<long long> off;    // per spec: mode 1 (out) start of the job's dumps; mode 2 (in) the dump to start from, -1 = none
    std::vector<int> rows;         // per spec, mode 2: rows of a dumped column (Q + 1 of the score pass)
    // mode 2, windows chained on the device (DevJob::seed_base ..): per spec the pair's first dump, the dump the first
    // window starts from, its lattice column, the target_start of the score pass's rectangle; hops: windows per job
    std::vector<long long> base;
    std::vector<int> d, t0w, t0_base;
    int hops = 0;
};

// cooperating waves per job of the region windows (SEED 2): two where the family has that form -- a job's later windows are
// a few hundred rows high and leave fewer waves idle than with four (north-star batch: 846 against 867 ms per step) --
// three waves: 894-916 ms, one wave with every strip boundary through HBM: 854-872 ms, four rows per lane on two waves:
// 851-857 ms -- C4GPU_WIN_NW=4 keeps four
int window_waves(int family) {
    int nw = c4cfg::num(c4cfg::WIN_NW, 2);
    if (nw != 2) nw = 4;
    return get_kernel_mw(family, MODE_REGION, true, true, nw, false, 2) ? nw : 4;
}

struct Engine {
    c4gpu_ctx *ctx;
    const c4gpu_model *model;
    int family;
    bool local;
    // The local-scope score / region kernels drop the row-0 validity mask (c4_viterbi_kernel.h, eval_cell):
    // a phantom candidate there is -987654321 plus one or two calc values and must stay far below every
    // candidate the reference has.  True while the calc magnitudes are small against 987654321; checked
    // here, and parameters outside that range get the general kernels (all masks kept) instead.
    bool local_exact = true;
    bool pk16_params_ok = false;         // see init: the parameters allow the packed 16-bit score pass
    // largest magnitude one transition can add (calc constants, substitution scores, a splice model's best column sum):
    // what the continuation kernels without the row-0 mask need to stay exact (cont_free_ok)
    double calc_bound = 0;
    int pk16_match_max = 1, pk16_max_intron = 0;
    // what ONE intron can add to a path (best 5' sum + best 3' sum + the opening constant; 0 when that is negative, as with
    // the default parameters: 13 + 16 - 30) and the fewest target columns it takes: a path gains at most
    // (Q + 1) x pk16_match_max + (T / pk16_intron_cols + 1) x pk16_intron_gain (pk16_fits)
    int pk16_intron_gain = 0, pk16_intron_cols = 4;
    int loop_tr_host[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};     // KParams::loop_tr
    DevBuf<KParams> kparams;
    // reusable device buffers
    DevBuf<DevJob> d_jobs;
    DevBuf<DevResult> d_results;
    DevBuf<DevVsa> d_vsa;
    DevBuf<uint32_t> d_runs, d_runs_out;
    DevBuf<unsigned long long> d_runs_used;
    DevBuf<uint8_t> d_ops;
    DevBuf<int> d_bnd, d_ckpt, d_ckpt_dump, d_queue;
    DevBuf<uint32_t> d_tb;
    DevBuf<int> d_sub_t, d_sub_q, d_sub_colptr, d_span, d_seed, d_pairs;
    // fused_reduced_paths: checkpoint jobs, their results and sub-alignment lists, the sub-alignment jobs built from them
    DevBuf<DevJob> d_fjobs, d_fsub_jobs;
    DevBuf<DevResult> d_fres, d_fsub_res;
    DevBuf<DevVsa> d_fvsa;
    DevBuf<int> d_ffirst, d_fflags, d_fout;
    DevBuf<unsigned long long> d_fstats;
    DevBuf<FusePair> d_fpairs;
    std::vector<DevJob> hf_jobs;
    std::vector<DevResult> hf_res;
    std::vector<int> hf_first, hf_flags, hf_out;
    std::vector<FusePair> hf_pairs;
    // reusable host staging of run_impl (a sub-alignment launch lists ~10^5 jobs: fresh vectors of that size
    // are page-faulted in on every call)
    std::vector<int> h_order;
    std::vector<long long> h_key;
    std::vector<DevJob> h_jobs;
    std::vector<DevResult> h_res;
    std::vector<uint32_t> h_runs;
    std::vector<DevVsa> h_vsa;
    std::vector<int> h_dump;
    std::vector<JobSpec> fp_specs;          // ... and of find_path_batch
    std::vector<JobSpec> fp_sub_specs;      // its sub-alignment launch (every entry rewritten; only resized when the count changes)
    std::vector<JobOut> fp_outs;
    // per-pair SubOpt of the Optimal_find_path in progress (NULL entries / NULL table: nothing blocked)
    const std::vector<const c4gpu_subopt *> *pair_sub = nullptr;

    int init(c4gpu_ctx *c, const c4gpu_model *m, const c4gpu_params *params) {
        KParams kp;
        if (init_host(c, m, params, kp)) return -1;
        return kparams.upload(&kp, 1, ctx->stream);
    }
    // everything init decides on the host (family, guards of the shortcuts and of the packed passes): no device call
    int init_host(c4gpu_ctx *c, const c4gpu_model *m, const c4gpu_params *params, KParams &kp) {
        ctx = c; model = m;
        family = model_family(*m);
        if (family < 0) {
            c4h::set_error(std::string("model [") + m->name + "] is not one of the device-accelerated families");
            return -1;
        }
        local = m->start_scope == C4GPU_SCOPE_ANYWHERE && m->end_scope == C4GPU_SCOPE_ANYWHERE;
        memset(&kp, 0, sizeof kp);
        for (int i = 0; i < 16; i++) loop_tr_host[i] = -1;
        for (int i = 0; i < m->n_calcs; i++) kp.calc_value[i] = m->calcs[i].value;
        kp.min_intron = params->min_intron; kp.max_intron = params->max_intron;
        if (params->max_intron < params->min_intron) {
            // the reference would reject every intron (length < min or > max is always true); the device
            // test folds both comparisons into one and needs a non-negative span for that
            kp.min_intron = kp.max_intron = 0x7fffffff;     // span 0 at a length no intron has: always rejected
        }
        kp.start_scope = m->start_scope; kp.end_scope = m->end_scope;
        bool protein = false;
        for (int i = 0; i < m->n_calcs; i++)
            if (m->calcs[i].kind == C4GPU_CALC_MATCH_PROTEIN || m->calcs[i].kind == C4GPU_CALC_MATCH_P2D) protein = true;
        memcpy(kp.submat, protein ? &params->protein_submat[0][0] : &params->dna_submat[0][0], sizeof kp.submat);
        {
            // largest magnitude one calc can contribute (the -987654321 sentinel of --forcegtag aside: it is
            // the same number in the reference)
            double pmax = 0;
            for (int i = 0; i < m->n_calcs; i++) pmax = std::max(pmax, std::fabs((double)m->calcs[i].value));
            for (int i = 0; i < 24 * 24; i++) pmax = std::max(pmax, std::fabs((double)kp.submat[i]));
            double smax = 0;
            if (family_has_splice(family))
                for (int k = 0; k < 4; k++) {
                    const c4gpu_splice_model &sp = params->splice[k];
                    double sum = 0;
                    for (int r = 0; r < sp.model_length && r < C4GPU_SPLICE_MAX_LEN; r++) {
                        double mx = 0;
                        for (int c = 0; c < 5; c++) mx = std::max(mx, std::fabs((double)sp.data[r][c]));
                        sum += mx;
                    }
                    smax = std::max(smax, sum + 1.0);
                }
            // a real candidate is at least -(states x largest calc); a phantom one at most LOW + 3 calcs
            local_exact = (pmax + smax) * (m->n_states + 4) < 4.0e8;
            calc_bound = pmax + smax;
            // the packed 16-bit score pass (c4_viterbi16_kernel.h): every constant far inside 16 bits, a usual intron window
            double mmax = 0;
            for (int i = 0; i < 24 * 24; i++) mmax = std::max(mmax, (double)kp.submat[i]);
            pk16_match_max = (int)std::max(1.0, mmax);
            pk16_params_ok = pmax <= 16000.0 && smax <= 16000.0 && params->min_intron >= 0 && params->min_intron <= 30000 &&
                             params->max_intron >= params->min_intron;
            pk16_max_intron = params->max_intron;
            if (family_has_splice(family)) {
                // largest value a site of each kind can score: every PSSM row at its best, rounded as the predictor rounds
                int best[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4; k++) {
                    const c4gpu_splice_model &sp = params->splice[k];
                    double sum = 0;
                    for (int r = 0; r < sp.model_length && r < C4GPU_SPLICE_MAX_LEN; r++) {
                        double mx = 0;
                        for (int c = 0; c < 5; c++) mx = std::max(mx, (double)sp.data[r][c]);
                        sum += mx;
                    }
                    best[k] = (int)std::floor(sum + 0.5) + 1;
                }
                // an intron enters its state through a pre-splice transition and leaves it through a post-splice transition OF THE
                // SAME STATE (est2genome: the forward strand's intron state and the reverse strand's are two states that share
                // nothing): the most one intron can add is the best such pairing per state, not the best pre-site of one strand
                // with the best post-site of the other (rounds 3-4: that bound put 2 x T / 30 on top of every query; per state
                // it is 1, and 1 kb x 100 kb queries of up to ~2 500 nt fit where ~1 865 did)
                long long gain = -0x40000000LL;
                for (int st = 0; st < m->n_states; st++) {
                    long long pre = -0x40000000LL, post = -0x40000000LL;
                    for (int k = 0; k < m->n_transitions; k++) {
                        const c4gpu_transition &t = m->transitions[k];
                        if (t.calc < 0) continue;
                        const c4gpu_calc &cc = m->calcs[t.calc];
                        const int prm = cc.param & 3;
                        if (cc.kind == C4GPU_CALC_SPLICE_PRE && t.output == st) pre = std::max<long long>(pre, (long long)cc.value + best[prm]);
                        if (cc.kind == C4GPU_CALC_SPLICE_POST && t.input == st) post = std::max<long long>(post, (long long)cc.value + best[prm]);
                    }
                    if (pre > -0x40000000LL && post > -0x40000000LL) gain = std::max(gain, pre + post);
                }
                pk16_intron_gain = (int)std::max<long long>(0, std::min<long long>(gain, 1 << 20));
                pk16_intron_cols = std::max(4, params->min_intron);
                // KParams::loop_tr (c4_viterbi_kernel.h, viterbi_kernel): a state s with a loop (0, 1) that adds nothing, entered
                // by pre-splice and left by post-splice transitions only, where leaving and coming back cannot pay: the best
                // 3' site + the best 5' site + the opening constant of s's own transitions is NEGATIVE (the sites at their best
                // as the predictor rounds them, splice.c:379-381 -- without the + 1 of slack the 16-bit guard above allows
                // itself, but with 0.001 for the float sums: 13 + 15 - 30 = -2 under the default parameters), and nothing
                // else that moves along the target without a query row adds anything (gaps cost).  Then every cell of a
                // one-row continuation from s takes the loop, whatever the order of the candidates.  C4GPU_LOOP_SHORTCUT=0: off.
                const bool loop_on = !(c4cfg::is(c4cfg::LOOP_SHORTCUT, 0));
                bool others_cost = true;
                for (int k = 0; k < m->n_transitions; k++) {
                    const c4gpu_transition &t = m->transitions[k];
                    if (t.advance_query != 0 || t.calc < 0) continue;
                    const c4gpu_calc &cc = m->calcs[t.calc];
                    if (cc.kind == C4GPU_CALC_SPLICE_PRE || cc.kind == C4GPU_CALC_SPLICE_POST) continue;
                    if (cc.kind != C4GPU_CALC_CONST || cc.value > 0) others_cost = false;
                }
                if (loop_on && family == FAM_EST2GENOME && others_cost && m->n_states <= 16) {
                    int tight[4];
                    for (int k = 0; k < 4; k++) {
                        const c4gpu_splice_model &sp = params->splice[k];
                        double sum = 0;
                        for (int r = 0; r < sp.model_length && r < C4GPU_SPLICE_MAX_LEN; r++) {
                            double mx = 0;
                            for (int c = 0; c < 5; c++) mx = std::max(mx, (double)sp.data[r][c]);
                            sum += mx;
                        }
                        tight[k] = (int)std::floor(sum + 0.5 + 1e-3);
                    }
                    for (int st = 0; st < m->n_states; st++) {
                        long long pre = -0x40000000LL, post = -0x40000000LL;
                        int loop = -1;
                        bool clean = true;
                        for (int k = 0; k < m->n_transitions; k++) {
                            const c4gpu_transition &t = m->transitions[k];
                            const c4gpu_calc *cc = t.calc >= 0 ? &m->calcs[t.calc] : nullptr;
                            if (t.input == st && t.output == st) {
                                if (t.advance_query == 0 && t.advance_target == 1 && !cc && loop < 0) loop = k; else clean = false;
                            } else if (t.output == st) {
                                if (cc && cc->kind == C4GPU_CALC_SPLICE_PRE) pre = std::max<long long>(pre, (long long)cc->value + tight[cc->param & 3]);
                                else clean = false;
                            } else if (t.input == st) {
                                if (cc && cc->kind == C4GPU_CALC_SPLICE_POST) post = std::max<long long>(post, (long long)cc->value + tight[cc->param & 3]);
                                else clean = false;
                            }
                        }
                        loop_tr_host[st] = (clean && loop >= 0 && pre > -0x40000000LL && post > -0x40000000LL && pre + post < 0) ? loop : -1;
                    }
                }
            }
            if (c4cfg::is(c4cfg::LOCAL_EXACT, 0)) local_exact = false;   // test hook
        }
        for (int i = 0; i < 16; i++) kp.loop_tr[i] = loop_tr_host[i];
        if (c4cfg::has(c4cfg::TRACE) && family == FAM_EST2GENOME) {
            std::string which;
            for (int i = 0; i < 16; i++) if (loop_tr_host[i] >= 0) which += " " + std::to_string(i) + ":" + std::to_string(loop_tr_host[i]);
            fprintf(stderr, "c4gpu trace: one-row sections answered without a DP for states (state:loop transition)%s\n",
                    which.empty() ? " none" : which.c_str());
        }
        for (int c = 0; c < 4096; c++) {
            const uint8_t row = params->submat_index[params->aa[params->trans[c]]];
            kp.codon_row[c] = row < 24 ? row : 0;        // '-' (empty mask) never scores: such columns are rejected at upload
        }
        return 0;
    }

    // The continuation kernels compiled without the row-0 validity mask (c4_viterbi_kernel.h, eval_cell: CONT && LOCAL)
    // hold, in states the reference leaves unset, -987654321 plus at most one calc per cell of a path: exact while
    // that stays far below every real score of the job, i.e. (Q + T + 2) x the largest calc well under 987654321 / 2.
    // C4GPU_CONT_FREE=0 keeps the kernels with every mask.
    bool cont_free_ok(long long q_plus_t) const {
        if (c4cfg::is(c4cfg::CONT_FREE, 0)) return false;
        return (double)(q_plus_t + 2) * std::max(calc_bound, 1.0) < 4.0e8;
    }

    // The packed 16-bit passes (c4_viterbi16_kernel.h, c4_ckpt16_kernel.h) are exact while everything a path can gain stays
    // inside 16 000: the substitution scores of its query rows plus, where an intron's two sites can outweigh its opening
    // penalty (a small --intronpenalty), that net gain once per intron the target has room for; and the intron length test
    // must not be able to fail on the upper side (the packed length counter saturates).
    bool pk16_fits(int query_length, int target_length) const {
        const double gain = (double)(query_length + 1) * pk16_match_max +
                            (double)(target_length / pk16_intron_cols + 1) * pk16_intron_gain;
        return gain <= 16000.0 && (!family_has_splice(family) || (long long)target_length + 4 <= (long long)pk16_max_intron);
    }
    // the four splice values of every target position as the packed passes add them (ss16_kernel): built once per batch,
    // finished before the lock is released (the other lane launches on another stream)
    int ensure_ss16(const ResidentSeqs &seqs) {
        if (!family_has_splice(family)) return 0;
        std::lock_guard<std::mutex> hold(seqs.ss16_lock);
        if (!seqs.ss16_built) {
            if (seqs.ss16.alloc((size_t)seqs.ss_len)) return -1;
            HIP_OK(pk16_build_splice(family, kparams.p, seqs.dev.ss, seqs.dev.ss_stride, seqs.ss_len, seqs.ss16.p, ctx->stream));
            HIP_OK(c4_stream_sync(ctx->stream));
            seqs.ss16_built = true;
        }
        return 0;
    }

    // Runs `specs` in `mode`; out[i] corresponds to specs[i].  Calls whose region holds blocked cells
    // (SubOpt_Index_create returns an index, subopt.c:250-266) go to the kernels compiled with blocking, the
    // others (it returns NULL) to the plain ones.
    int run(const ResidentSeqs &seqs, int mode, bool cont, const std::vector<JobSpec> &specs, std::vector<JobOut> &out,
            SeedPlan *seed = nullptr) {
        const int n = (int)specs.size();
        if (seed) return run_impl(seqs, mode, cont, specs, out, nullptr, seed);
        for (int i = 0; i < n; i++)
            if (specs[i].span_in || specs[i].span_out) return run_impl(seqs, mode, cont, specs, out, nullptr);
        std::vector<int> plain, blocked;
        std::vector<RegionPoints> pts;
        for (int i = 0; i < n; i++) {
            const c4gpu_subopt *so = specs[i].sub ? specs[i].sub : (pair_sub ? (*pair_sub)[specs[i].pair] : nullptr);
            RegionPoints rp;
            if (so && !so->points.empty()) c4h::subopt_region_points(so, specs[i].region, rp);
            if (rp.empty()) plain.push_back(i);
            else { blocked.push_back(i); pts.push_back(std::move(rp)); }
        }
        if (blocked.empty()) return run_impl(seqs, mode, cont, specs, out, nullptr);
        // a handful of calls without blocked cells would occupy a corner of the device for as long as a full
        // launch: they ride along in the blocking kernels with empty lists
        if (plain.size() * 4 < blocked.size()) {
            std::vector<RegionPoints> all(n);
            for (size_t x = 0; x < blocked.size(); x++) all[blocked[x]] = std::move(pts[x]);
            return run_impl(seqs, mode, cont, specs, out, &all);
        }
        out.assign(n, JobOut());
        std::vector<JobSpec> part;
        std::vector<JobOut> part_out;
        for (int i : plain) part.push_back(specs[i]);
        if (run_impl(seqs, mode, cont, part, part_out, nullptr)) return -1;
        for (size_t x = 0; x < plain.size(); x++) out[plain[x]] = std::move(part_out[x]);
        part.clear();
        for (int i : blocked) part.push_back(specs[i]);
        if (run_impl(seqs, mode, cont, part, part_out, &pts)) return -1;
        for (size_t x = 0; x < blocked.size(); x++) out[blocked[x]] = std::move(part_out[x]);
        return 0;
    }

    int run_impl(const ResidentSeqs &seqs, int mode, bool cont, const std::vector<JobSpec> &specs,
                 std::vector<JobOut> &out, const std::vector<RegionPoints> *pts, SeedPlan *seed = nullptr) {
        const bool trace = c4cfg::has(c4cfg::TRACE);
        const auto t_begin = std::chrono::steady_clock::now();
        struct Trace {
            bool on; std::chrono::steady_clock::time_point t0; int mode, n;
            ~Trace() {
                if (on) fprintf(stderr, "c4gpu trace: run mode %d jobs %d host+device %.3f ms\n", mode, n,
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
        } tr{trace, t_begin, mode, (int)specs.size()};
        auto lap = [&](const char *what) {
            if (trace) fprintf(stderr, "c4gpu trace:   %-18s at %.3f ms\n", what,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
        };
        const int n = (int)specs.size();
        // every field of entries [0, n) is rewritten after the launch; a vector that is reused for launches of very
        // different sizes (4 096 checkpoint jobs, then 778 443 sub-alignments, every step) is never shrunk: building
        // and tearing down its tail was the largest host item of such a step
        if ((int)out.size() < n) out.resize(n);
        if (!n) return 0;
        const bool use_local = local && local_exact && !cont && (mode == MODE_SCORE || mode == MODE_REGION);
        // packed region-start slot: (query_start << tshift) | target_start must fit 31 bits for every job
        auto nbits = [](int v) { int b = 0; while ((1LL << b) <= v) b++; return b; };
        // C4GPU_PACK=0 forces the two-slot form (what targets beyond 2^31 / query-rows columns get): read on
        // every call so that a test can switch it
        bool pack = (mode == MODE_REGION) && !(c4cfg::is(c4cfg::PACK, 0));
        for (int i = 0; i < n && pack; i++)
            pack = nbits(specs[i].region.query_length) + nbits(specs[i].region.target_length) <= 31;
        const int wpe_env = c4cfg::num(c4cfg::WPE, 0);
        int span = 0;
        for (int i = 0; i < n; i++) {
            const int sp = specs[i].span_in ? 1 : (specs[i].span_out ? 2 : 0);
            if (i && sp != span) { c4h::set_error("jobs with and without span matrices in one call"); return -1; }
            span = sp;
        }
        bool cont_free = false;
        if (cont && (mode == MODE_PATH || mode == MODE_CKPT) && !pts && !span) {
            long long worst = 0;
            bool cells_leave = false;             // checkpoint cells handed to the caller must be the reference's in every state
            for (int i = 0; i < n; i++) {
                worst = std::max(worst, (long long)specs[i].region.query_length + specs[i].region.target_length);
                cells_leave |= specs[i].dump_checkpoints;
            }
            cont_free = !cells_leave && cont_free_ok(worst);
        }
        const uint8_t *staged_codes = nullptr;         // set with the staged packed score pass: its residue-code table
        const KernelInfo *ki = get_kernel(family, mode, cont, cont ? cont_free : use_local, pack, pts ? 0 : wpe_env, pts != nullptr, span);
        if (!ki && cont_free) ki = get_kernel(family, mode, cont, false, pack, pts ? 0 : wpe_env, pts != nullptr, span);
        if (!ki) { c4h::set_error("no compiled kernel for this model/mode"); return -1; }
        const int span_cs = 1 + model->total_shadow_designations;
        // whole-rectangle passes whose query spans several 64*R-row strips run on 4 cooperating waves per
        // job (strip carry rows stay in LDS instead of HBM); C4GPU_MW=0 forces the one-wave kernels
        const int mw_env = c4cfg::num(c4cfg::MW, 1);
        if (seed) {
            int win_nw = seed->mode == 2 ? window_waves(family) : 4;
            // the score pass of a launch with too few jobs to occupy the device on four waves each (256 proteins against one
            // chromosome): eight waves of half the rows (C4GPU_MW=4 keeps four)
            if (seed->mode == 1 && mw_env != 4 && (long long)n * 8 <= 2LL * 4 * ctx->prop.multiProcessorCount &&
                get_kernel_mw(family, mode, true, false, 8, false, 1))
                win_nw = 8;
            ki = get_kernel_mw(family, mode, true, mode == MODE_REGION, win_nw, false, seed->mode);
            if (!ki || !use_local || (mode == MODE_REGION && !pack)) { c4h::set_error("no seeded kernel for this launch"); return -1; }
            // the score pass with dumps: two jobs per lane in packed 16-bit halves where every score fits (C4GPU_PK16=0: never)
            const int pk_env = c4cfg::num(c4cfg::PK16, 1);
            const KernelInfo *kpk = (seed->mode == 1 && pk_env && pk16_params_ok && n >= 2) ? get_kernel_pk16(family, pk_env == 3 ? 0 : pk_env == 4 ? 2 : 1) : nullptr;      // 3: the all-asm form (c4_viterbi16_kernel.h, VAR 0)
            if (kpk) {
                bool fits = true;
                for (int i = 0; i < n && fits; i++) fits = pk16_fits(specs[i].region.query_length, specs[i].region.target_length);
                // variant 1 reads the four splice values of a column as one packed 8-byte entry
                if (fits && pk_env != 3 && ensure_ss16(seqs)) return -1;
                if (fits) ki = kpk;
                // ... and with the packed region windows behind it (c4_win16_kernel.h; C4GPU_WIN16=0: the 32-bit windows) it
                // writes its dumps as 16-bit rows: window rows and columns must fit 15 / 16 bits
                const int w16_env = c4cfg::num(c4cfg::WIN16, 1);
                const KernelInfo *kd = (fits && pk_env == 1 && w16_env) ? get_kernel_pk16(family, 3) : nullptr;
                // (the packed windows index a query profile by the targets' dense codes: at most eight residue codes in the batch)
                if (kd && get_kernel_win16(family, 0) && seed->kshift <= 15 && seqs.tdense_n > 0) {
                    bool rows_ok = true;
                    for (int i = 0; i < n && rows_ok; i++) rows_ok = specs[i].region.query_length < 32000;
                    if (rows_ok) { ki = kd; seed->fmt16 = true; }
                    // ... and with its column loop fed from LDS alone (IO 1) where every query fits the strips of one workgroup
                    // and the targets hold few enough residue codes for the query profile (C4GPU_PK16_IO=0: never; 1: with a barrier per chunk instead of progress counters)
                    const int io_env = c4cfg::num(c4cfg::PK16_IO, 2);
                    const KernelInfo *ke = (rows_ok && io_env) ? get_kernel_pk16(family, io_env == 2 ? 5 : 4) : nullptr;     // 2 (default): progress counters between the cooperating waves; 1: a barrier per chunk
                    // seven or eight codes (IUPAC ambiguity codes in the targets): the staged form with the larger profile, where every
                    // query fits its four strips of 256 rows (C4GPU_PK16_C8=0: the form that loads per step)
                    if (ke && seqs.tdense_n > pk16_staged_codes() && seqs.tdense_n <= 8 && io_env == 2 &&
                        !(c4cfg::is(c4cfg::PK16_C8, 0)) && get_kernel_pk16(family, 8)) {
                        bool strips_ok = true;
                        for (int i = 0; i < n && strips_ok; i++) strips_ok = specs[i].region.query_length + 1 <= pk16_staged_rows();
                        if (strips_ok) { ki = get_kernel_pk16(family, 8); staged_codes = seqs.tdense.p; }
                    }
                    if (ke && seqs.tdense_n > 0 && seqs.tdense_n <= pk16_staged_codes()) {
                        bool strips_ok = true;
                        for (int i = 0; i < n && strips_ok; i++) strips_ok = specs[i].region.query_length + 1 <= pk16_staged_rows();
                        if (strips_ok) { ki = ke; staged_codes = seqs.tdense.p; }
                        else if (io_env == 2 && !(c4cfg::is(c4cfg::PK16_R6, 0))) {
                            // queries of 1 024 .. 1 535 rows: six rows per lane put them into the four strips of one workgroup
                            // (C4GPU_PK16_R6=0: the form that loads per step, in two passes over the target)
                            const KernelInfo *kh = get_kernel_pk16(family, 7);
                            bool six_ok = kh != nullptr;
                            for (int i = 0; i < n && six_ok; i++) six_ok = specs[i].region.query_length + 1 <= pk16_staged_rows6();
                            if (six_ok) { ki = kh; staged_codes = seqs.tdense.p; }
                            // ... longer ones in several super-strips of that form (C4GPU_PK16_LONG=0: the per-step form)
                            else if (get_kernel_pk16(family, 9) && !(c4cfg::is(c4cfg::PK16_LONG, 0))) {
                                ki = get_kernel_pk16(family, 9); staged_codes = seqs.tdense.p;
                            }
                        }
                        // ... on eight waves of two rows per lane where the launch has at most one pair of jobs per compute unit (the
                        // shard of a strong-scaled run): twice the waves on the same rows (C4GPU_PK16_NW8=0: never; 1: always)
                        const int nw8_env = c4cfg::num(c4cfg::PK16_NW8, -1);
                        const KernelInfo *kg = (strips_ok && io_env == 2 && nw8_env != 0) ? get_kernel_pk16(family, 6) : nullptr;
                        if (kg && (nw8_env == 1 || (n + 1) / 2 <= ctx->prop.multiProcessorCount)) ki = kg;
                    }
                }
            }
            if (seed->mode == 1) { seed->seedw = ki->seedw; seed->dc = ki->max_at; }
            if (seed->mode == 2 && seed->fmt16) {
                const int w16_env = c4cfg::num(c4cfg::WIN16, 1);     // 2..9: one shape whatever the jobs (tests, measurement)
                int shape = w16_env == 9 ? 0 : w16_env - 1;
                if (w16_env <= 1) {          // the strips of a window on two cooperating waves where the first windows have two strips and more
                    long long strips = 0;
                    for (int i = 0; i < n; i++) strips += (specs[i].region.query_length + 1 + 255) / 256;
                    // ... on four where the launch has at most one pair of jobs per compute unit (the shard of a strong-scaled run:
                    // 512 pairs, region windows 30.5 -> 20.7 ms per step, profiles/r05_shard_sweep.log)
                    shape = strips >= 2LL * n ? ((n + 1) / 2 <= ctx->prop.multiProcessorCount ? 4 : 7) : 0;
                }
                ki = get_kernel_win16(family, shape);
                if (!ki || !seqs.ss16_built) { c4h::set_error("no packed window kernel for this launch"); return -1; }
            }
        } else if (mw_env && !cont && (mode == MODE_SCORE || mode == MODE_REGION)) {
            const KernelInfo *kmw = get_kernel_mw(family, mode, use_local, pack, 4, pts != nullptr);
            if (kmw) {
                long long strips = 0;
                for (int i = 0; i < n; i++) strips += (specs[i].region.query_length + 1 + 64 * kmw->R - 1) / (64 * kmw->R);
                if (strips >= 3LL * n) ki = kmw;
            }
            // 8 waves x 2 rows per lane cover the same rows per workgroup with twice the waves: taken when the
            // launch has too few jobs to occupy the device with 4 waves each (C4GPU_MW=4 keeps 4)
            const KernelInfo *kmw8 = (ki == kmw && mw_env != 4 && !pts) ? get_kernel_mw(family, mode, use_local, pack, 8) : nullptr;
            if (kmw8 && (long long)n * 8 <= 2LL * 4 * ctx->prop.multiProcessorCount) ki = kmw8;
        }
        if (seed && c4cfg::has(c4cfg::TRACE)) fprintf(stderr, "c4gpu trace:   seeded pass %d with kernel %s\n", seed->mode, ki->name);
        // longest first (persistent waves pull from the queue head)
        std::vector<int> &order = h_order;
        order.resize(n);
        std::iota(order.begin(), order.end(), 0);
        auto cells = [&](int i) { return (long long)(specs[i].region.query_length + 1) * (specs[i].region.target_length + 1); };
        {
            // ... unless the launch is tens of thousands of small jobs (the sub-alignments between checkpoints):
            // any order balances those, and sorting them is then the largest item on the host
            std::vector<long long> &key = h_key;
            key.resize(n);
            long long biggest = 0;
            for (int i = 0; i < n; i++) { key[i] = cells(i); biggest = std::max(biggest, key[i]); }
            if (ki->pairs)         // two jobs per lane, paired by the host: jobs of one root together, each group longest first
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                    return specs[a].root != specs[b].root ? specs[a].root < specs[b].root : key[a] > key[b];
                });
            else if (!(n > 32768 && biggest < (1 << 20)))
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b]; });
        }
        std::vector<DevJob> &jobs = h_jobs;
        jobs.resize(n);
        long long ops_total = 0, vsa_total = 0, dump_total = 0, max_T = 0, max_tb = 0, max_ckpt = 0, total_cells = 0;
        long long max_runs = 0, sub_cols = 0, span_total = 0, seed_total = 0;
        if (seed && seed->mode == 1) seed->off.assign(n, -1);
        std::vector<int> sub_t, sub_q;
        // the fields that do not depend on the jobs before it ...
        parallel_for(n, 32768, [&](long long first, long long last) {
            for (long long x = first; x < last; x++) {
                const JobSpec &s = specs[order[x]];
                DevJob &j = jobs[x];
                memset(&j, 0, sizeof j);
                j.seed_off = -1;
                if (seed) j.seed_kshift = seed->kshift;
                j.pair = s.pair; j.q0 = s.region.query_start; j.t0 = s.region.target_start;
                j.Q = s.region.query_length; j.T = s.region.target_length;
                j.first_state = s.first_state; j.final_state = s.final_state; j.cp_count = s.cp_count;
                j.root = s.root;
                j.tshift = nbits(j.T);
                memcpy(j.first_cell, s.first_cell, sizeof j.first_cell);
                j.ckpt_off = -1;
            }
        });
        // ... and the running offsets into the launch's buffers
        for (int x = 0; x < n; x++) {
            const JobSpec &s = specs[order[x]];
            DevJob &j = jobs[x];
            if (span) {
                j.span_off = span_total;
                span_total += (long long)(s.region.query_length + 1) * (s.region.target_length + 1) * span_cs;
            }
            if (pts) {
                const RegionPoints &rp = (*pts)[order[x]];
                j.sub_off = sub_cols; j.sub_pt_off = (int)sub_t.size(); j.sub_pt_n = (int)rp.size();
                sub_cols += s.region.target_length + 2;
                for (const auto &p : rp) { sub_t.push_back(p.first); sub_q.push_back(p.second); }
            }
            if (seed) {
                if (seed->mode == 1) {            // dumps d = 1 .. T >> kshift, two columns of Q + 1 rows each
                    j.seed_off = seed_total; j.seed_rows = s.region.query_length + 1;
                    seed->off[order[x]] = seed_total;
                    seed_total += (long long)(s.region.target_length >> seed->kshift) * ki->max_at * (s.region.query_length + 1) *
                                  ki->seedw;
                } else {
                    j.seed_off = seed->off[order[x]]; j.seed_rows = seed->rows[order[x]];
                    if (seed->hops) {
                        j.seed_base = seed->base[order[x]]; j.win_d = seed->d[order[x]]; j.win_t0w = seed->t0w[order[x]];
                        j.win_t0_base = seed->t0_base[order[x]]; j.win_hops = seed->hops;
                    }
                }
            }
            j.ops_off = ops_total; j.ops_cap = 0; j.vsa_off = (int)vsa_total;
            total_cells += (long long)(j.Q + 1) * (j.T + 1);
            max_T = std::max<long long>(max_T, j.T);
            if (mode == MODE_PATH) {
                const long long strips = (j.Q + 1 + 64 * ki->R - 1) / (64 * ki->R);
                j.ops_cap = 3 * (j.Q + j.T) + 16;
                max_runs = std::max<long long>(max_runs, j.ops_cap);
                ops_total += j.ops_cap;
                max_tb = std::max(max_tb, strips * (long long)(j.T + 64) * 64 * ki->R);
            }
            if (mode == MODE_CKPT) {
                const long long ck = (long long)j.cp_count * ki->max_at * (j.Q + 1) * ki->n_states * ki->cs;
                max_ckpt = std::max(max_ckpt, ck);
                vsa_total += j.cp_count + 1;
                if (s.dump_checkpoints) { j.ckpt_off = dump_total; dump_total += ck; }
            }
        }
        // windows chained on the device: a later window of a job spans one dump interval plus the dumped columns, which can be
        // more than every FIRST window of the launch: the strip carry rows are laid out for the longest window any hop can have
        if (seed && seed->mode == 2 && seed->hops)
            max_T = std::max<long long>(max_T, (1LL << seed->kshift) + 2LL * ki->max_at);
        lap("jobs built");
        // persistent grid: as many waves as the device keeps resident, bounded by the scratch it implies
        int blocks_per_cu = 0;
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, ki->func, 64 * ki->waves, 0));
        if (blocks_per_cu < 1) blocks_per_cu = 1;
        if (trace) fprintf(stderr, "c4gpu trace:   kernel %s: %d workgroups per CU\n", ki->name, blocks_per_cu);
        // the kernels that run two jobs per lane in pairs the host lists: neighbours of the same root
        std::vector<int> pair_list;
        if (ki->pairs)
            for (int x = 0; x < n;) {
                const bool two = x + 1 < n && jobs[x + 1].root == jobs[x].root;
                pair_list.push_back(x); pair_list.push_back(two ? x + 1 : -1);
                x += two ? 2 : 1;
            }
        long long grid = std::min<long long>(ki->pairs ? (long long)pair_list.size() / 2 : n, (long long)blocks_per_cu * ctx->prop.multiProcessorCount);
        // strip carry rows in HBM are only needed when a job has more strips than one workgroup holds at once
        // (one for the single-wave kernels, `waves` for the cooperating ones)
        long long carry_T = 0;
        for (int x = 0; x < n; x++)
            if ((jobs[x].Q + 1 + 64 * ki->R - 1) / (64 * ki->R) > (ki->hbm_carry ? 1 : ki->waves)) carry_T = max_T;
        // per workgroup: one "empty" column (what the first strip reads as its row above) + two carry rows
        const long long bnd_per_wave = (2 * ((carry_T ? carry_T : 0) + 1) + 1) * (long long)std::max(ki->bnd, 1);
        const long long bytes_per_wave = bnd_per_wave * 4 + max_tb * 4 + max_ckpt * 4 + max_runs * 4;
        // compact run array: paths are mostly long runs, so a fraction of the worst case is plenty; a
        // launch that overflows it is repeated with the worst case
        long long runs_capacity = std::min<long long>(ops_total, std::max<long long>(1 << 20, (long long)n * 256));
        const long long budget = (long long)(ctx->prop.totalGlobalMem / 4);
        if (bytes_per_wave * grid > budget) grid = std::max<long long>(1, budget / std::max<long long>(1, bytes_per_wave));
        hipStream_t s = ctx->stream;
        std::vector<DevResult> &res = h_res;
        res.resize(n);
        std::vector<uint32_t> &runs = h_runs;
        runs.clear();
        std::vector<DevVsa> &vsa = h_vsa;         // kept between launches: fresh vectors of this size are zeroed and
        std::vector<int> &dump = h_dump;          // page-faulted in on every call (50 MB for a C2-shaped batch)
        vsa.resize(vsa_total);
        dump.resize(dump_total);
        for (int attempt = 0; attempt < 2; attempt++) {
            int zero = 0;
            unsigned long long zero64 = 0;
            if (d_jobs.upload(jobs.data(), n, s) || d_results.alloc(n) || d_queue.upload(&zero, 1, s) ||
                d_runs_used.upload(&zero64, 1, s) || d_bnd.alloc(bnd_per_wave * grid) || d_vsa.alloc(vsa_total) ||
                d_runs.alloc(max_runs * grid) || d_runs_out.alloc(runs_capacity) ||
                d_tb.alloc(max_tb * grid) || d_ckpt.alloc(max_ckpt * grid) || d_ckpt_dump.alloc(dump_total))
                return -1;
            LaunchArgs a;
            a.kp = kparams.p; a.seqs = seqs.dev; a.jobs = d_jobs.p; a.n_jobs = n; a.results = d_results.p;
            a.seqs.sub_colptr = nullptr; a.seqs.sub_rows = nullptr;
            a.seqs.span_in = nullptr; a.seqs.span_out = nullptr;
            if (span) {                                      // matrices of all jobs, in job order
                std::vector<int> host(span_total);
                for (int x = 0; x < n; x++) {
                    const JobSpec &sp = specs[order[x]];
                    const int32_t *src = span == 1 ? sp.span_in : sp.span_out;
                    const long long cnt = (long long)(jobs[x].Q + 1) * (jobs[x].T + 1) * span_cs;
                    memcpy(host.data() + jobs[x].span_off, src, sizeof(int) * cnt);
                }
                if (d_span.upload(host.data(), span_total, s)) return -1;
                a.seqs.span_in = d_span.p; a.seqs.span_out = d_span.p;
            }
            if (pts) {
                sub_q.push_back(0);                      // the kernels' row prefetch may touch one entry past the last list
                if (d_sub_t.upload(sub_t.data(), sub_t.size(), s) || d_sub_q.upload(sub_q.data(), sub_q.size(), s) ||
                    d_sub_colptr.alloc(2 * sub_cols)) return -1;
                hipLaunchKernelGGL(subopt_colptr_kernel, dim3(std::min(n, 65535)), dim3(256), 0, s, d_jobs.p, n,
                                   d_sub_t.p, d_sub_q.p, d_sub_colptr.p);
                HIP_OK(hipGetLastError());
                a.seqs.sub_colptr = d_sub_colptr.p; a.seqs.sub_rows = d_sub_q.p;
            }
            a.seqs.ss16 = seqs.ss16_built ? seqs.ss16.p : nullptr;
            if (seed && seed->mode == 2 && seed->fmt16) {
                // the packed windows' dense target codes and code table ride in the two pointers no packed kernel reads otherwise
                if (seqs.tdense_n <= 0) { c4h::set_error("packed region windows without a residue-code table"); return -1; }
                a.seqs.sub_rows = reinterpret_cast<const int *>(seqs.tcode_dense.p);
                a.seqs.sub_colptr = reinterpret_cast<const int *>(seqs.tdense.p);
            }
            a.seqs.seed = nullptr;
            if (seed) {
                if (seed->mode == 1 && d_seed.alloc((size_t)std::max<long long>(seed_total, 1))) return -1;
                a.seqs.seed = d_seed.p;
            }
            a.vsas = d_vsa.p; a.ops = nullptr; a.queue = d_queue.p; a.grid = (int)grid; a.stream = s;
            if (ki->pairs) {
                if (d_pairs.upload(pair_list.data(), pair_list.size(), s)) return -1;
                a.aux = d_pairs.p; a.n_aux = (int)(pair_list.size() / 2);
            }
            if (staged_codes) a.aux = reinterpret_cast<const int *>(staged_codes);
            a.scratch.bnd = d_bnd.p; a.scratch.bnd_stride = bnd_per_wave; a.scratch.carry = carry_T ? 1 : 0;
            a.scratch.tb = max_tb ? d_tb.p : nullptr; a.scratch.tb_stride = max_tb;
            a.scratch.ckpt = max_ckpt ? d_ckpt.p : nullptr; a.scratch.ckpt_stride = max_ckpt;
            a.scratch.ckpt_dump = d_ckpt_dump.p;
            a.scratch.runs = d_runs.p; a.scratch.runs_stride = max_runs;
            a.scratch.runs_out = d_runs_out.p; a.scratch.runs_capacity = runs_capacity;
            a.scratch.runs_used = d_runs_used.p;
            lap("uploaded");
            if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev0, s));
            HIP_OK(ki->launch(a));
            if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev1, s));
            unsigned long long used = 0;
            if (d_results.download(res.data(), n, s) || d_runs_used.download(&used, 1, s) ||
                d_vsa.download(vsa.data(), vsa_total, s) || d_ckpt_dump.download(dump.data(), dump_total, s))
                return -1;
            HIP_OK(c4_stream_sync(s));
            lap("kernel + results");
            if (ctx->timing) {
                float ms = 0;
                HIP_OK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
                ctx->kernel_ms[mode] += ms; ctx->kernel_launches[mode]++; ctx->kernel_cells[mode] += total_cells;
            }
            if ((long long)used > runs_capacity) {          // rare: paths with very short runs
                if (attempt == 1) { c4h::set_error("traceback runs exceed their worst-case buffer"); return -1; }
                runs_capacity = ops_total;
                continue;
            }
            runs.resize(used);
            if (d_runs_out.download(runs.data(), used, s)) return -1;
            HIP_OK(c4_stream_sync(s));
            if (span == 2) {                                 // END cells back into the callers' matrices
                std::vector<int> host(span_total);
                if (d_span.download(host.data(), span_total, s)) return -1;
                HIP_OK(c4_stream_sync(s));
                for (int x = 0; x < n; x++) {
                    const long long cnt = (long long)(jobs[x].Q + 1) * (jobs[x].T + 1) * span_cs;
                    memcpy(specs[order[x]].span_out, host.data() + jobs[x].span_off, sizeof(int) * cnt);
                }
            }
            break;
        }
        lap("runs downloaded");
        std::atomic<int> overflow{0};
        parallel_for(n, 32768, [&](long long first, long long last) {
            for (long long x = first; x < last; x++) {
                JobOut &o = out[order[x]];
                o.res = res[x];
                if (res[x].flags & FLAG_OPS_OVERFLOW) { overflow = 1; continue; }
                if (mode == MODE_PATH) o.runs.assign_reversed(runs.data() + res[x].ops_off, res[x].n_ops);   // the walk wrote END -> START
                else o.runs.n = 0;
                if (mode == MODE_CKPT) o.vsa.assign(vsa.begin() + jobs[x].vsa_off, vsa.begin() + jobs[x].vsa_off + res[x].n_vsa);
                else o.vsa.clear();
                if (mode == MODE_CKPT && jobs[x].ckpt_off >= 0) {
                    const long long ck = (long long)jobs[x].cp_count * ki->max_at * (jobs[x].Q + 1) * ki->n_states * ki->cs;
                    o.checkpoints.assign(dump.begin() + jobs[x].ckpt_off, dump.begin() + jobs[x].ckpt_off + ck);
                } else {
                    o.checkpoints.clear();
                }
            }
        });
        if (overflow) { c4h::set_error("traceback path longer than its buffer"); return -1; }
        return 0;
    }
};

// ---- Optimal_find_path over a batch (optimal.c:368-413) ---------------------------------------------------------
struct Segment {                 // one Viterbi_SubAlignment (viterbi.c:482-496) in path order
    c4gpu_region region;
    int first_state;
    int final_cell[CELL_MAX];
    bool needs_checkpoints;      // Viterbi_use_reduced_space(vsa->region): recurse (optimal.c:203)
};

struct PairPlan {
    bool active = false, reduced = false;
    c4gpu_score region_score = 0;
    c4gpu_region ar;
    int end_from = 0;            // the state END was entered from in the region pass's best end cell (0: that pass did not say)
    std::vector<Segment> segs;   // reduced-space: the flattened vsa_list
};

bool model_is_global(const c4gpu_model *m) {     // C4_Model_is_global, c4.c:1959
    return m->start_scope == C4GPU_SCOPE_CORNER && m->end_scope == C4GPU_SCOPE_CORNER;
}

// Optimal_find_path_reduced_space for ONE pair exactly as the reference sequences it (optimal.c:160-345):
// every Viterbi call is its own launch and each sub-DP receives the final cell the previous one actually
// produced.  Slow path: only used when the batched prediction of those cells fails its verification.
struct SeqVsa { c4gpu_region region; int first_state; int final_cell[CELL_MAX]; };

int sequential_recur(Engine &eng, const ResidentSeqs &seqs, int pair, int dpmemory_mb, const c4gpu_region &region,
                     int first_state, const int *first_cell, int final_state, int *final_cell_out,
                     c4gpu_score *score_out, std::vector<SeqVsa> &leaves) {
    const c4gpu_model *m = eng.model;
    JobSpec js;
    js.pair = pair; js.region = region; js.first_state = first_state; js.final_state = final_state;
    memcpy(js.first_cell, first_cell, sizeof js.first_cell);
    js.cp_count = c4h::checkpoint_rows(m, &region, dpmemory_mb);
    std::vector<JobOut> outs;
    if (eng.run(seqs, MODE_CKPT, true, std::vector<JobSpec>(1, js), outs)) return -1;
    *score_out = outs[0].res.score;
    memcpy(final_cell_out, outs[0].res.final_cell, sizeof(int) * CELL_MAX);
    std::vector<SeqVsa> sub;
    for (int v = (int)outs[0].vsa.size() - 1; v >= 0; v--) {          // path order
        const DevVsa &dv = outs[0].vsa[v];
        SeqVsa sv;
        sv.region = c4gpu_region{dv.qs, dv.ts, dv.ql, dv.tl};
        sv.first_state = dv.first_state;
        memcpy(sv.final_cell, dv.final_cell, sizeof sv.final_cell);
        sub.push_back(sv);
    }
    for (size_t k = 0; k < sub.size(); k++) {
        if (c4h::use_reduced_space(m, &sub[k].region, dpmemory_mb)) {
            const int *sub_first = k ? sub[k - 1].final_cell : first_cell;
            const int sub_final_state = (k + 1 < sub.size()) ? sub[k + 1].first_state : final_state;
            c4gpu_score dummy;
            if (sequential_recur(eng, seqs, pair, dpmemory_mb, sub[k].region, sub[k].first_state, sub_first,
                                 sub_final_state, sub[k].final_cell, &dummy, leaves)) return -1;
        } else {
            leaves.push_back(sub[k]);
        }
    }
    return 0;
}

int sequential_reduced_path(Engine &eng, const ResidentSeqs &seqs, int pair, int dpmemory_mb,
                            const c4gpu_region &ar, c4gpu_alignment *a) {
    const c4gpu_model *m = eng.model;
    int zero[CELL_MAX] = {0}, final_cell[CELL_MAX];
    std::vector<SeqVsa> leaves;
    c4gpu_score score = 0;
    if (sequential_recur(eng, seqs, pair, dpmemory_mb, ar, m->start_state, zero, m->end_state, final_cell, &score,
                         leaves)) return -1;
    c4gpu_alignment_clear(a);
    a->score = score; a->region = ar; a->valid = 1;
    int cap = 0;
    for (size_t k = 0; k < leaves.size(); k++) {                      // Optimal_compute_subalignments
        JobSpec js;
        js.pair = pair; js.region = leaves[k].region; js.first_state = leaves[k].first_state;
        memcpy(js.first_cell, k ? leaves[k - 1].final_cell : zero, sizeof js.first_cell);
        js.final_state = (k + 1 < leaves.size()) ? leaves[k + 1].first_state : m->end_state;
        std::vector<JobOut> outs;
        if (eng.run(seqs, MODE_PATH, true, std::vector<JobSpec>(1, js), outs)) return -1;
        memcpy(leaves[k].final_cell, outs[0].res.final_cell, sizeof(int) * CELL_MAX);   // optimal.c:243,301
        for (uint32_t r : outs[0].runs) c4h::alignment_add(a, &cap, (int)(r >> 24), (int)(r & 0xffffff));
    }
    return 0;
}

// Steps 3 and 4 of find_path_batch for the pairs in `red` without the host in between (see fuse_expand_kernel): one
// checkpoint launch, one sub-alignment launch, one stitch; done[i] = 1 for every pair whose alignment was completed
// here.  The others (and every pair when the route does not apply) are left untouched for the host route.
// unfinished[pair]: the checkpoint pass's own result (score, final cell, sub-alignment list) of every pair the route did not
// finish, so that the host route does not run that pass again for them (a handful of whole-rectangle checkpoint jobs on a
// handful of waves takes as long as thousands: 433 ms for 11 chance alignments across 1 kb x 93 kb).
int fused_reduced_paths(Engine &eng, const ResidentSeqs &seqs, const std::vector<int> &red, const std::vector<PairPlan> &plan,
                        int dpmemory_mb, c4gpu_alignment *alignments, std::vector<char> &done, std::map<int, JobOut> &unfinished) {
    const bool trace = c4cfg::has(c4cfg::TRACE);
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "c4gpu trace:   fused: %-22s at %.3f ms\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const c4gpu_model *m = eng.model;
    c4gpu_ctx *ctx = eng.ctx;
    const int n = (int)red.size();
    if (!n || eng.pair_sub) return 0;
    if (c4cfg::is(c4cfg::FUSED, 0)) return 0;
    const int wpe_env = c4cfg::num(c4cfg::WPE, 0);
    long long worst = 0;
    for (int i : red) worst = std::max(worst, (long long)plan[i].ar.query_length + plan[i].ar.target_length);
    const bool cont_free = eng.cont_free_ok(worst);              // the sub-alignments lie inside their pair's region
    const KernelInfo *kc = get_kernel(eng.family, MODE_CKPT, true, cont_free, false, wpe_env, false, 0);
    const KernelInfo *kp = get_kernel(eng.family, MODE_PATH, true, cont_free, false, wpe_env, false, 0);
    if (!kc) kc = get_kernel(eng.family, MODE_CKPT, true, false, false, wpe_env, false, 0);
    if (!kp) kp = get_kernel(eng.family, MODE_PATH, true, false, false, wpe_env, false, 0);
    if (!kc || !kp) return 0;
    // the packed 16-bit checkpoint kernel (c4_ckpt16_kernel.h: two jobs per lane) for every job whose scores, checkpoint
    // payloads and intron lengths fit its halves — in its rooted form (the component of the state the path's END is entered
    // from: one strand of est2genome) where the region pass reported that state, else over every inner state;
    // C4GPU_CK16=0: never, 2..8: one shape whatever the jobs (tests, measurement), C4GPU_CK16_ROOT=0: never the rooted form (read on every
    // call: a test switches them)
    const int ck_env = c4cfg::num(c4cfg::CK16, 1);
    const bool ck_root_env = !(c4cfg::is(c4cfg::CK16_ROOT, 0));
    const bool ck16_on = ck_env > 0 && cont_free && eng.pk16_params_ok && seqs.tdense_n > 0;      // (dense target codes: Prof16)
    const KernelInfo *kc16 = ck16_on ? get_kernel_ck16(eng.family, 0, false) : nullptr;
    const KernelInfo *kc16r = (ck16_on && ck_root_env) ? get_kernel_ck16(eng.family, ck_env == 8 ? 0 : ck_env - 1, true) : nullptr;   // 1: chosen below, 8: variant 0
    const int ck16_tmax = c4cfg::num(c4cfg::CK16_TMAX, 0x7fffffff);      // test hook
    hipStream_t s = ctx->stream;
    const c4h::MemRule rule{m->max_query_advance, m->max_target_advance, m->n_states, m->total_shadow_designations};
    // -- the checkpoint jobs: the rooted packed kernel's first (root by root), then the packed kernel's, then the 32-bit kernel's,
    // each group longest first (persistent waves pull from the queue head)
    std::vector<int> order(n), cpn(n);
    std::vector<char> group(n, 0);                   // 2: packed, rooted; 1: packed; 0: 32-bit
    std::iota(order.begin(), order.end(), 0);
    auto cells = [&](int x) { const c4gpu_region &r = plan[red[x]].ar; return (long long)(r.query_length + 1) * (r.target_length + 1); };
    auto root_of = [&](int x) { return plan[red[x]].end_from; };
    int count_g[3] = {0, 0, 0};
    for (int x = 0; x < n; x++) {
        const c4gpu_region &r = plan[red[x]].ar;
        cpn[x] = c4h::checkpoint_rows(m, &r, dpmemory_mb);
        if (!kc16) continue;
        // payload ((row x states) + state) x max_target_advance + k in 16 bits; checkpoint columns max_target_advance apart at least
        const bool ok = eng.pk16_fits(r.query_length, r.target_length) && r.target_length <= ck16_tmax &&
                        ((long long)r.query_length + 2) * kc16->n_states * kc16->max_at <= 65535 && cpn[x] >= 1 &&
                        r.target_length / (cpn[x] + 1) >= kc16->max_at;
        group[x] = !ok ? 0 : (kc16r && root_of(x) > 1) ? 2 : 1;
        count_g[(int)group[x]]++;
    }
    for (int g = 2; g >= 1; g--)
        if (count_g[g] == 1) {                          // a lone packed job gains nothing: the group below takes it
            for (int x = 0; x < n; x++) if (group[x] == g) group[x] = (char)(g - 1);
            count_g[g - 1]++; count_g[g] = 0;
        }
    if (!count_g[1] && !count_g[2]) { kc16 = nullptr; kc16r = nullptr; }
    const int n16r = count_g[2], n16a = count_g[1], n16 = n16r + n16a;
    if (kc16r && n16r && ck_env == 1) {
        // the shape by the strips of 256 rows the rooted jobs have: four (two) cooperating waves per pair of jobs where the
        // jobs fill them -- the launch then lasts as long as its work, not as its longest job's strips one after the other
        // (north-star batch, two lanes: 481 -> 446 ms per step; profiles/r04_ck16_sweep.log) --, one wave per pair of short queries
        long long strips = 0;
        int rows_max = 0;
        for (int x = 0; x < n; x++)
            if (group[x] == 2) {
                strips += (plan[red[x]].ar.query_length + 1 + 255) / 256;
                rows_max = std::max(rows_max, plan[red[x]].ar.query_length + 1);
            }
        // (the four-wave shape at three waves per SIMD, 168 registers: 65 -> 52 ms per launch, step 436 -> 430 ms;
        // profiles/r04_ck16_w3_sweep.log)
        kc16r = get_kernel_ck16(eng.family, strips >= 3LL * n16r ? 8 : strips >= 2LL * n16r ? 5 : 0, true);
        // regions of 1 025 .. 1 152 rows are five strips of 256 -- a second round for one wave of four -- and three strips of
        // 384: the six-rows-per-lane shape on three waves takes them in one round (cDNAs of 1.1 kb)
        if (rows_max > 1024 && rows_max <= 3 * 384 && strips >= 4LL * n16r && get_kernel_ck16(eng.family, 6, true))
            kc16r = get_kernel_ck16(eng.family, 6, true);
    }
    if (n16 && eng.ensure_ss16(seqs)) return -1;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (group[a] != group[b]) return group[a] > group[b];
        if (group[a] == 2 && root_of(a) != root_of(b)) return root_of(a) < root_of(b);
        return cells(a) > cells(b);
    });
    if (c4cfg::has(c4cfg::TRACE))
        fprintf(stderr, "c4gpu trace:   fused: kernels %s, %s\nc4gpu trace:   fused: packed checkpoint kernels %s for %d, %s for %d of %d jobs\n",
                kc->name, kp->name, (kc16r && n16r) ? kc16r->name : "-", n16r, (kc16 && n16a) ? kc16->name : "-", n16a, n);
    std::vector<DevJob> &jobs = eng.hf_jobs;
    jobs.resize(n);
    long long vsa_total = 0, max_ckpt = 0, max_T = 0, ckpt_cells = 0, max_ckpt16 = 0, max_T16 = 0;
    bool carry = false, carry16 = false;
    int bnd16 = 0;
    for (int x = 0; x < n; x++) {
        const c4gpu_region &r = plan[red[order[x]]].ar;
        DevJob &j = jobs[x];
        memset(&j, 0, sizeof j);
        j.pair = red[order[x]]; j.q0 = r.query_start; j.t0 = r.target_start; j.Q = r.query_length; j.T = r.target_length;
        j.first_state = m->start_state; j.final_state = m->end_state;
        j.cp_count = cpn[order[x]];
        j.root = x < n16r ? root_of(order[x]) : 0;
        int tb = 0;
        while ((1LL << tb) <= j.T) tb++;
        j.tshift = tb;
        j.ckpt_off = -1; j.seed_off = -1;
        j.vsa_off = (int)vsa_total;
        vsa_total += j.cp_count + 1;
        ckpt_cells += (long long)(j.Q + 1) * (j.T + 1);
        if (x < n16) {
            const KernelInfo *k16 = x < n16r ? kc16r : kc16;
            max_ckpt16 = std::max(max_ckpt16, (long long)j.cp_count * k16->max_at * (j.Q + 1) * (x < n16r ? k16->ckw_root : k16->ckw));
            max_T16 = std::max<long long>(max_T16, j.T);
            bnd16 = std::max(bnd16, k16->bnd);
            if ((j.Q + 1 + 64 * k16->R - 1) / (64 * k16->R) > 1) carry16 = true;
        } else {
            max_ckpt = std::max(max_ckpt, (long long)j.cp_count * kc->max_at * (j.Q + 1) * kc->n_states * kc->cs);
            max_T = std::max<long long>(max_T, j.T);
            if ((j.Q + 1 + 64 * kc->R - 1) / (64 * kc->R) > kc->waves) carry = true;
        }
    }
    if (vsa_total > 0x7fffffffLL) return 0;
    // the packed kernels' pairs: neighbours of the same kernel and root
    std::vector<int> pair_list;
    int pairs_r = 0, pairs_a = 0;
    for (int x = 0; x < n16;) {
        const int lim = x < n16r ? n16r : n16;
        const bool two = x + 1 < lim && jobs[x + 1].root == jobs[x].root;
        pair_list.push_back(x); pair_list.push_back(two ? x + 1 : -1);
        (x < n16r ? pairs_r : pairs_a)++;
        x += two ? 2 : 1;
    }
    auto grid_for = [&](const KernelInfo *ki, long long jobs_n, long long bytes_per_wave) -> long long {
        int blocks_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, ki->func, 64 * ki->waves, 0) != hipSuccess || blocks_per_cu < 1)
            blocks_per_cu = 1;
        long long grid = std::min<long long>(jobs_n, (long long)blocks_per_cu * ctx->prop.multiProcessorCount);
        const long long budget = (long long)(ctx->prop.totalGlobalMem / 4);
        if (bytes_per_wave * grid > budget) grid = std::max<long long>(1, budget / std::max<long long>(1, bytes_per_wave));
        return grid;
    };
    int zero = 0;
    unsigned long long zero64 = 0;
    {
        // up to three launches on the lane's stream, one after the other: they share the carry rows and the checkpoint slabs
        const int zeros[3] = {0, 0, 0};
        const int n32 = n - n16;
        const long long bnd_per_wave = (2 * ((carry ? max_T : 0) + 1) + 1) * (long long)std::max(kc->bnd, 1);
        const long long grid = n32 ? grid_for(kc, n32, bnd_per_wave * 4 + max_ckpt * 4) : 0;
        const long long bnd16_per_wave = n16 ? (2 * ((carry16 ? max_T16 : 0) + 1) + 1) * (long long)std::max(bnd16, 1) : 0;
        const long long grid16r = pairs_r ? grid_for(kc16r, pairs_r, bnd16_per_wave * 4 + 2 * max_ckpt16 * 4) : 0;
        const long long grid16a = pairs_a ? grid_for(kc16, pairs_a, bnd16_per_wave * 4 + 2 * max_ckpt16 * 4) : 0;
        const long long grid16 = std::max(grid16r, grid16a);
        if (eng.d_fjobs.upload(jobs.data(), n, s) || eng.d_fres.alloc(n) || eng.d_queue.upload(zeros, 3, s) ||
            eng.d_pairs.upload(pair_list.data(), pair_list.size(), s) ||
            eng.d_bnd.alloc(std::max(bnd_per_wave * grid, bnd16_per_wave * grid16)) || eng.d_fvsa.alloc(vsa_total) ||
            eng.d_ckpt.alloc(std::max(max_ckpt * grid, 2 * max_ckpt16 * grid16)) || eng.d_ckpt_dump.alloc(1))
            return -1;
        LaunchArgs a;
        memset(&a.scratch, 0, sizeof a.scratch);
        a.kp = eng.kparams.p; a.seqs = seqs.dev;
        a.seqs.sub_colptr = nullptr; a.seqs.sub_rows = nullptr; a.seqs.span_in = nullptr; a.seqs.span_out = nullptr;
        a.seqs.seed = nullptr;
        a.seqs.ss16 = seqs.ss16_built ? seqs.ss16.p : nullptr;
        a.vsas = eng.d_fvsa.p; a.ops = nullptr; a.stream = s;
        a.scratch.bnd = eng.d_bnd.p;
        a.scratch.ckpt_dump = eng.d_ckpt_dump.p;
        if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev0, s));
        // the packed kernels index the whole job / result arrays through their pair lists
        a.jobs = eng.d_fjobs.p; a.n_jobs = n16; a.results = eng.d_fres.p;
        a.scratch.bnd_stride = bnd16_per_wave; a.scratch.carry = carry16 ? 1 : 0;
        a.scratch.ckpt = eng.d_ckpt.p; a.scratch.ckpt_stride = max_ckpt16;
        if (pairs_r || pairs_a) {
            a.seqs.sub_rows = reinterpret_cast<const int *>(seqs.tcode_dense.p);       // the packed pass's dense target codes and code
            a.seqs.sub_colptr = reinterpret_cast<const int *>(seqs.tdense.p);          // table (Prof16, c4_ckpt16_kernel.h)
        }
        if (pairs_r) {
            a.queue = eng.d_queue.p; a.grid = (int)grid16r; a.aux = eng.d_pairs.p; a.n_aux = pairs_r;
            HIP_OK(kc16r->launch(a));
        }
        if (pairs_a) {
            a.queue = eng.d_queue.p + 1; a.grid = (int)grid16a; a.aux = eng.d_pairs.p + 2 * pairs_r; a.n_aux = pairs_a;
            HIP_OK(kc16->launch(a));
        }
        a.seqs.sub_rows = nullptr; a.seqs.sub_colptr = nullptr;
        if (n32) {
            a.jobs = eng.d_fjobs.p + n16; a.n_jobs = n32; a.results = eng.d_fres.p + n16; a.queue = eng.d_queue.p + 2; a.grid = (int)grid;
            a.aux = nullptr; a.n_aux = 0;
            a.scratch.bnd_stride = bnd_per_wave; a.scratch.carry = carry ? 1 : 0;
            a.scratch.ckpt = max_ckpt ? eng.d_ckpt.p : nullptr; a.scratch.ckpt_stride = max_ckpt;
            HIP_OK(kc->launch(a));
        }
        if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev1, s));
    }
    std::vector<DevResult> &res = eng.hf_res;
    res.resize(n);
    if (eng.d_fres.download(res.data(), n, s)) return -1;
    HIP_OK(c4_stream_sync(s));
    if (ctx->timing) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->kernel_ms[MODE_CKPT] += ms; ctx->kernel_launches[MODE_CKPT]++; ctx->kernel_cells[MODE_CKPT] += ckpt_cells;
    }
    lap("checkpoint pass");
    // -- the sub-alignment jobs, on the device
    std::vector<int> &first = eng.hf_first;
    first.assign(n + 1, 0);
    for (int x = 0; x < n; x++) first[x + 1] = first[x] + ((res[x].flags & FLAG_NO_END) ? 0 : res[x].n_vsa);
    const long long n_sub = first[n];
    if (!n_sub) return 0;
    std::vector<unsigned long long> stats(FUSE_STATS, 0);
    if (eng.d_ffirst.upload(first.data(), n + 1, s) || eng.d_fsub_jobs.alloc(n_sub) || eng.d_fflags.alloc(n) ||
        eng.d_fstats.upload(stats.data(), FUSE_STATS, s))
        return -1;
    if (eng.d_fflags.zero(n, s)) return -1;
    hipLaunchKernelGGL(fuse_expand_kernel, dim3(n), dim3(64), 0, s, eng.d_fjobs.p, eng.d_fvsa.p, eng.d_ffirst.p, n,
                       eng.d_fsub_jobs.p, rule, dpmemory_mb, kp->R, eng.d_fflags.p, eng.d_fstats.p,
                       (c4cfg::is(c4cfg::BYROOT, 0)) ? 0 : 1);
    HIP_OK(hipGetLastError());
    if (eng.d_fstats.download(stats.data(), FUSE_STATS, s)) return -1;
    HIP_OK(c4_stream_sync(s));
    lap("jobs listed");
    const long long max_runs = (long long)stats[FUSE_MAX_OPS_CAP], max_tb = (long long)stats[FUSE_MAX_TB];
    const long long sub_T = (long long)stats[FUSE_MAX_T], ops_total = (long long)stats[FUSE_OPS_TOTAL];
    const bool sub_carry = (long long)stats[FUSE_MAX_STRIPS] > kp->waves;
    const long long bnd_per_wave = (2 * ((sub_carry ? sub_T : 0) + 1) + 1) * (long long)std::max(kp->bnd, 1);
    const long long grid = grid_for(kp, n_sub, bnd_per_wave * 4 + max_tb * 4 + max_runs * 4);
    long long runs_capacity = std::min<long long>(ops_total, std::max<long long>(1 << 20, n_sub * 256));
    unsigned long long used = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (eng.d_fsub_res.alloc(n_sub) || eng.d_queue.upload(&zero, 1, s) || eng.d_runs_used.upload(&zero64, 1, s) ||
            eng.d_bnd.alloc(bnd_per_wave * grid) || eng.d_runs.alloc(max_runs * grid) || eng.d_runs_out.alloc(runs_capacity) ||
            eng.d_tb.alloc(max_tb * grid))
            return -1;
        LaunchArgs a;
        memset(&a.scratch, 0, sizeof a.scratch);
        a.kp = eng.kparams.p; a.seqs = seqs.dev; a.jobs = eng.d_fsub_jobs.p; a.n_jobs = (int)n_sub; a.results = eng.d_fsub_res.p;
        a.seqs.sub_colptr = nullptr; a.seqs.sub_rows = nullptr; a.seqs.span_in = nullptr; a.seqs.span_out = nullptr;
        a.seqs.seed = nullptr;
        a.vsas = nullptr; a.ops = nullptr; a.queue = eng.d_queue.p; a.grid = (int)grid; a.stream = s;
        a.scratch.bnd = eng.d_bnd.p; a.scratch.bnd_stride = bnd_per_wave; a.scratch.carry = sub_carry ? 1 : 0;
        a.scratch.tb = max_tb ? eng.d_tb.p : nullptr; a.scratch.tb_stride = max_tb;
        a.scratch.runs = eng.d_runs.p; a.scratch.runs_stride = max_runs;
        a.scratch.runs_out = eng.d_runs_out.p; a.scratch.runs_capacity = runs_capacity;
        a.scratch.runs_used = eng.d_runs_used.p;
        if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev0, s));
        HIP_OK(kp->launch(a));
        if (ctx->timing) HIP_OK(hipEventRecord(ctx->ev1, s));
        if (eng.d_runs_used.download(&used, 1, s)) return -1;
        HIP_OK(c4_stream_sync(s));
        if (ctx->timing) {
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            ctx->kernel_ms[MODE_PATH] += ms; ctx->kernel_launches[MODE_PATH]++;
        }
        if ((long long)used <= runs_capacity) break;
        if (attempt == 1) { c4h::set_error("traceback runs exceed their worst-case buffer"); return -1; }
        runs_capacity = ops_total;                           // rare: paths with very short runs
    }
    lap("sub-alignment pass");
    // -- verify + concatenate per pair
    if (eng.d_fout.alloc(2 * (size_t)std::max<unsigned long long>(used, 1)) || eng.d_fpairs.alloc(n) ||
        eng.d_runs_used.upload(&zero64, 1, s))
        return -1;
    hipLaunchKernelGGL(fuse_stitch_kernel, dim3((n + 63) / 64), dim3(64), 0, s, eng.d_fjobs.p, eng.d_fvsa.p, eng.d_ffirst.p, n,
                       eng.d_fsub_res.p, eng.d_runs_out.p, 1 + m->total_shadow_designations, eng.d_fflags.p, eng.d_runs_used.p,
                       eng.d_fout.p, eng.d_fpairs.p, n16, cell_strict() ? 1 : 0);
    HIP_OK(hipGetLastError());
    std::vector<FusePair> &fps = eng.hf_pairs;
    fps.resize(n);
    unsigned long long out_used = 0;
    if (eng.d_fpairs.download(fps.data(), n, s) || eng.d_runs_used.download(&out_used, 1, s)) return -1;
    HIP_OK(c4_stream_sync(s));
    std::vector<int> &out = eng.hf_out;
    out.resize(2 * (size_t)out_used);
    if (out_used) {
        if (eng.d_fout.download(out.data(), 2 * (size_t)out_used, s)) return -1;
        HIP_OK(c4_stream_sync(s));
    }
    lap("stitched + downloaded");
    int n_done = 0;
    for (int x = 0; x < n; x++) {
        if (fps[x].status != 0) continue;
        const int i = red[order[x]];
        c4gpu_alignment &a = alignments[i];
        a.score = res[x].score;
        a.region = plan[i].ar;
        a.valid = 1;
        a.n_ops = fps[x].count;
        if (a.n_ops) {
            a.op_transition = (int32_t *)malloc(sizeof(int32_t) * a.n_ops);
            a.op_length = (int32_t *)malloc(sizeof(int32_t) * a.n_ops);
            const int *src = out.data() + 2 * fps[x].off;
            for (int k = 0; k < a.n_ops; k++) { a.op_transition[k] = src[2 * k]; a.op_length[k] = src[2 * k + 1]; }
        }
        done[i] = 1;
        n_done++;
    }
    if (n_done < n) {
        std::vector<DevVsa> vsa((size_t)vsa_total);
        if (eng.d_fvsa.download(vsa.data(), vsa_total, s)) return -1;
        HIP_OK(c4_stream_sync(s));
        for (int x = 0; x < n; x++) {
            if (fps[x].status == 0) continue;
            JobOut o;
            o.res = res[x];
            o.runs.n = 0;
            o.packed = x < n16;
            if (!(res[x].flags & FLAG_NO_END)) o.vsa.assign(vsa.begin() + jobs[x].vsa_off, vsa.begin() + jobs[x].vsa_off + res[x].n_vsa);
            unfinished[red[order[x]]] = std::move(o);
        }
    }
    if (c4cfg::has(c4cfg::TRACE))
        fprintf(stderr, "c4gpu trace:   fused: %d of %d pairs finished on the device route, %lld sub-alignments\n", n_done, n, n_sub);
    lap("alignments built");
    return 0;
}

// FIND_REGION of whole rectangles in two passes (c4_viterbi_kernel.h, SEED): a score pass that also dumps the DP state
// every K columns, then region-start payload passes over windows of one dump interval each, walking left from the end
// cell until the payload is a real start.  Scores, end cells and starts are those of the one-pass kernel (every
// window cell is computed from the whole-rectangle pass's own values); the payload work shrinks from the whole target
// to the alignment's extent.  out[x] for pairs[x]: score, qe, te always; qs, ts where the score reaches thr(pair).
template <class Thr>
int windowed_region_pass(Engine &eng, const ResidentSeqs &seqs, const std::vector<int> &pairs,
                         const std::vector<PairPlan> &plan, Thr thr, int kshift, std::vector<DevResult> &out) {
    const int n = (int)pairs.size();
    auto nbits = [](int v) { int b = 0; while ((1LL << b) <= v) b++; return b; };
    std::vector<JobSpec> specs(n);
    std::vector<JobOut> outs;
    for (int x = 0; x < n; x++) { specs[x].pair = pairs[x]; specs[x].region = plan[pairs[x]].ar; }
    SeedPlan sp1;
    sp1.mode = 1; sp1.kshift = kshift;
    if (eng.run(seqs, MODE_SCORE, false, specs, outs, &sp1)) return -1;
    const long long seedw = sp1.seedw;                   // ints per dumped row: the score kernel's format (32-bit cells or Dump16)
    const int dc = sp1.dc;                               // dumped columns per dump (d*K - (dc - 1) .. d*K)
    out.assign(n, DevResult());
    // the windows of a pair follow each other inside one workgroup (viterbi_kernel_mw, SEED 2): ONE launch over the first
    // windows of all wanted pairs; a job walks left from the end cell, one dump interval per window, until its payload is a
    // real start or its hop budget is spent
    std::vector<int> want;
    for (int x = 0; x < n; x++) {
        out[x] = outs[x].res;
        out[x].qs = out[x].ts = 0;
        if (!outs[x].res.end_set || outs[x].res.score < thr(pairs[x])) continue;
        want.push_back(x);
    }
    const size_t wanted = want.size();
    // hop budget: enough windows for the longest way back any wanted pair can have (a path that starts in the first dump
    // interval: reverse strands of the north-star batch, 289 of 1 024 pairs at 12 hops -- the one-pass kernel that finished
    // them cost 127 ms per 1 024 pairs, the extra hops cost nothing: profiles/r03_step.md), at least 12, at most 64
    int need_hops = 12;
    for (int x : want) need_hops = std::max(need_hops, ((outs[x].res.te - 1) >> kshift) + 2);
    const int max_hops = c4cfg::num(c4cfg::WINDOW_HOPS, std::min(need_hops, 64));
    std::vector<int> open;                                       // pairs whose path runs back further than the hop budget
    std::vector<char> demoted(n, 0);                             // ... and pairs the packed route could not serve (below)
    long long windows = 0;
    // C4GPU_STRICT=1 (debugging): a packed pass that disagrees with the 32-bit kernels fails the call instead of handing the
    // pair to them; C4GPU_FORCE_CORNER_MISMATCH=k (test hook): every k-th wanted pair is treated as such a disagreement
    const bool strict = c4cfg::nonzero(c4cfg::STRICT);
    const int force_miss = c4cfg::num(c4cfg::FORCE_CORNER_MISMATCH, 0);
    if (sp1.fmt16) {
        // the packed windows compute the component of the state END was entered from (DevResult::last_srp of the score pass);
        // a pair without one goes to the one-pass 32-bit kernel, which needs none
        std::vector<int> keep;
        for (int x : want) {
            if (outs[x].res.last_srp > 1) { keep.push_back(x); continue; }
            if (strict) { c4h::set_error("windowed region pass: the score pass did not say where END was entered from"); return -1; }
            if (c4cfg::has(c4cfg::TRACE)) fprintf(stderr, "c4gpu trace:   pair %d: no root from the packed score pass, one-pass kernel\n", pairs[x]);
            open.push_back(x); demoted[x] = 1;
        }
        want.swap(keep);
    }
    if (!want.empty()) {
        std::vector<JobSpec> hs(want.size());
        SeedPlan sp2;
        sp2.mode = 2; sp2.kshift = kshift; sp2.hops = std::max(1, max_hops); sp2.fmt16 = sp1.fmt16;
        sp2.off.resize(want.size()); sp2.rows.resize(want.size()); sp2.base.resize(want.size());
        sp2.d.resize(want.size()); sp2.t0w.resize(want.size()); sp2.t0_base.resize(want.size());
        for (size_t h = 0; h < want.size(); h++) {
            const int x = want[h];
            const c4gpu_region &ar = plan[pairs[x]].ar;
            const int d = (outs[x].res.te - 1) >> kshift;
            const int t0w = d >= 1 ? (d << kshift) - (dc - 1) : 0;      // window column 0 = lattice column t0w
            hs[h].pair = pairs[x];
            hs[h].region = c4gpu_region{ar.query_start, ar.target_start + t0w, outs[x].res.qe, outs[x].res.te - t0w};
            hs[h].final_state = eng.model->end_state;
            if (sp1.fmt16) {
                // the packed windows compute the component of the state END was entered from, and end in that state
                // (the score pass reports it: DevResult::last_srp)
                hs[h].final_state = hs[h].root = outs[x].res.last_srp;
            }
            sp2.rows[h] = ar.query_length + 1;
            sp2.base[h] = sp1.off[x];
            sp2.off[h] = d >= 1 ? sp1.off[x] + (long long)(d - 1) * dc * (ar.query_length + 1) * seedw : -1;
            sp2.d[h] = d; sp2.t0w[h] = t0w; sp2.t0_base[h] = ar.target_start;
        }
        std::vector<JobOut> wouts;
        if (eng.run(seqs, MODE_REGION, false, hs, wouts, &sp2)) return -1;
        for (size_t h = 0; h < want.size(); h++) {
            const DevResult &r = wouts[h].res;
            DevResult &o = out[want[h]];
            if (!r.end_set || r.score != o.score || (force_miss > 0 && h % (size_t)force_miss == 0)) {
                // never seen outside the test hook; should a packed pass ever disagree with itself, the pair is the one-pass
                // 32-bit kernel's (the other pairs of the call keep their results)
                if (c4cfg::has(c4cfg::TRACE))
                    fprintf(stderr, "c4gpu trace:   pair %d: score pass %d at (%d, %d), window corner %d (set %d): one-pass kernel\n", pairs[want[h]],
                            o.score, o.qe, o.te, r.score, (int)r.end_set);
                if (strict) { c4h::set_error("windowed region pass: a window's corner cell differs from the score pass"); return -1; }
                open.push_back(want[h]); demoted[want[h]] = 1;
                continue;
            }
            windows += r.n_vsa;
            if (r.pad >= 0) { o.qs = r.qs; o.ts = r.ts; }
            else open.push_back(want[h]);
        }
    }
    if (!open.empty()) {
        // the one-pass kernel over their whole rectangles finishes them (same result: it is what the windows reproduce
        // piece by piece)
        std::vector<JobSpec> fs(open.size());
        for (size_t h = 0; h < open.size(); h++) { fs[h].pair = pairs[open[h]]; fs[h].region = plan[pairs[open[h]]].ar; }
        if (eng.run(seqs, MODE_REGION, false, fs, outs)) return -1;
        for (size_t h = 0; h < open.size(); h++) {
            const DevResult &r = outs[h].res;
            DevResult &o = out[open[h]];
            if (r.score != o.score || r.qe != o.qe || r.te != o.te) {
                if (c4cfg::has(c4cfg::TRACE))
                    fprintf(stderr, "c4gpu trace:   pair %d: score pass %d at (%d, %d), one-pass kernel %d at (%d, %d)\n", pairs[open[h]],
                            o.score, o.qe, o.te, r.score, r.qe, r.te);
                if (strict || !demoted[open[h]]) {
                    c4h::set_error("windowed region pass: the one-pass kernel disagrees with the score pass");
                    return -1;
                }
                // a pair the packed route gave up on: the one-pass 32-bit kernel's result stands
                o = r;
            }
            o.qs = r.qs; o.ts = r.ts;
            if (demoted[open[h]]) o.last_srp = 0;       // "root not known": the checkpoint pass takes its unrooted / 32-bit form
        }
    }
    const std::vector<int> &hops = open;
    const long long round = windows;
    if (wanted >= 64 && !force_miss) {
        const double rate = 1.0 - (double)hops.size() / (double)wanted;
        eng.ctx->window_rate = eng.ctx->window_rate < 0 ? rate : 0.5 * eng.ctx->window_rate + 0.5 * rate;
    }
    if (c4cfg::has(c4cfg::TRACE))
        fprintf(stderr, "c4gpu trace: windowed region pass: %d pairs, dumps every %d columns, %lld windows in one launch, %zu of %zu "
                "paths left to the one-pass kernel\n", n, 1 << kshift, round, hops.size(), wanted);
    return 0;
}

// subs (may be NULL): per-pair SubOpt, the `subopt` argument the reference hands to every Viterbi_calculate of
// the path (optimal.c:368-413); active (may be NULL): pairs to run, the others get no alignment.
// initial (may be NULL): the `region` argument of each pair's Optimal_find_path (default: the whole rectangle) — what
// GAM_Result_refine_alignment passes for --refine region (gam.c:618-640).
int find_path_batch(Engine &eng, const ResidentSeqs &seqs, int dpmemory_mb, c4gpu_score threshold,
                    c4gpu_alignment *alignments, const std::vector<const c4gpu_subopt *> *subs = nullptr,
                    const uint8_t *active = nullptr, const std::vector<c4gpu_score> *pair_thresholds = nullptr,
                    const c4gpu_region *initial = nullptr, bool own_all = true) {
    // own_all = false: one of two lanes working on the same arrays (find_path_lanes): only the entries of `active` are this
    // call's to touch, and the caller has cleared them
    const c4gpu_model *m = eng.model;
    const int n = seqs.n_pairs;
    // per-pair thresholds (GAM_get_query_threshold with --percent, gam.c:677-705): never below `threshold`
    const c4gpu_score base_threshold = threshold;
    auto thr = [&](int pair) {
        return (pair_thresholds && (*pair_thresholds)[pair] > base_threshold) ? (*pair_thresholds)[pair] : base_threshold;
    };
    struct SubScope {                // every eng.run below sees the pairs' blocked cells
        Engine &e;
        SubScope(Engine &e_, const std::vector<const c4gpu_subopt *> *s) : e(e_) { e.pair_sub = s; }
        ~SubScope() { e.pair_sub = nullptr; }
    } sub_scope(eng, subs);
    const bool trace = c4cfg::has(c4cfg::TRACE);
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "c4gpu trace: find_path_batch: %-28s at %.3f ms\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const bool strict_cells = cell_strict();
    std::vector<PairPlan> plan(n);
    std::vector<JobSpec> &specs = eng.fp_specs;
    std::vector<JobOut> &outs = eng.fp_outs;
    specs.clear();
    std::vector<int> owner;
    if (own_all) for (int i = 0; i < n; i++) memset(&alignments[i], 0, sizeof(c4gpu_alignment));
    // -- step 1: where the whole rectangle is too large for a traceback, find the region first
    std::vector<int> region_pairs;
    for (int i = 0; i < n; i++) {
        plan[i].ar = initial ? initial[i] : c4gpu_region{0, 0, seqs.qlen[i], seqs.tlen[i]};
        plan[i].active = !active || active[i];
        if (!plan[i].active) continue;
        if (c4h::use_reduced_space(m, &plan[i].ar, dpmemory_mb)) {
            plan[i].reduced = true;
            if (!model_is_global(m)) region_pairs.push_back(i);        // Optimal_find_region, optimal.c:135
        }
    }
    const size_t step1_total = region_pairs.size();
    double &hit_rate = eng.ctx->hit_rate[subs ? 1 : 0];
    // long targets under a local model, nothing blocked: the two-pass (windowed) form of the region pass
    std::vector<std::pair<int, DevResult>> region_done;
    {
        // a dump every 8 192 columns (measured on the north-star batch with 32-bit dumps: score pass 410 ms against 462 ms at 4 096,
        // windows 204 against 181 ms; 16 384: 408 and 272 ms), every 4 096 while some target of the call is too short for that
        // -- and where the packed score pass serves, whose 16-bit dump rows cost it 1.4 ms per launch more at 4 096 while the
        // windows save 9 (step on two lanes 442 -> 436 ms; 2 048: 438; profiles/r04_kshift_sweep.log)
        // (4 096 only where the packed pass will really run: a call with a pair that does not fit its 16 bits -- a long query, a
        // small --intronpenalty -- or with C4GPU_PK16=0 dumps 32-bit rows, for which 8 192 is the better interval)
        bool pk16_serves = eng.family == FAM_EST2GENOME && eng.pk16_params_ok && region_pairs.size() >= 2 &&
                           !(c4cfg::is(c4cfg::PK16, 0)) && get_kernel_pk16(eng.family, 1) != nullptr;
        for (size_t x = 0; x < region_pairs.size() && pk16_serves; x++)
            pk16_serves = eng.pk16_fits(plan[region_pairs[x]].ar.query_length, plan[region_pairs[x]].ar.target_length);
        int kshift_env = pk16_serves ? 12 : 13;
        for (int i : region_pairs)
            if (plan[i].ar.target_length < (4 << 13) && plan[i].ar.target_length >= (4 << 12)) kshift_env = 12;
        kshift_env = c4cfg::num(c4cfg::SEED_KSHIFT, kshift_env);
        const int kshift = std::max(2, std::min(kshift_env, 20));
        const bool off = c4cfg::is(c4cfg::WINDOWED, 0);
        const KernelInfo *k1 = get_kernel_mw(eng.family, MODE_SCORE, true, false, 4, false, 1);
        const KernelInfo *k2 = get_kernel_mw(eng.family, MODE_REGION, true, true, 4, false, 2);
        // where most alignments ran past the hop budget in the earlier batches of this context (chance alignments across
        // whole windows: all-against-all without a threshold), the one-pass kernel is the cheaper form
        const bool pays = eng.ctx->window_rate < 0 || eng.ctx->window_rate >= 0.5 || c4cfg::has(c4cfg::SEED_KSHIFT);
        if (!off && pays && !subs && eng.local && eng.local_exact && k1 && k2 && !(c4cfg::is(c4cfg::PACK, 0))) {
            std::vector<int> win_pairs, rest;
            for (int i : region_pairs) {
                const c4gpu_region &ar = plan[i].ar;
                const bool rows_ok = ar.query_length + 1 > 2 * 64 * k2->R;       // the cooperating-wave kernels' domain
                if (rows_ok && ar.target_length >= (4 << kshift) && ar.query_length < (1 << 20)) win_pairs.push_back(i);
                else rest.push_back(i);
            }
            if (!win_pairs.empty()) {
                std::vector<DevResult> wres;
                if (windowed_region_pass(eng, seqs, win_pairs, plan, thr, kshift, wres)) return -1;
                for (size_t x = 0; x < win_pairs.size(); x++) region_done.emplace_back(win_pairs[x], wres[x]);
                region_pairs.swap(rest);
            }
        }
    }
    // A pair whose best score is below the threshold ends here (optimal.c:144-145).  The score alone costs
    // 0.64 of a region pass (no region-start payload), so where few pairs reach the threshold — all-vs-all
    // runs, the last round of the sub-optimal loop — a FIND_SCORE pass goes first and only the survivors get
    // the region pass.  Same recurrence, same score (FIND_SCORE and FIND_REGION differ in payload only); the
    // choice follows the hit rate of the previous batches of this context, sampled on the first large one.
    {
        const bool sf_env = c4cfg::has(c4cfg::SCORE_FIRST);                // 0 never, 1 always, unset adaptive
        const bool can = threshold > C4GPU_IMPOSSIBLY_LOW_SCORE && region_pairs.size() >= 512;
        const size_t total = region_pairs.size();
        auto score_filter = [&](size_t first, size_t count, size_t *kept) -> int {
            specs.clear();
            for (size_t x = first; x < first + count; x++) { JobSpec s; s.pair = region_pairs[x]; s.region = plan[region_pairs[x]].ar; specs.push_back(s); }
            if (eng.run(seqs, MODE_SCORE, false, specs, outs)) return -1;
            *kept = 0;
            for (size_t x = 0; x < count; x++) {
                if (outs[x].res.score < thr(region_pairs[first + x])) plan[region_pairs[first + x]].active = false;
                else (*kept)++;
            }
            return 0;
        };
        bool score_first = false;
        size_t sampled = 0, kept = 0, all_kept = 0;
        if (can && sf_env) score_first = c4cfg::num(c4cfg::SCORE_FIRST, 0) != 0;
        else if (can && hit_rate >= 0) score_first = hit_rate < 0.3;
        else if (can) {
            // one device-filling launch costs about the same as a small one: sample that many pairs
            sampled = std::min<size_t>(8 * (size_t)eng.ctx->prop.multiProcessorCount, total);
            if (score_filter(0, sampled, &kept)) return -1;
            all_kept = kept;
            score_first = (double)kept / (double)sampled < 0.3;
        }
        if (score_first && sampled < total) {
            if (score_filter(sampled, total - sampled, &kept)) return -1;
            all_kept += kept;
        }
        if (score_first || sampled) {
            std::vector<int> survivors;
            for (int i : region_pairs) if (plan[i].active) survivors.push_back(i);
            region_pairs.swap(survivors);
        }
        (void)all_kept;
    }
    specs.clear();
    for (int i : region_pairs) { JobSpec s; s.pair = i; s.region = plan[i].ar; specs.push_back(s); }
    if (eng.run(seqs, MODE_REGION, false, specs, outs)) return -1;
    for (size_t x = 0; x < region_pairs.size(); x++) region_done.emplace_back(region_pairs[x], outs[x].res);
    for (const auto &pr : region_done) {
        PairPlan &p = plan[pr.first];
        const DevResult &r = pr.second;
        if (r.score < thr(pr.first)) { p.active = false; continue; }
        p.region_score = r.score;
        p.end_from = r.last_srp;                 // the windowed pass's packed score kernel reports it (c4_viterbi16_kernel.h); else 0
        // Viterbi_Data_finalise, viterbi.c:633-653 (curr_*_start are relative to the region the pass ran over)
        if (m->start_scope != C4GPU_SCOPE_QUERY) p.ar.query_start += r.qs;
        if (m->start_scope != C4GPU_SCOPE_TARGET) p.ar.target_start += r.ts;
        p.ar.query_length = r.qe - (m->start_scope != C4GPU_SCOPE_QUERY ? r.qs : 0);
        p.ar.target_length = r.te - (m->start_scope != C4GPU_SCOPE_TARGET ? r.ts : 0);
    }
    if (step1_total >= 64 && threshold > C4GPU_IMPOSSIBLY_LOW_SCORE) {
        size_t hits = 0;
        for (const auto &pr : region_done) hits += plan[pr.first].active ? 1 : 0;
        const double rate = (double)hits / (double)step1_total;
        hit_rate = hit_rate < 0 ? rate : 0.5 * hit_rate + 0.5 * rate;
    }
    lap("region pass done");
    // -- step 2: quadratic-space path wherever the (alignment) region fits (optimal.c:349-364, 382-390)
    specs.clear(); owner.clear();
    for (int i = 0; i < n; i++) {
        PairPlan &p = plan[i];
        if (!p.active) continue;
        if (p.reduced && c4h::use_reduced_space(m, &p.ar, dpmemory_mb)) continue;
        p.reduced = false;
        JobSpec s; s.pair = i; s.region = p.ar;
        specs.push_back(s); owner.push_back(i);
    }
    if (eng.run(seqs, MODE_PATH, false, specs, outs)) return -1;
    for (size_t x = 0; x < owner.size(); x++) {
        const int i = owner[x];
        const DevResult &r = outs[x].res;
        c4gpu_alignment &a = alignments[i];
        a.score = r.score;
        // Viterbi_Data_create_Alignment, viterbi.c:380-383
        a.region.query_start = plan[i].ar.query_start + r.qs;
        a.region.target_start = plan[i].ar.target_start + r.ts;
        a.region.query_length = r.qe - r.qs;
        a.region.target_length = r.te - r.ts;
        int cap = 0;
        for (uint32_t r : outs[x].runs) c4h::alignment_add(&a, &cap, (int)(r >> 24), (int)(r & 0xffffff));
        a.valid = 1;
    }
    lap("quadratic paths done");
    // -- step 3: reduced space: checkpoint passes, recursively (optimal.c:160-230,315-345)
    std::vector<int> red;
    for (int i = 0; i < n; i++)
        if (plan[i].active && plan[i].reduced) {
            Segment s;
            memset(&s, 0, sizeof s);
            s.region = plan[i].ar; s.first_state = m->start_state; s.needs_checkpoints = true;
            plan[i].segs.assign(1, s);
            red.push_back(i);
        }
    std::map<int, JobOut> ckpt_done;            // first checkpoint pass of the pairs the device route left over
    if (!red.empty() && !subs) {
        // the device route first: whatever it finishes leaves the list
        std::vector<char> done(n, 0);
        if (fused_reduced_paths(eng, seqs, red, plan, dpmemory_mb, alignments, done, ckpt_done)) return -1;
        std::vector<int> rest;
        for (int i : red) if (!done[i]) rest.push_back(i);
        red.swap(rest);
        lap("device route done");
    }
    std::vector<c4gpu_score> red_score(n, 0);
    std::vector<char> redo(n, 0), packed_pair(n, 0);        // packed_pair: predicted cells from the packed checkpoint pass
    for (const auto &kv : ckpt_done) packed_pair[kv.first] = kv.second.packed ? 1 : 0;
    bool first_round = true;
    for (;;) {
        struct Ref { int pair, seg; };
        std::vector<Ref> refs;
        specs.clear();
        for (int i : red) {
            std::vector<Segment> &sg = plan[i].segs;
            for (size_t k = 0; k < sg.size(); k++) {
                if (!sg[k].needs_checkpoints) continue;
                JobSpec s; s.pair = i; s.region = sg[k].region;
                s.first_state = sg[k].first_state;
                // optimal.c:204-213: first cell = final cell of the previous sub-alignment (or the zero cell),
                // final state = first state of the next one (or END)
                if (k > 0) memcpy(s.first_cell, sg[k - 1].final_cell, sizeof s.first_cell);
                s.final_state = (k + 1 < sg.size()) ? sg[k + 1].first_state : m->end_state;
                s.cp_count = c4h::checkpoint_rows(m, &s.region, dpmemory_mb);
                specs.push_back(s);
                refs.push_back(Ref{i, (int)k});
            }
        }
        if (specs.empty()) break;
        lap("checkpoint jobs listed");
        bool have_all = first_round && !ckpt_done.empty();
        for (size_t x = 0; x < refs.size() && have_all; x++) have_all = ckpt_done.count(refs[x].pair) != 0;
        if (have_all) {                              // the device route ran exactly these jobs: its results are this round's
            if (outs.size() < refs.size()) outs.resize(refs.size());
            for (size_t x = 0; x < refs.size(); x++) outs[x] = ckpt_done[refs[x].pair];
        } else if (eng.run(seqs, MODE_CKPT, true, specs, outs)) return -1;
        lap("checkpoint pass done");
        // expand from the back so that segment indices stay valid; the jobs of one pair are adjacent in the list and
        // pairs do not touch each other's segments
        std::vector<int> group;
        for (size_t x = 0; x < refs.size(); x++)
            if (!x || refs[x].pair != refs[x - 1].pair) group.push_back((int)x);
        group.push_back((int)refs.size());
        parallel_for((long long)group.size() - 1, 256, [&](long long g0, long long g1) {
          for (long long g = g0; g < g1; g++)
            for (int x = group[g + 1] - 1; x >= group[g]; x--) {
            std::vector<Segment> &sg = plan[refs[x].pair].segs;
            const int k = refs[x].seg;
            if (first_round) red_score[refs[x].pair] = outs[x].res.score;
            std::vector<Segment> children;
            children.reserve(outs[x].vsa.size());
            for (int v = (int)outs[x].vsa.size() - 1; v >= 0; v--) {     // path order = reverse of the list
                const DevVsa &dv = outs[x].vsa[v];
                Segment c;
                c.region = c4gpu_region{dv.qs, dv.ts, dv.ql, dv.tl};
                c.first_state = dv.first_state;
                memcpy(c.final_cell, dv.final_cell, sizeof c.final_cell);
                c.needs_checkpoints = c4h::use_reduced_space(m, &c.region, dpmemory_mb);
                children.push_back(c);
            }
            // optimal.c:214-217 passes vsa->final_cell as the buffer the recursive pass overwrites; siblings
            // scheduled in the same round used the old value: it must not have changed
            if (!first_round && !children.empty() &&
                !final_cell_equiv(children.back().final_cell, sg[k].final_cell, 1 + m->total_shadow_designations,
                                  sg[k].region.target_start + sg[k].region.target_length, packed_pair[refs[x].pair] != 0, strict_cells)) {
                // Who has used the old value?  Only a sibling whose own nested pass ran in this round seeded with it: the
                // segment right behind this one, if it is a checkpoint job of this round too.  Every other use lies ahead --
                // the sub-alignment pass of step 4 seeds segment k + 1 with the cell the nested pass has just written (what
                // optimal.c:283,301 does) and verifies every final cell against its prediction there -- so only that case
                // sends the pair down the call-by-call route (one launch per section: 50 launches of 2 ms for a chance
                // alignment across a 100 kb window, profiles/r05_wide_trace.md).
                const bool next_in_round = x + 1 < group[g + 1] && refs[x + 1].seg == k + 1;
                if (c4cfg::has(c4cfg::TRACE))
                    fprintf(stderr, "c4gpu trace: pair %d nested segment %d: final cell %d/%d after the nested pass, %d/%d predicted%s\n",
                            refs[x].pair, k, children.back().final_cell[0], children.back().final_cell[1], sg[k].final_cell[0], sg[k].final_cell[1],
                            next_in_round ? ": the next segment's nested pass used the old cell, sequential route" : "");
                if (next_in_round || c4cfg::nonzero(c4cfg::NESTED_REDO)) redo[refs[x].pair] = 1;
            }
            sg.erase(sg.begin() + k);
            sg.insert(sg.begin() + k, children.begin(), children.end());
            }
        });
        first_round = false;
    }
    lap("segments expanded");
    // -- step 4: the sub-alignments themselves (Optimal_compute_subalignments, optimal.c:266-313)
    specs.clear();
    struct Ref2 { int pair, seg; };
    std::vector<Ref2> refs2;
    std::vector<size_t> seg_first(red.size() + 1, 0);          // first sub-alignment job of each reduced-space pair
    for (size_t r = 0; r < red.size(); r++) seg_first[r + 1] = seg_first[r] + plan[red[r]].segs.size();
    std::vector<JobSpec> &sub_specs = eng.fp_sub_specs;
    if (sub_specs.size() != seg_first[red.size()]) sub_specs.resize(seg_first[red.size()]);
    refs2.resize(seg_first[red.size()]);
    parallel_for((long long)red.size(), 256, [&](long long r0, long long r1) {
        for (long long r = r0; r < r1; r++) {
            const int i = red[r];
            const std::vector<Segment> &sg = plan[i].segs;
            for (size_t k = 0; k < sg.size(); k++) {
                JobSpec s; s.pair = i; s.region = sg[k].region;
                s.first_state = sg[k].first_state;
                if (k > 0) memcpy(s.first_cell, sg[k - 1].final_cell, sizeof s.first_cell);
                s.final_state = (k + 1 < sg.size()) ? sg[k + 1].first_state : m->end_state;
                sub_specs[seg_first[r] + k] = s;
                refs2[seg_first[r] + k] = Ref2{i, (int)k};
            }
        }
    });
    lap("sub-alignment jobs listed");
    if (eng.run(seqs, MODE_PATH, true, sub_specs, outs)) return -1;
    lap("sub-alignment pass done");
    {
        std::vector<int> cap(n, 0);
        const int path_cs = 1 + m->total_shadow_designations;
        const bool force_seq = c4cfg::has(c4cfg::FORCE_SEQUENTIAL);    // test hook
        if (force_seq) for (int i : red) redo[i] = 1;
        // The reference threads the final cell of each sub-DP into the next one (optimal.c:283,301); we
        // predicted it from the checkpoint rows to run all sub-DPs in one launch.  Verify: up to and including
        // the first sub-alignment whose final cell differs from the prediction, the batch did what the
        // reference does; the ones after it were seeded with a cell the reference would not have used.
        struct Repair { int pair, next_seg; int seed[CELL_MAX]; };
        std::vector<Repair> repairs;
        std::vector<int> repairing(n, -1);                 // first sub-alignment to recompute, per pair
        std::mutex repairs_lock;
        parallel_for((long long)red.size(), 256, [&](long long r0, long long r1) {
          for (long long r = r0; r < r1; r++)
            for (size_t x = seg_first[r]; x < seg_first[r + 1]; x++) {
            const int i = refs2[x].pair, k = refs2[x].seg;
            c4gpu_alignment &a = alignments[i];
            std::vector<Segment> &sg = plan[i].segs;
            if (k == 0) {
                a.score = red_score[i];
                a.region = plan[i].ar;
                a.valid = 1;
            }
            if (redo[i] || repairing[i] >= 0) continue;     // redone below from the first stale sub-alignment
            for (uint32_t r : outs[x].runs) c4h::alignment_add(&a, &cap[i], (int)(r >> 24), (int)(r & 0xffffff));
            if (!final_cell_equiv(outs[x].res.final_cell, sg[k].final_cell, path_cs,
                                  sg[k].region.target_start + sg[k].region.target_length, packed_pair[i] != 0, strict_cells)) {
                if (c4cfg::has(c4cfg::TRACE))
                    fprintf(stderr, "c4gpu trace: pair %d sub-alignment %d: final cell %d/%d computed, %d/%d predicted\n", i,
                            k, outs[x].res.final_cell[0], outs[x].res.final_cell[1], sg[k].final_cell[0], sg[k].final_cell[1]);
                if (k + 1 < (int)sg.size()) {
                    Repair rp; rp.pair = i; rp.next_seg = k + 1;
                    memcpy(rp.seed, outs[x].res.final_cell, sizeof rp.seed);
                    std::lock_guard<std::mutex> hold(repairs_lock);
                    repairs.push_back(rp);
                    repairing[i] = k + 1;
                }
            }
            }
        });
        // the repair launches below take the pairs in list order
        std::sort(repairs.begin(), repairs.end(), [](const Repair &x, const Repair &y) { return x.pair < y.pair; });
        // Optimal_compute_subalignments (optimal.c:266-313) for the stale tails, all affected pairs in lock-step: one small
        // launch per sub-alignment that has to be recomputed, seeded with the cell its predecessor actually produced.  A
        // recomputed sub-alignment that ends in the cell the batch had PREDICTED for it re-joins the batch: the sub-alignments
        // behind it were seeded with exactly that cell, so what the batch computed for them is what the reference computes
        // (same inputs), up to the next one whose final cell differs -- where the next repair starts.  (Without this a miss
        // early in a wide region cost one launch per remaining section: 50 launches of 2 ms for a chance alignment across a
        // 100 kb window, profiles/r05_wide_trace.md; differences in a cell are mostly an intron start that the next match
        // state forgets.)  C4GPU_REPAIR_REJOIN=0: recompute every sub-alignment behind a miss (test hook).
        const bool rejoin = !(c4cfg::is(c4cfg::REPAIR_REJOIN, 0));
        std::vector<size_t> first_of(n, 0);
        for (size_t r = 0; r < red.size(); r++) first_of[red[r]] = seg_first[r];
        std::vector<JobOut> routs;
        long long repair_launches = 0, repaired = 0, rejoined = 0;
        while (!repairs.empty()) {
            specs.clear();
            for (const Repair &rp : repairs) {
                const std::vector<Segment> &sg = plan[rp.pair].segs;
                JobSpec s; s.pair = rp.pair; s.region = sg[rp.next_seg].region;
                s.first_state = sg[rp.next_seg].first_state;
                memcpy(s.first_cell, rp.seed, sizeof s.first_cell);
                s.final_state = (rp.next_seg + 1 < (int)sg.size()) ? sg[rp.next_seg + 1].first_state : m->end_state;
                specs.push_back(s);
            }
            if (eng.run(seqs, MODE_PATH, true, specs, routs)) return -1;
            repair_launches++; repaired += (long long)repairs.size();
            std::vector<Repair> next;
            for (size_t x = 0; x < repairs.size(); x++) {
                Repair rp = repairs[x];
                c4gpu_alignment &a = alignments[rp.pair];
                const std::vector<Segment> &sg = plan[rp.pair].segs;
                const int nseg = (int)sg.size();
                for (uint32_t r : routs[x].runs) c4h::alignment_add(&a, &cap[rp.pair], (int)(r >> 24), (int)(r & 0xffffff));
                memcpy(rp.seed, routs[x].res.final_cell, sizeof rp.seed);
                const int j = rp.next_seg;
                if (j + 1 >= nseg) continue;
                const bool joined = rejoin && final_cell_equiv(rp.seed, sg[j].final_cell, path_cs,
                                                               sg[j].region.target_start + sg[j].region.target_length, packed_pair[rp.pair] != 0, strict_cells);
                if (!joined) { rp.next_seg = j + 1; next.push_back(rp); continue; }
                rejoined++;
                // the batch's own results from j + 1 on, up to (and including) the next sub-alignment that misses its prediction
                for (int k = j + 1; k < nseg; k++) {
                    const JobOut &bo = outs[first_of[rp.pair] + k];
                    for (uint32_t r : bo.runs) c4h::alignment_add(&a, &cap[rp.pair], (int)(r >> 24), (int)(r & 0xffffff));
                    if (k + 1 < nseg && !final_cell_equiv(bo.res.final_cell, sg[k].final_cell, path_cs,
                                                          sg[k].region.target_start + sg[k].region.target_length, packed_pair[rp.pair] != 0, strict_cells)) {
                        rp.next_seg = k + 1;
                        memcpy(rp.seed, bo.res.final_cell, sizeof rp.seed);
                        next.push_back(rp);
                        break;
                    }
                }
            }
            repairs.swap(next);
        }
        if (repair_launches && c4cfg::has(c4cfg::TRACE))
            fprintf(stderr, "c4gpu trace: stale tails: %lld sub-alignments recomputed in %lld launches, %lld re-joined the batch\n",
                    repaired, repair_launches, rejoined);
    }
    lap("alignments assembled");
    for (int i : red)
        if (redo[i] && sequential_reduced_path(eng, seqs, i, dpmemory_mb, plan[i].ar, &alignments[i])) return -1;
    for (int i = 0; i < n; i++) {
        if (active && !active[i]) continue;
        c4gpu_alignment &a = alignments[i];
        if (a.valid && a.score < thr(i)) c4gpu_alignment_clear(&a);     // optimal.c:408-411
    }
    return 0;
}

// ---- two launch lanes ---------------------------------------------------------------------------------------------
// The passes of Optimal_find_path are persistent kernels that end in a tail: the last round of jobs leaves part of the
// device idle (the packed score pass of the north-star batch runs 2 048 lane pairs on 768 resident workgroups: 2.67
// rounds), the windows and the checkpoint pass likewise.  A large batch is therefore cut into two halves of equal work
// that walk through the passes on two streams from two host threads: the workgroups of one half's next pass start as the
// other half's tail drains (measured on the north-star batch: 928 -> 839 ms per step; four lanes: 1 036 ms).  Both lanes
// read the same resident sequences; each has its own stream, events, statistics and launch buffers.  C4GPU_LANES=1
// keeps one lane, =2 forces two.
struct SideLane {
    c4gpu_ctx ctx;
    Engine eng;
    ~SideLane() {
        if (ctx.stream) (void)hipStreamDestroy(ctx.stream);
        if (ctx.ev0) (void)hipEventDestroy(ctx.ev0);
        if (ctx.ev1) (void)hipEventDestroy(ctx.ev1);
    }
    int init(const c4gpu_ctx *main, const c4gpu_model *m, const c4gpu_params *p) {
        ctx.device = main->device; ctx.prop = main->prop; ctx.timing = main->timing;
        HIP_OK(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreate(&ctx.ev0));
        HIP_OK(hipEventCreate(&ctx.ev1));
        if (eng.init(&ctx, m, p)) return -1;
        HIP_OK(c4_stream_sync(ctx.stream));
        return 0;
    }
};

// would this call be cut in two?  (callers create the side lane only then)
bool lanes_wanted(const ResidentSeqs &seqs, const uint8_t *active, bool blocking) {
    const int env = c4cfg::num(c4cfg::LANES, 0);
    if (blocking || env == 1) return false;
    long long n_act = 0;
    double cells = 0;
    for (int i = 0; i < seqs.n_pairs; i++) {
        if (active && !active[i]) continue;
        n_act++;
        cells += (double)(seqs.qlen[i] + 1) * (double)(seqs.tlen[i] + 1);
    }
    if (env == 2) return n_act >= 2;
    // each half must still fill the device more than once with whole-rectangle jobs that take long enough to have a tail
    return n_act >= 2048 && cells / (double)n_act >= 2.0e6;
}

int find_path_lanes(Engine &eng, SideLane *side, const ResidentSeqs &seqs, int dpmemory_mb, c4gpu_score threshold,
                    c4gpu_alignment *alignments, const uint8_t *active = nullptr,
                    const std::vector<c4gpu_score> *pair_thresholds = nullptr, const c4gpu_region *initial = nullptr) {
    const int n = seqs.n_pairs;
    if (!side || !lanes_wanted(seqs, active, false))
        return find_path_batch(eng, seqs, dpmemory_mb, threshold, alignments, nullptr, active, pair_thresholds, initial);
    // equal work per lane: every pair goes to the lane with fewer first-pass cells so far
    std::vector<uint8_t> mask[2] = {std::vector<uint8_t>(n, 0), std::vector<uint8_t>(n, 0)};
    double load[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        if (active && !active[i]) continue;
        const int l = load[1] < load[0] ? 1 : 0;
        mask[l][i] = 1;
        load[l] += (double)((initial ? initial[i].query_length : seqs.qlen[i]) + 1) * (double)((initial ? initial[i].target_length : seqs.tlen[i]) + 1);
    }
    for (int i = 0; i < n; i++) memset(&alignments[i], 0, sizeof(c4gpu_alignment));
    side->ctx.timing = eng.ctx->timing;
    int r1 = 0;
    std::string err1;
    const int device = eng.ctx->device;
    // an exception (std::bad_alloc of a host vector) must neither leave the second thread nor skip its join
    std::thread second([&] {
        try {
            if (hipSetDevice(device) != hipSuccess) { r1 = -1; err1 = "hipSetDevice on the second lane"; return; }
            r1 = find_path_batch(side->eng, seqs, dpmemory_mb, threshold, alignments, nullptr, mask[1].data(), pair_thresholds, initial, false);
            if (r1) err1 = c4h::g_error;
        } catch (const std::exception &e) { r1 = -1; err1 = std::string("second launch lane: ") + e.what(); }
    });
    int r0 = 0;
    try {
        r0 = find_path_batch(eng, seqs, dpmemory_mb, threshold, alignments, nullptr, mask[0].data(), pair_thresholds, initial, false);
    } catch (const std::exception &e) { r0 = -1; c4h::set_error(std::string("first launch lane: ") + e.what()); }
    second.join();
    if (r0) return r0;
    if (r1) { c4h::set_error(err1); return r1; }
    return 0;
}

}  // namespace

// ---- the C ABI --------------------------------------------------------------------------------------------------
struct c4gpu_batch {
    c4gpu_ctx *ctx;
    c4gpu_model model;
    c4gpu_params params;
    Engine eng;
    ResidentSeqs seqs;
    std::vector<c4gpu_score> scores;
    std::vector<c4gpu_region> regions;
    std::vector<c4gpu_alignment> alignments;
    // the sub-optimal loop (c4gpu_batch_next_paths): one SubOpt per pair, pairs still in the loop
    std::vector<c4gpu_subopt *> subopts;
    std::vector<uint8_t> in_loop;
    std::vector<c4gpu_score> pair_thresholds;        // c4gpu_batch_set_thresholds; empty = none
    // engines of further models run on the same resident sequences (c4gpu_batch_viterbi_model): BSDP's derived
    // terminal / join / span models of the batch's model, keyed by the flattened model's bytes
    struct ExtraEngine { c4gpu_model model; Engine eng; };
    std::map<std::string, std::unique_ptr<ExtraEngine>> extra;
    std::unique_ptr<SideLane> side;                  // the second launch lane of large batches (find_path_lanes), made on first use
    void clear_loop() {
        for (c4gpu_subopt *so : subopts) c4gpu_subopt_destroy(so);
        subopts.clear(); in_loop.clear();
    }
};

// The next batch on its way to the device while the current one is aligned (c4gpu_stage_load on one thread, c4gpu_batch_run
// on another): its own stream (non-blocking: no implicit synchronisation with the streams the passes run on), its own
// engine for the parameter block the packed splice array is built from, page-locked host buffers and device arrays that the
// batch it is swapped into hands back for the batch after.
struct c4gpu_stage {
    c4gpu_ctx ctx;
    c4gpu_model model;
    c4gpu_params params;
    Engine eng;
    ResidentSeqs seqs;
    bool loaded = false;
    double load_ms = 0;
    ~c4gpu_stage() {
        if (ctx.stream) (void)hipStreamDestroy(ctx.stream);
        if (ctx.ev0) (void)hipEventDestroy(ctx.ev0);
        if (ctx.ev1) (void)hipEventDestroy(ctx.ev1);
    }
};

extern "C" {

int c4gpu_abi_version(void) { return C4GPU_ABI_VERSION; }
int c4gpu_config_reload(void) {
    (void)c4cfg::get();                       // (the once-flag is spent before the table is written again)
    c4cfg::load_from_environment();
    int n = 0;
    for (int k = 0; k < c4cfg::N_KEYS; k++) n += c4cfg::table().e[k].set ? 1 : 0;
    return n;
}
const char *c4gpu_last_error(void) { return c4h::g_error.c_str(); }

static std::atomic<int> g_warm_cancel{0};        // c4gpu_ctx_warm_cancel; reset by every c4gpu_ctx_create

c4gpu_ctx *c4gpu_ctx_create(int device_ordinal) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        c4h::set_error(std::string("no HIP device available (") + hipGetErrorString(e) +
                       "): libc4gpu has no CPU fallback");
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= count) { c4h::set_error("bad device ordinal"); return nullptr; }
    g_warm_cancel.store(0, std::memory_order_relaxed);       // a cancelled warm-up belongs to the context that was being left
    c4gpu_ctx *ctx = new c4gpu_ctx;
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_ordinal) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        c4h::set_error("cannot initialise the HIP device");
        delete ctx;
        return nullptr;
    }
    if (!strstr(ctx->prop.gcnArchName, "gfx950")) {
        c4h::set_error(std::string("device is ") + ctx->prop.gcnArchName + ", kernels are built for gfx950 only");
        delete ctx;
        return nullptr;
    }
    g_retired.set_cap_for(ctx->prop.totalGlobalMem);
    return ctx;
}

int c4gpu_model_device_family(const c4gpu_model *model) { return model_family(*model); }

int c4gpu_packed_route_fits(const c4gpu_model *model, const c4gpu_params *params, int32_t query_length, int32_t target_length) {
    std::unique_ptr<Engine> eng(new Engine);
    std::unique_ptr<KParams> kp(new KParams);
    if (eng->init_host(nullptr, model, params, *kp)) return -1;
    return (eng->pk16_params_ok && eng->family == FAM_EST2GENOME && eng->pk16_fits(query_length, target_length)) ? 1 : 0;
}

int c4gpu_loop_sections(const c4gpu_model *model, const c4gpu_params *params, int32_t *loop_transition, int32_t n_states) {
    std::unique_ptr<Engine> eng(new Engine);
    std::unique_ptr<KParams> kp(new KParams);
    if (eng->init_host(nullptr, model, params, *kp)) return -1;
    int found = 0;
    for (int s = 0; s < n_states; s++) {
        loop_transition[s] = s < 16 ? kp->loop_tr[s] : -1;
        found += loop_transition[s] >= 0;
    }
    return found;
}

int c4gpu_memrule_device(c4gpu_ctx *ctx, const c4gpu_model *model, int dpmemory_mb, const int32_t *query_length,
                         const int32_t *target_length, int32_t n, int32_t *reduced, int32_t *rows) {
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    DevBuf<int> dq, dt, dr, dw;
    hipStream_t s = ctx->stream;
    const c4h::MemRule rule{model->max_query_advance, model->max_target_advance, model->n_states, model->total_shadow_designations};
    if (dq.upload(query_length, n, s) || dt.upload(target_length, n, s) || dr.alloc(n) || dw.alloc(n)) return -1;
    hipLaunchKernelGGL(memrule_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rule, dpmemory_mb, dq.p, dt.p, n, dr.p, dw.p);
    HIP_OK(hipGetLastError());
    if (dr.download(reduced, n, s) || dw.download(rows, n, s)) return -1;
    HIP_OK(c4_stream_sync(s));
    return 0;
}

// Load the code objects a heuristic run touches first (sequence preparation, HSP extension, word scan; the SDP passes of
// every family) without launching anything: hipFuncGetAttributes resolves a kernel, which loads its translation unit's
// code object.  Meant for a background thread while the host still reads sequences (the drop-in: c4gpu_shim.c).
// c4gpu_ctx_warm_cancel(): a warm-up that is running (on whatever thread) returns before its next load, one that has not
// started loads nothing -- so that a caller that is about to leave can join its warm-up thread within one load (the drop-in's
// way out, integration/c4gpu_shim.c: no thread is inside the HIP runtime when the exit handlers run).
void c4gpu_ctx_warm_cancel(void) { g_warm_cancel.store(1, std::memory_order_relaxed); }
void c4gpu_ctx_warm(c4gpu_ctx *ctx) {
    if (!ctx || g_warm_cancel.load(std::memory_order_relaxed) || hipSetDevice(ctx->device) != hipSuccess) return;
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)encode_kernel);
    const c4sdp::SdpKernels *ks[4] = {c4sdp::sdp_kernels_affine(), c4sdp::sdp_kernels_protein2dna(), c4sdp::sdp_kernels_est2genome(),
                                      c4sdp::sdp_kernels_protein2genome()};
    for (const c4sdp::SdpKernels *k : ks) {
        if (g_warm_cancel.load(std::memory_order_relaxed)) break;
        (void)hipFuncGetAttributes(&a, k->rev_func);
        if (g_warm_cancel.load(std::memory_order_relaxed)) break;
        (void)hipFuncGetAttributes(&a, k->fwd_func);
    }
    (void)hipGetLastError();
}

void c4gpu_ctx_destroy(c4gpu_ctx *ctx) {
    if (!ctx) return;
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->sdp_arena) (void)hipFree(ctx->sdp_arena);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    g_retired.flush();
}

void c4gpu_ctx_set_stream(c4gpu_ctx *ctx, void *hip_stream) {
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->owns_stream = false;
    ctx->stream = (hipStream_t)hip_stream;
}

int c4gpu_ctx_sdp_reserve(c4gpu_ctx *ctx, int64_t bytes) {
    if (hipSetDevice(ctx->device) != hipSuccess) { c4h::set_error("c4gpu_ctx_sdp_reserve: cannot select the device"); return -1; }
    if (bytes <= 0) {
        ctx->sdp_arena_keep = false;
        if (ctx->sdp_arena) (void)hipFree(ctx->sdp_arena);
        ctx->sdp_arena = nullptr; ctx->sdp_arena_bytes = 0;
        return 0;
    }
    size_t free_b = 0, total_b = 0;
    if (dev_mem_info(&free_b, &total_b) != hipSuccess) { c4h::set_error("c4gpu_ctx_sdp_reserve: hipMemGetInfo failed"); return -1; }
    size_t want = std::min<size_t>((size_t)bytes, (size_t)((double)(free_b + ctx->sdp_arena_bytes) * 0.6));
    want &= ~(((size_t)1 << 16) - 1);                     // whole 64 KB chunks
    ctx->sdp_arena_keep = true;
    if (ctx->sdp_arena_bytes >= want) return 0;
    if (ctx->sdp_arena) (void)hipFree(ctx->sdp_arena);
    ctx->sdp_arena = nullptr; ctx->sdp_arena_bytes = 0;
    if (dev_malloc(&ctx->sdp_arena, want) != hipSuccess) {
        (void)hipGetLastError();
        ctx->sdp_arena = nullptr;
        c4h::set_error("c4gpu_ctx_sdp_reserve: allocation failed");
        return -1;
    }
    ctx->sdp_arena_bytes = want;
    return 0;
}

int c4gpu_ctx_own_stream(c4gpu_ctx *ctx) {
    if (ctx->owns_stream) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess) { c4h::set_error("c4gpu_ctx_own_stream: cannot select the device"); return -1; }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        c4h::set_error("c4gpu_ctx_own_stream: cannot create a stream");
        return -1;
    }
    ctx->stream = s;
    ctx->owns_stream = true;
    return 0;
}

int c4gpu_ctx_device_info(c4gpu_ctx *ctx, char *name, size_t name_len, int *n_cu, int64_t *mem_bytes) {
    if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = (int64_t)ctx->prop.totalGlobalMem;
    return 0;
}

int c4gpu_splice_predict(c4gpu_ctx *ctx, const c4gpu_params *params, const uint8_t *target, int32_t target_len,
                         int32_t *out[4]) {
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    c4gpu_pair pair{(const uint8_t *)"A", 1, target, target_len};
    ResidentSeqs seqs;
    if (seqs.build(ctx, FAM_EST2GENOME, params, &pair, 1)) return -1;
    for (int k = 0; k < 4; k++) {
        HIP_OK(hipMemcpyAsync(out[k], seqs.ss.p + (long long)k * seqs.dev.ss_stride, sizeof(int) * (size_t)target_len,
                              hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_OK(c4_stream_sync(ctx->stream));
    return 0;
}

}  // extern "C"

extern "C" int c4gpu_hsp_extend_batch(c4gpu_ctx *ctx, const c4gpu_params *params, int match_type, const c4gpu_pair *pairs,
                                      int32_t n_pairs, int32_t seedlen, int32_t dropoff, const c4gpu_hsp_seed *seeds,
                                      int32_t n_seeds, c4gpu_hsp *out);
extern "C" int c4gpu_hsp_extend_chains(c4gpu_ctx *ctx, const c4gpu_params *params, int match_type, const c4gpu_pair *pairs,
                                       int32_t n_pairs, int32_t seedlen, int32_t dropoff, const c4gpu_hsp_seed *seeds,
                                       int32_t n_seeds, const int32_t *chain, int32_t n_chains, const int32_t *horizon0,
                                       c4gpu_hsp *out) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        if (match_type < C4GPU_MATCH_DNA2DNA || match_type > C4GPU_MATCH_PROTEIN2DNA) { c4h::set_error("unknown match type"); return -1; }
        if (!n_seeds) return 0;
        const int aq = 1, at = match_type == C4GPU_MATCH_PROTEIN2DNA ? 3 : 1;
        std::vector<int> first((size_t)n_chains + 1, 0), order((size_t)n_seeds);
        for (int k = 0; k < n_seeds; k++) {
            const c4gpu_hsp_seed &sd = seeds[k];
            if (sd.pair < 0 || sd.pair >= n_pairs || sd.query_start < 0 || sd.target_start < 0 || chain[k] < 0 || chain[k] >= n_chains ||
                sd.query_start + seedlen * aq > pairs[sd.pair].query_len || sd.target_start + seedlen * at > pairs[sd.pair].target_len) {
                c4h::set_error("an HSP seed lies outside its pair or names no chain");
                return -1;
            }
            first[chain[k] + 1]++;
        }
        for (int c = 0; c < n_chains; c++) first[c + 1] += first[c];
        {
            std::vector<int> fill(first.begin(), first.end() - 1);
            for (int k = 0; k < n_seeds; k++) order[fill[chain[k]]++] = k;            // index order inside every chain
        }
        ResidentSeqs seqs;
        if (seqs.build(ctx, match_type == C4GPU_MATCH_PROTEIN2DNA ? FAM_UNGAPPED_P2D : FAM_UNGAPPED, params, pairs, n_pairs)) return -1;
        std::vector<HspJob> jobs(n_pairs);
        for (int i = 0; i < n_pairs; i++) jobs[i] = HspJob{seqs.qoff[i], seqs.toff[i], seqs.qlen[i], seqs.tlen[i]};
        DevBuf<HspJob> d_jobs;
        DevBuf<c4gpu_hsp_seed> d_seeds;
        DevBuf<c4gpu_hsp> d_out;
        DevBuf<int> d_submat, d_order, d_first, d_h0;
        const int32_t *mat = match_type == C4GPU_MATCH_DNA2DNA ? &params->dna_submat[0][0] : &params->protein_submat[0][0];
        hipStream_t s = ctx->stream;
        if (d_jobs.upload(jobs.data(), n_pairs, s) || d_seeds.upload(seeds, n_seeds, s) || d_out.alloc(n_seeds) ||
            d_submat.upload(mat, 24 * 24, s) || d_order.upload(order.data(), n_seeds, s) ||
            d_first.upload(first.data(), (size_t)n_chains + 1, s) || d_h0.upload(horizon0, n_chains, s)) return -1;
        const int block = 64, grid = std::min((n_chains + block - 1) / block, 65535);
        hipLaunchKernelGGL(hsp_chain_kernel, dim3(grid), dim3(block), 0, s, seqs.qcode.p, seqs.tcode.p, d_jobs.p, d_seeds.p, d_order.p,
                           d_first.p, n_chains, d_h0.p, d_submat.p, aq, at, seedlen, dropoff, d_out.p);
        HIP_OK(hipGetLastError());
        if (d_out.download(out, n_seeds, s)) return -1;
        HIP_OK(c4_stream_sync(s));
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_hsp_extend_chains: ") + e.what());
        return -1;
    }
}

extern "C" int c4gpu_hsp_extend_batch(c4gpu_ctx *ctx, const c4gpu_params *params, int match_type, const c4gpu_pair *pairs,
                                      int32_t n_pairs, int32_t seedlen, int32_t dropoff, const c4gpu_hsp_seed *seeds,
                                      int32_t n_seeds, c4gpu_hsp *out) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        if (match_type < C4GPU_MATCH_DNA2DNA || match_type > C4GPU_MATCH_PROTEIN2DNA) { c4h::set_error("unknown match type"); return -1; }
        if (!n_seeds) return 0;
        const int aq = 1, at = match_type == C4GPU_MATCH_PROTEIN2DNA ? 3 : 1;
        for (int k = 0; k < n_seeds; k++) {
            const c4gpu_hsp_seed &sd = seeds[k];
            if (sd.pair < 0 || sd.pair >= n_pairs || sd.query_start < 0 || sd.target_start < 0 ||
                sd.query_start + seedlen * aq > pairs[sd.pair].query_len || sd.target_start + seedlen * at > pairs[sd.pair].target_len) {
                c4h::set_error("an HSP seed lies outside its pair");
                return -1;
            }
        }
        // the coded arrays of a protein2dna batch are exactly what PROTEIN2DNA scoring reads (row of the codon at each
        // target position); the 1:1 matches use the plain residue rows
        ResidentSeqs seqs;
        if (seqs.build(ctx, match_type == C4GPU_MATCH_PROTEIN2DNA ? FAM_UNGAPPED_P2D : FAM_UNGAPPED, params, pairs, n_pairs)) return -1;
        std::vector<HspJob> jobs(n_pairs);
        for (int i = 0; i < n_pairs; i++) jobs[i] = HspJob{seqs.qoff[i], seqs.toff[i], seqs.qlen[i], seqs.tlen[i]};
        DevBuf<HspJob> d_jobs;
        DevBuf<c4gpu_hsp_seed> d_seeds;
        DevBuf<c4gpu_hsp> d_out;
        DevBuf<int> d_submat;
        const int32_t *mat = match_type == C4GPU_MATCH_DNA2DNA ? &params->dna_submat[0][0] : &params->protein_submat[0][0];
        hipStream_t s = ctx->stream;
        if (d_jobs.upload(jobs.data(), n_pairs, s) || d_seeds.upload(seeds, n_seeds, s) || d_out.alloc(n_seeds) ||
            d_submat.upload(mat, 24 * 24, s)) return -1;
        const int block = 64, grid = std::min((n_seeds + block - 1) / block, 65535);
        hipLaunchKernelGGL(hsp_extend_kernel, dim3(grid), dim3(block), 0, s, seqs.qcode.p, seqs.tcode.p, d_jobs.p, d_seeds.p,
                           n_seeds, d_submat.p, aq, at, seedlen, dropoff, d_out.p);
        HIP_OK(hipGetLastError());
        if (d_out.download(out, n_seeds, s)) return -1;
        HIP_OK(c4_stream_sync(s));
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_hsp_extend_batch: ") + e.what());
        return -1;
    }
}

static int viterbi_jobs(Engine &eng, const ResidentSeqs &seqs, int mode, const c4gpu_viterbi_job *jobs,
                        int32_t n_jobs, c4gpu_viterbi_result *results) {
    // jobs with and without continuation run different kernels (the model copy with CORNER scopes)
    for (int cont = 0; cont < 2; cont++) {
        std::vector<JobSpec> specs;
        std::vector<int> idx;
        for (int i = 0; i < n_jobs; i++) {
            if ((jobs[i].use_continuation != 0) != (cont != 0)) continue;
            JobSpec s;
            s.pair = jobs[i].pair; s.region = jobs[i].region;
            s.first_state = jobs[i].continuation.first_state; s.final_state = jobs[i].continuation.final_state;
            for (int l = 0; l < CELL_MAX; l++) s.first_cell[l] = jobs[i].continuation.first_cell[l];
            s.cp_count = jobs[i].checkpoint_count;
            s.dump_checkpoints = (mode == C4GPU_MODE_FIND_CHECKPOINTS);
            s.sub = jobs[i].subopt;
            s.span_in = jobs[i].start_cells; s.span_out = jobs[i].end_cells;
            specs.push_back(s); idx.push_back(i);
        }
        if ((mode == C4GPU_MODE_FIND_CHECKPOINTS || mode == C4GPU_MODE_FIND_REGION) && !specs.empty() &&
            ((mode == C4GPU_MODE_FIND_CHECKPOINTS) != (cont != 0))) {
            c4h::set_error("FIND_CHECKPOINTS runs with a continuation, FIND_REGION without (optimal.c:47-68)");
            return -1;
        }
        std::vector<JobOut> outs;
        if (eng.run(seqs, mode, cont != 0, specs, outs)) return -1;
        for (size_t x = 0; x < idx.size(); x++) {
            c4gpu_viterbi_result &r = results[idx[x]];
            const DevResult &d = outs[x].res;
            memset(&r, 0, sizeof r);
            r.score = d.score; r.query_start = d.qs; r.target_start = d.ts; r.query_end = d.qe; r.target_end = d.te;
            for (int l = 0; l < CELL_MAX; l++) r.final_cell[l] = d.final_cell[l];
            r.last_srp = d.last_srp;
            r.n_ops = 0;
            for (uint32_t run : outs[x].runs) r.n_ops += (int)(run & 0xffffff);
            if (r.n_ops) {
                r.ops = (int32_t *)malloc(sizeof(int32_t) * r.n_ops);
                int k = 0;
                for (uint32_t run : outs[x].runs)
                    for (uint32_t c = 0; c < (run & 0xffffff); c++) r.ops[k++] = (int)(run >> 24);
            }
            if (!outs[x].checkpoints.empty()) {
                r.checkpoints = (c4gpu_score *)malloc(sizeof(int) * outs[x].checkpoints.size());
                memcpy(r.checkpoints, outs[x].checkpoints.data(), sizeof(int) * outs[x].checkpoints.size());
            }
        }
    }
    return 0;
}


extern "C" {

int c4gpu_viterbi_batch(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params, int mode,
                        const c4gpu_pair *pairs, int32_t n_pairs, const c4gpu_viterbi_job *jobs, int32_t n_jobs,
                        c4gpu_viterbi_result *results) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        Engine eng;
        ResidentSeqs seqs;
        if (eng.init(ctx, model, params) || seqs.build(ctx, eng.family, params, pairs, n_pairs)) return -1;
        return viterbi_jobs(eng, seqs, mode, jobs, n_jobs, results);
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_viterbi_batch: ") + e.what());
        return -1;
    }
}

int c4gpu_batch_viterbi(c4gpu_batch *b, int mode, const c4gpu_viterbi_job *jobs, int32_t n_jobs,
                        c4gpu_viterbi_result *results) {
    try {
        if (hipSetDevice(b->ctx->device) != hipSuccess) return -1;
        return viterbi_jobs(b->eng, b->seqs, mode, jobs, n_jobs, results);
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_viterbi: ") + e.what());
        return -1;
    }
}

int c4gpu_batch_viterbi_model(c4gpu_batch *b, const c4gpu_model *model, int mode, const c4gpu_viterbi_job *jobs,
                              int32_t n_jobs, c4gpu_viterbi_result *results) {
    try {
        if (hipSetDevice(b->ctx->device) != hipSuccess) return -1;
        const std::string key(reinterpret_cast<const char *>(model), sizeof(c4gpu_model));
        auto it = b->extra.find(key);
        if (it == b->extra.end()) {
            std::unique_ptr<c4gpu_batch::ExtraEngine> e(new c4gpu_batch::ExtraEngine);
            e->model = *model;
            if (e->eng.init(b->ctx, &e->model, &b->params)) return -1;
            // the resident arrays were prepared for the batch's own model: the other model must read the same ones
            if (family_is_p2d(e->eng.family) != family_is_p2d(b->eng.family) ||
                (family_has_splice(e->eng.family) && !family_has_splice(b->eng.family)) ||
                (family_has_phase(e->eng.family) && !family_has_phase(b->eng.family))) {
                c4h::set_error(std::string("model [") + model->name + "] needs sequence arrays this batch was not built with");
                return -1;
            }
            it = b->extra.emplace(key, std::move(e)).first;
        }
        return viterbi_jobs(it->second->eng, b->seqs, mode, jobs, n_jobs, results);
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_viterbi_model: ") + e.what());
        return -1;
    }
}

void c4gpu_viterbi_result_clear(c4gpu_viterbi_result *r) {
    free(r->ops);
    free(r->checkpoints);
    r->ops = nullptr; r->checkpoints = nullptr;
}

static int score_pass(Engine &eng, const ResidentSeqs &seqs, int mode, std::vector<JobOut> &outs) {
    std::vector<JobSpec> specs(seqs.n_pairs);
    for (int i = 0; i < seqs.n_pairs; i++) {
        specs[i].pair = i;
        specs[i].region = c4gpu_region{0, 0, seqs.qlen[i], seqs.tlen[i]};
    }
    return eng.run(seqs, mode, false, specs, outs);
}

int c4gpu_optimal_find_score_batch(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                   const c4gpu_pair *pairs, int32_t n_pairs, c4gpu_score *scores) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        Engine eng;
        ResidentSeqs seqs;
        std::vector<JobOut> outs;
        if (eng.init(ctx, model, params) || seqs.build(ctx, eng.family, params, pairs, n_pairs) ||
            score_pass(eng, seqs, MODE_SCORE, outs)) return -1;
        for (int i = 0; i < n_pairs; i++) scores[i] = outs[i].res.score;
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_optimal_find_score_batch: ") + e.what());
        return -1;
    }
}

int c4gpu_optimal_find_path_batch(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                  const c4gpu_pair *pairs, int32_t n_pairs, int dpmemory_mb, c4gpu_score threshold,
                                  c4gpu_alignment *alignments) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        Engine eng;
        ResidentSeqs seqs;
        if (eng.init(ctx, model, params) || seqs.build(ctx, eng.family, params, pairs, n_pairs)) return -1;
        SideLane side;
        const bool two = lanes_wanted(seqs, nullptr, false);
        if (two && side.init(ctx, model, params)) return -1;
        return find_path_lanes(eng, two ? &side : nullptr, seqs, dpmemory_mb, threshold, alignments);
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_optimal_find_path_batch: ") + e.what());
        return -1;
    }
}

int c4gpu_optimal_find_path_batch_subopt(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                         const c4gpu_pair *pairs, int32_t n_pairs, int dpmemory_mb,
                                         c4gpu_score threshold, const c4gpu_subopt *const *subopts,
                                         const uint8_t *active, c4gpu_alignment *alignments) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return -1;
        Engine eng;
        ResidentSeqs seqs;
        if (eng.init(ctx, model, params) || seqs.build(ctx, eng.family, params, pairs, n_pairs)) return -1;
        std::vector<const c4gpu_subopt *> subs(n_pairs, nullptr);
        if (subopts) for (int i = 0; i < n_pairs; i++) subs[i] = subopts[i];
        return find_path_batch(eng, seqs, dpmemory_mb, threshold, alignments, &subs, active);
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_optimal_find_path_batch_subopt: ") + e.what());
        return -1;
    }
}

c4gpu_batch *c4gpu_batch_create(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params,
                                const c4gpu_pair *pairs, int32_t n_pairs) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
        c4gpu_batch *b = new c4gpu_batch;
        b->ctx = ctx; b->model = *model; b->params = *params;
        if (b->eng.init(ctx, &b->model, &b->params) || b->seqs.build(ctx, b->eng.family, &b->params, pairs, n_pairs)) {
            delete b;
            return nullptr;
        }
        return b;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_create: ") + e.what());
        return nullptr;
    }
}

void c4gpu_batch_destroy(c4gpu_batch *b) {
    if (!b) return;
    b->clear_loop();
    for (auto &a : b->alignments) c4gpu_alignment_clear(&a);
    delete b;
    g_retired.flush();
}

c4gpu_stage *c4gpu_stage_create(c4gpu_ctx *ctx, const c4gpu_model *model, const c4gpu_params *params) {
    try {
        if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
        std::unique_ptr<c4gpu_stage> st(new c4gpu_stage);
        st->ctx.device = ctx->device; st->ctx.prop = ctx->prop;
        st->model = *model; st->params = *params;
        // lowest priority: the passes of the batch that is running get the compute units first, the staging kernels fill
        // what their tails leave idle (a load has a whole step's time)
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&st->ctx.stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
            hipEventCreate(&st->ctx.ev0) != hipSuccess || hipEventCreate(&st->ctx.ev1) != hipSuccess) {
            c4h::set_error("c4gpu_stage_create: cannot create the staging stream");
            return nullptr;
        }
        if (st->eng.init(&st->ctx, &st->model, &st->params)) return nullptr;
        if (c4_stream_sync(st->ctx.stream) != hipSuccess) return nullptr;
        return st.release();
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_stage_create: ") + e.what());
        return nullptr;
    }
}

int c4gpu_stage_load(c4gpu_stage *st, const c4gpu_pair *pairs, int32_t n_pairs) {
    try {
        if (hipSetDevice(st->ctx.device) != hipSuccess) return -1;
        const auto t0 = std::chrono::steady_clock::now();
        st->loaded = false;
        // the packed passes' splice array: written by the splice kernel itself here (ss16_kernel's formula with the calc
        // constants of the pre-splice transitions as `fold`), where the first packed launch would otherwise build it inside
        // the step (C4GPU_PK16=0: no packed pass, no array)
        const bool pk = !(c4cfg::is(c4cfg::PK16, 0)) && n_pairs >= 2 && st->eng.pk16_params_ok &&
                        st->eng.family == FAM_EST2GENOME;
        SpliceFold fold{{0, 0, 0, 0}};
        for (int i = 0; i < st->model.n_calcs; i++)
            if (st->model.calcs[i].kind == C4GPU_CALC_SPLICE_PRE) fold.add[st->model.calcs[i].param & 3] = st->model.calcs[i].value;
            else if (st->model.calcs[i].kind == C4GPU_CALC_SPLICE_POST) fold.add[st->model.calcs[i].param & 3] = 0;
        if (st->seqs.build(&st->ctx, st->eng.family, &st->params, pairs, n_pairs, true, pk ? &fold : nullptr)) return -1;
        if (pk && st->eng.ensure_ss16(st->seqs)) return -1;            // (only where the tiled splice kernel did not run)
        if (pk && c4cfg::has(c4cfg::SS16_CHECK)) {
            // test hook: the array the splice kernel wrote against the one ss16_kernel builds from the int arrays
            DevBuf<uint2> chk;
            const size_t nn = (size_t)st->seqs.ss_len;
            if (chk.alloc(nn)) return -1;
            HIP_OK(pk16_build_splice(st->eng.family, st->eng.kparams.p, st->seqs.dev.ss, st->seqs.dev.ss_stride, st->seqs.ss_len, chk.p, st->ctx.stream));
            std::vector<uint2> a(nn), b(nn);
            if (chk.download(a.data(), nn, st->ctx.stream) || st->seqs.ss16.download(b.data(), nn, st->ctx.stream)) return -1;
            HIP_OK(c4_stream_sync(st->ctx.stream));
            for (int i = 0; i < st->seqs.n_pairs; i++)
                for (long long x = st->seqs.toff[i]; x < st->seqs.toff[i] + st->seqs.tlen[i]; x++)
                    if (a[x].x != b[x].x || a[x].y != b[x].y) {
                        c4h::set_error("C4GPU_SS16_CHECK: the fused packed splice array differs from ss16_kernel's");
                        return -1;
                    }
        }
        st->loaded = true;
        st->load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_stage_load: ") + e.what());
        return -1;
    }
}

double c4gpu_stage_load_ms(const c4gpu_stage *st) { return st->load_ms; }

void c4gpu_stage_destroy(c4gpu_stage *st) { delete st; g_retired.flush(); }

int c4gpu_batch_swap_stage(c4gpu_batch *b, c4gpu_stage *st) {
    if (!st->loaded) { c4h::set_error("c4gpu_batch_swap_stage: the stage holds no loaded batch"); return -1; }
    if (memcmp(&b->model, &st->model, sizeof(c4gpu_model)) || memcmp(&b->params, &st->params, sizeof(c4gpu_params))) {
        c4h::set_error("c4gpu_batch_swap_stage: batch and stage were made for different models / parameters");
        return -1;
    }
    b->clear_loop();
    for (auto &a : b->alignments) c4gpu_alignment_clear(&a);
    b->alignments.clear(); b->scores.clear(); b->regions.clear(); b->pair_thresholds.clear();
    b->seqs.swap_with(st->seqs);
    st->loaded = false;
    return 0;
}

int c4gpu_batch_run(c4gpu_batch *b, int what, int dpmemory_mb, c4gpu_score threshold) {
    try {
        if (hipSetDevice(b->ctx->device) != hipSuccess) return -1;
        const int n = b->seqs.n_pairs;
        if (what == 0 || what == 1) {
            std::vector<JobOut> outs;
            if (score_pass(b->eng, b->seqs, what == 0 ? MODE_SCORE : MODE_REGION, outs)) return -1;
            b->scores.resize(n); b->regions.resize(n);
            for (int i = 0; i < n; i++) {
                const DevResult &r = outs[i].res;
                b->scores[i] = r.score;
                b->regions[i] = c4gpu_region{r.qs, r.ts, r.qe - r.qs, r.te - r.ts};
            }
            return 0;
        }
        for (auto &a : b->alignments) c4gpu_alignment_clear(&a);
        b->alignments.assign(n, c4gpu_alignment{});
        b->clear_loop();
        if (!b->side && lanes_wanted(b->seqs, nullptr, false)) {
            b->side.reset(new SideLane);
            if (b->side->init(b->ctx, &b->model, &b->params)) { b->side.reset(); return -1; }
        }
        if (find_path_lanes(b->eng, b->side.get(), b->seqs, dpmemory_mb, threshold, b->alignments.data(), nullptr,
                            b->pair_thresholds.empty() ? nullptr : &b->pair_thresholds)) return -1;
        b->scores.resize(n); b->regions.resize(n);
        for (int i = 0; i < n; i++) { b->scores[i] = b->alignments[i].score; b->regions[i] = b->alignments[i].region; }
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_run: ") + e.what());
        return -1;
    }
}

int c4gpu_batch_run_regions(c4gpu_batch *b, const c4gpu_region *regions, const uint8_t *active, int dpmemory_mb,
                            c4gpu_score threshold) {
    try {
        if (hipSetDevice(b->ctx->device) != hipSuccess) return -1;
        const int n = b->seqs.n_pairs;
        for (int i = 0; i < n; i++) {
            if (active && !active[i]) continue;
            const c4gpu_region &r = regions[i];
            if (r.query_start < 0 || r.target_start < 0 || r.query_length < 0 || r.target_length < 0 ||
                r.query_start + r.query_length > b->seqs.qlen[i] || r.target_start + r.target_length > b->seqs.tlen[i]) {
                c4h::set_error("c4gpu_batch_run_regions: a region lies outside its pair");
                return -1;
            }
        }
        for (auto &a : b->alignments) c4gpu_alignment_clear(&a);
        b->alignments.assign(n, c4gpu_alignment{});
        b->clear_loop();
        if (!b->side && lanes_wanted(b->seqs, active, false)) {
            b->side.reset(new SideLane);
            if (b->side->init(b->ctx, &b->model, &b->params)) { b->side.reset(); return -1; }
        }
        if (find_path_lanes(b->eng, b->side.get(), b->seqs, dpmemory_mb, threshold, b->alignments.data(), active, nullptr, regions))
            return -1;
        b->scores.resize(n); b->regions.resize(n);
        for (int i = 0; i < n; i++) { b->scores[i] = b->alignments[i].score; b->regions[i] = b->alignments[i].region; }
        return 0;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_run_regions: ") + e.what());
        return -1;
    }
}

int c4gpu_batch_set_thresholds(c4gpu_batch *b, const c4gpu_score *per_pair) {
    if (!per_pair) { b->pair_thresholds.clear(); return 0; }
    b->pair_thresholds.assign(per_pair, per_pair + b->seqs.n_pairs);
    return 0;
}

// GAM_Result_exhaustive_create's do/while (gam.c:1158-1172) for the whole batch: block what the previous
// round found (GAM_Result_add_alignment -> SubOpt_add_alignment, gam.c:673), then the next best paths.
int c4gpu_batch_next_paths(c4gpu_batch *b, int dpmemory_mb, c4gpu_score threshold) {
    try {
        if (hipSetDevice(b->ctx->device) != hipSuccess) return -1;
        const int n = b->seqs.n_pairs;
        if ((int)b->alignments.size() != n) { c4h::set_error("c4gpu_batch_next_paths needs a c4gpu_batch_run(b, 2, ...) first"); return -1; }
        if (b->subopts.empty()) {
            b->subopts.resize(n);
            for (int i = 0; i < n; i++) b->subopts[i] = c4gpu_subopt_create(b->seqs.qlen[i], b->seqs.tlen[i]);
            b->in_loop.assign(n, 1);
        }
        std::vector<const c4gpu_subopt *> subs(n);
        int still = 0;
        for (int i = 0; i < n; i++) {
            if (b->in_loop[i] && b->alignments[i].valid) {
                if (c4gpu_subopt_add_alignment(b->subopts[i], &b->model, &b->alignments[i])) return -1;
                still++;
            } else {
                b->in_loop[i] = 0;
            }
            subs[i] = b->subopts[i];
        }
        for (auto &a : b->alignments) c4gpu_alignment_clear(&a);
        if (!still) return 0;
        if (find_path_batch(b->eng, b->seqs, dpmemory_mb, threshold, b->alignments.data(), &subs, b->in_loop.data(),
                            b->pair_thresholds.empty() ? nullptr : &b->pair_thresholds)) return -1;
        int found = 0;
        for (int i = 0; i < n; i++) {
            found += b->alignments[i].valid ? 1 : 0;
            b->scores[i] = b->alignments[i].score; b->regions[i] = b->alignments[i].region;
        }
        return found;
    } catch (const std::exception &e) {
        c4h::set_error(std::string("c4gpu_batch_next_paths: ") + e.what());
        return -1;
    }
}

int c4gpu_batch_scores(c4gpu_batch *b, c4gpu_score *scores, c4gpu_region *regions) {
    for (size_t i = 0; i < b->scores.size(); i++) {
        if (scores) scores[i] = b->scores[i];
        if (regions) regions[i] = b->regions[i];
    }
    return (int)b->scores.size();
}

int c4gpu_batch_alignment(c4gpu_batch *b, int32_t i, c4gpu_alignment *out) {
    if (i < 0 || i >= (int)b->alignments.size()) return -1;
    const c4gpu_alignment &a = b->alignments[i];
    *out = a;
    out->op_transition = out->op_length = nullptr;
    if (a.n_ops) {
        out->op_transition = (int32_t *)malloc(sizeof(int32_t) * a.n_ops);
        out->op_length = (int32_t *)malloc(sizeof(int32_t) * a.n_ops);
        memcpy(out->op_transition, a.op_transition, sizeof(int32_t) * a.n_ops);
        memcpy(out->op_length, a.op_length, sizeof(int32_t) * a.n_ops);
    }
    return 0;
}

// every alignment of the batch in one int32 stream: first a row of 7 ints per pair (valid, score, region (4), n_ops), then
// the (transition, length) pairs of all valid alignments in pair order; returns the ints needed (written only when they
// fit `cap`): what a rank ships to the rank that prints
int64_t c4gpu_batch_export(c4gpu_batch *b, int32_t *out, int64_t cap) {
    int64_t need = 7 * (int64_t)b->alignments.size();
    for (const c4gpu_alignment &a : b->alignments) need += a.valid ? 2 * (int64_t)a.n_ops : 0;
    if (!out || need > cap) return need;
    int64_t pos = 0, ops = 7 * (int64_t)b->alignments.size();
    for (const c4gpu_alignment &a : b->alignments) {
        out[pos++] = a.valid; out[pos++] = a.valid ? a.score : 0;
        out[pos++] = a.region.query_start; out[pos++] = a.region.target_start;
        out[pos++] = a.region.query_length; out[pos++] = a.region.target_length;
        out[pos++] = a.valid ? a.n_ops : 0;
        if (a.valid) for (int k = 0; k < a.n_ops; k++) { out[ops++] = a.op_transition[k]; out[ops++] = a.op_length[k]; }
    }
    return need;
}

int c4gpu_batch_kernel_stats(c4gpu_batch *b, int mode, int reset, double *ms, int64_t *launches, int64_t *cells) {
    c4gpu_ctx *ctx = b->ctx;
    if (mode < 0 || mode > 3) return -1;
    c4gpu_ctx *lanes[2] = {ctx, b->side ? &b->side->ctx : nullptr};        // a large batch runs on two lanes (find_path_lanes)
    if (ms) *ms = 0;
    if (launches) *launches = 0;
    if (cells) *cells = 0;
    for (c4gpu_ctx *c : lanes) {
        if (!c) continue;
        if (ms) *ms += c->kernel_ms[mode];
        if (launches) *launches += c->kernel_launches[mode];
        if (cells) *cells += c->kernel_cells[mode];
        if (reset) { c->kernel_ms[mode] = 0; c->kernel_launches[mode] = 0; c->kernel_cells[mode] = 0; }
        c->timing = true;
    }
    return 0;
}

}  // extern "C"

// ---- SDP on the device (seeded flavour): its own file, same translation unit ------------------------------------
#include "c4_sdp_dev.inc"
#include "c4_seed_dev.inc"
