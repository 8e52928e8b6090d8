// valu_issue_microbench.hip — what is the VALU issue roofline of gfx950 for the integer ops the Viterbi
// kernel is made of?  ("settle the 2-vs-4-cycle question".)
//
// Each wave runs ITER iterations of a block of NCH independent chains of ONE instruction (inline asm, so
// the compiler cannot merge, reorder into other opcodes or drop them); the wave reads s_memtime before and
// after.  Workgroups of 64*W threads, one per CU (forced with a 96 KB LDS allocation), so W waves share a
// CU = W/4 per SIMD.  Reported per (op, waves per SIMD):
//   cyc_per_inst_wave   = shader cycles one wave needs per instruction (s_memtime ticks: shader clock)
//   inst_per_clk_simd   = waves_per_simd / cyc_per_inst_wave  (wave-instructions per cycle per SIMD)
// plus the dependent-chain latency of each op (NCH = 1).  Wall-clock is measured too (HIP events) and
// gives the effective shader clock under this load.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_microbench.hip -o build/valu_issue_microbench
// Run on the MI355X: build/valu_issue_microbench > gpurun_out/valu_issue.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 8192;              // blocks of 8 instructions per wave

enum Op { ADD_U32, MAX_I32, CNDMASK, CMP_LT, MOV, MOV_DPP, ADD3, MAX3, PK_ADD_I16, PK_MAX_I16, AND_B32, LSHL_ADD,
          ADD_CO, MED3, MIN_I32, SUB_U32, CMP_CNDMASK_PAIR, ADD_MAX_PAIR, DP_MIX, N_OPS };
static const char *op_name[N_OPS] = {"v_add_u32", "v_max_i32", "v_cndmask_b32", "v_cmp_lt_i32(vcc)", "v_mov_b32",
                                     "v_mov_b32_dpp(wave_shr:1)", "v_add3_u32", "v_max3_i32", "v_pk_add_i16", "v_pk_max_i16",
                                     "v_and_b32", "v_lshl_add_u32", "v_add_co_u32", "v_med3_i32", "v_min_i32", "v_sub_u32",
                                     "pair:v_cmp_lt_i32+v_cndmask_b32", "pair:v_add_u32+v_max_i32",
                                     "mix:add,cmp,cndmask,cndmask (one DP transition with one payload)"};
static const int op_insts[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 4};

// One asm statement per unrolled block (the assembler text of 8 instructions): hipcc pads the boundary of
// every inline-asm statement that touches vcc with an s_nop, which would cost issue slots of its own.
#define A0 "%0"
#define A1 "%1"
#define A2 "%2"
#define A3 "%3"
#define A4 "%4"
#define A5 "%5"
#define A6 "%6"
#define A7 "%7"
#define B0 "%8"
#define B1 "%9"
#define B2 "%10"
#define B3 "%11"
#define B4 "%12"
#define B5 "%13"
#define B6 "%14"
#define B7 "%15"
#define K "%16"
#define I_ADD_U32(A, B) "v_add_u32 " A ", " A ", " K "\n\t"
#define I_MAX_I32(A, B) "v_max_i32 " A ", " A ", " K "\n\t"
#define I_MIN_I32(A, B) "v_min_i32 " A ", " A ", " K "\n\t"
#define I_SUB_U32(A, B) "v_sub_u32 " A ", " A ", " K "\n\t"
#define I_CNDMASK(A, B) "v_cndmask_b32 " A ", " A ", " K ", vcc\n\t"
#define I_CMP_LT(A, B) "v_cmp_lt_i32 vcc, " A ", " K "\n\t"
#define I_MOV(A, B) "v_mov_b32 " A ", " K "\n\t"
#define I_MOV_DPP(A, B) "v_mov_b32_dpp " A ", " B " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD3(A, B) "v_add3_u32 " A ", " A ", " K ", " K "\n\t"
#define I_MAX3(A, B) "v_max3_i32 " A ", " A ", " K ", " B "\n\t"
#define I_MED3(A, B) "v_med3_i32 " A ", " A ", " K ", " B "\n\t"
#define I_PK_ADD_I16(A, B) "v_pk_add_i16 " A ", " A ", " K "\n\t"
#define I_PK_MAX_I16(A, B) "v_pk_max_i16 " A ", " A ", " K "\n\t"
#define I_AND_B32(A, B) "v_and_b32 " A ", " A ", " K "\n\t"
#define I_LSHL_ADD(A, B) "v_lshl_add_u32 " A ", " A ", 1, " K "\n\t"
#define I_ADD_CO(A, B) "v_add_co_u32 " A ", vcc, " A ", " K "\n\t"
#define I_CMP_CNDMASK_PAIR(A, B) "v_cmp_lt_i32 vcc, " A ", " K "\n\tv_cndmask_b32 " A ", " A ", " K ", vcc\n\t"
#define I_ADD_MAX_PAIR(A, B) "v_add_u32 " B ", " A ", " K "\n\tv_max_i32 " A ", " A ", " B "\n\t"
// t = src + calc ; win = old < t ; score = win ? t : old ; payload = win ? p : payload
#define I_DP_MIX(A, B) "v_add_u32 " B ", " A ", " K "\n\tv_cmp_lt_i32 vcc, " A ", " B "\n\tv_cndmask_b32 " A ", " A ", " B ", vcc\n\t" \
                       "v_cndmask_b32 " B ", " B ", " K ", vcc\n\t"
#define BLOCK8(I) I(A0, B0) I(A1, B1) I(A2, B2) I(A3, B3) I(A4, B4) I(A5, B5) I(A6, B6) I(A7, B7)
#define CHAIN8(I) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0)
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(k) : "vcc"
#define CASE(NAME) \
    if constexpr (OP == NAME) { if constexpr (NCH == 8) asm volatile(BLOCK8(I_##NAME) OPERANDS); else asm volatile(CHAIN8(I_##NAME) OPERANDS); }

// 8 instructions (x the op's own count): 8 independent chains (NCH == 8) or one dependent chain (NCH == 1)
template <int OP, int NCH>
__device__ __forceinline__ void block(int (&a)[8], int (&b)[8], int k) {
    CASE(ADD_U32) CASE(MAX_I32) CASE(MIN_I32) CASE(SUB_U32) CASE(CNDMASK) CASE(CMP_LT) CASE(MOV) CASE(MOV_DPP) CASE(ADD3)
    CASE(MAX3) CASE(MED3) CASE(PK_ADD_I16) CASE(PK_MAX_I16) CASE(AND_B32) CASE(LSHL_ADD) CASE(ADD_CO)
    CASE(CMP_CNDMASK_PAIR) CASE(ADD_MAX_PAIR) CASE(DP_MIX)
}

template <int OP, int NCH>
__global__ __launch_bounds__(1024) void bench_kernel(long long *cycles, int *sink, int k) {
    extern __shared__ int lds[];
    int a[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { a[c] = threadIdx.x + c; b[c] = k + c; }
    if (threadIdx.x == 0) lds[0] = k;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER / 4; it++) {      // 4 blocks of 8 per trip: loop overhead is 2 SALU ops per 32+ VALU
        block<OP, NCH>(a, b, k);
        block<OP, NCH>(a, b, k);
        block<OP, NCH>(a, b, k);
        block<OP, NCH>(a, b, k);
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += a[c] + b[c];
    if (s == 0x7fffffff) sink[0] = s + lds[0];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Row { std::string op; int nch, waves_per_simd; double cyc_per_inst_wave, inst_per_clk_simd, eff_clock_ghz; };

template <int OP, int NCH>
static Row run(int waves_per_cu, int n_cu, long long *d_cycles, int *d_sink) {
    // one workgroup of up to 16 waves per CU (96 KB of LDS keeps a second one out); 8 waves per SIMD = two
    // workgroups of 1024 threads per CU (70 KB each: two fit in 160 KB, three do not)
    const int blocks_per_cu = waves_per_cu > 16 ? 2 : 1;
    const int threads = 64 * waves_per_cu / blocks_per_cu;
    const int blocks = n_cu * blocks_per_cu;
    const size_t lds = (blocks_per_cu == 2 ? 70 : 96) * 1024;
    CHECK(hipFuncSetAttribute((const void *)bench_kernel<OP, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench_kernel<OP, NCH>), dim3(blocks), dim3(threads), lds, 0, d_cycles, d_sink, 3);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((bench_kernel<OP, NCH>), dim3(blocks), dim3(threads), lds, 0, d_cycles, d_sink, 3);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> cyc((size_t)n_cu * waves_per_cu);
    CHECK(hipMemcpy(cyc.data(), d_cycles, cyc.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (long long c : cyc) { sum += (double)c; mx = mx > (double)c ? mx : (double)c; }
    const double mean = sum / cyc.size();
    const double insts = (double)ITER * 8 * op_insts[OP];      // ITER blocks of 8
    Row r;
    r.op = op_name[OP]; r.nch = NCH; r.waves_per_simd = waves_per_cu / 4;
    r.cyc_per_inst_wave = mean / insts;
    r.inst_per_clk_simd = (waves_per_cu / 4.0) / r.cyc_per_inst_wave;
    r.eff_clock_ghz = mx / (ms * 1e-3) / 1e9;   // the slowest wave spans ~the whole kernel
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

template <int OP>
static void sweep(std::vector<Row> &rows, int n_cu, long long *d_cycles, int *d_sink) {
    rows.push_back(run<OP, 1>(4, n_cu, d_cycles, d_sink));           // dependent chain, 1 wave per SIMD: latency
    for (int w : {4, 8, 16, 32})                                     // 1, 2, 4, 8 waves per SIMD, 8 independent chains
        rows.push_back(run<OP, 8>(w, n_cu, d_cycles, d_sink));
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    long long *d_cycles; int *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(long long) * n_cu * 32));
    CHECK(hipMalloc(&d_sink, 64));
    std::vector<Row> rows;
    sweep<ADD_U32>(rows, n_cu, d_cycles, d_sink);
    sweep<MAX_I32>(rows, n_cu, d_cycles, d_sink);
    sweep<MIN_I32>(rows, n_cu, d_cycles, d_sink);
    sweep<SUB_U32>(rows, n_cu, d_cycles, d_sink);
    sweep<CNDMASK>(rows, n_cu, d_cycles, d_sink);
    sweep<CMP_LT>(rows, n_cu, d_cycles, d_sink);
    sweep<MOV>(rows, n_cu, d_cycles, d_sink);
    sweep<MOV_DPP>(rows, n_cu, d_cycles, d_sink);
    sweep<AND_B32>(rows, n_cu, d_cycles, d_sink);
    sweep<ADD3>(rows, n_cu, d_cycles, d_sink);
    sweep<MAX3>(rows, n_cu, d_cycles, d_sink);
    sweep<MED3>(rows, n_cu, d_cycles, d_sink);
    sweep<LSHL_ADD>(rows, n_cu, d_cycles, d_sink);
    sweep<ADD_CO>(rows, n_cu, d_cycles, d_sink);
    sweep<PK_ADD_I16>(rows, n_cu, d_cycles, d_sink);
    sweep<PK_MAX_I16>(rows, n_cu, d_cycles, d_sink);
    sweep<CMP_CNDMASK_PAIR>(rows, n_cu, d_cycles, d_sink);
    sweep<ADD_MAX_PAIR>(rows, n_cu, d_cycles, d_sink);
    sweep<DP_MIX>(rows, n_cu, d_cycles, d_sink);
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"clock_khz\": %d, \"iter\": %d,\n \"rows\": [\n",
           prop.name, prop.gcnArchName, n_cu, prop.clockRate, ITER);
    for (size_t i = 0; i < rows.size(); i++) {
        const Row &r = rows[i];
        printf("  {\"op\": \"%s\", \"independent_chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_per_wave\": %.3f, "
               "\"wave_inst_per_clk_per_simd\": %.4f, \"effective_clock_ghz\": %.3f}%s\n",
               r.op.c_str(), r.nch, r.waves_per_simd, r.cyc_per_inst_wave, r.inst_per_clk_simd, r.eff_clock_ghz,
               i + 1 < rows.size() ? "," : "");
    }
    printf(" ]}\n");
    return 0;
}
