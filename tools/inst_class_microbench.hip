// inst_class_microbench.hip — the issue cost of every instruction class of the packed score pass's column loop
// (c4_viterbi16_kernel.h) at the occupancies the register file allows ("microbenchmark each
// instruction class of its inner loop at 3 waves/SIMD and write the per-step cycle budget").
//
// Each wave runs ITER x 4 blocks of 8 instructions (8 independent chains, or one dependent chain: NCH = 1) of ONE
// class, written as one asm statement per block, between two s_memtime reads.  Workgroups of 64 x W threads, one (or
// two) per CU, kept apart by their LDS allocation, so W waves share a CU = W / 4 per SIMD.  Reported per
// (class, waves per SIMD):
//   cycles_per_inst_per_wave    shader cycles one wave needs per instruction
//   wave_inst_per_clk_per_simd  = waves per SIMD / that  (what the SIMD issues per cycle)
//   cycles_per_inst_per_simd    = its inverse: the price of one such instruction in SIMD cycles at that occupancy
//   effective_clock_ghz         slowest wave's cycles / wall time of the launch (HIP events)
// LDS classes: 8 reads in flight, then s_waitcnt lgkmcnt(0); address patterns: lane-linear (no bank conflict), one
// address for all lanes (broadcast), pseudo-random inside a 24 x 24 int table (the substitution matrix's pattern).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/inst_class_microbench.hip -o build/inst_class_microbench
// Run on the MI355X: build/inst_class_microbench > gpurun_out/inst_class.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;              // blocks of 8 instructions per wave

enum Op { ADD_U32, MAX_I32, PK_ADD_I16C, PK_SUB_I16C, PK_MAX_I16, PK_MAX_U16, PK_ADD_U16, PK_ASHR, PK_LT_MASK, PERM, BFI, AND_OR,
          BITOP3, AND_B32, OR_B32, DPP_WAVE_SHR, DPP_ROW_SHR, PK_TRANS, BIAS_TRANS, PK_SELECT, I32_TRANS,
          LDS_B32_LINEAR, LDS_B32_BCAST, LDS_B32_TABLE, LDS_B64_TABLE, LDS_U16_D16_TABLE, N_OPS };
static const char *op_name[N_OPS] = {
    "v_add_u32", "v_max_i32", "v_pk_add_i16 clamp", "v_pk_sub_i16 clamp", "v_pk_max_i16", "v_pk_max_u16", "v_pk_add_u16",
    "v_pk_ashrrev_i16", "pair:v_pk_sub_i16 clamp+v_pk_ashrrev_i16 (a<b mask)", "v_perm_b32", "v_bfi_b32", "v_and_or_b32",
    "v_bitop3_b32", "v_and_b32", "v_or_b32", "v_mov_b32_dpp wave_shr:1", "v_mov_b32_dpp row_shr:1",
    "pair:v_pk_add_i16 clamp+v_pk_max_i16 (one packed transition)", "pair:v_add_u32+v_pk_max_u16 (biased packed transition)",
    "quad:v_pk_sub_i16+v_pk_ashrrev_i16+v_bfi_b32+v_pk_max_i16 (transition with a payload)",
    "pair:v_add_u32+v_max_i32 (one 32-bit transition)",
    "ds_read_b32 lane-linear", "ds_read_b32 one address", "ds_read_b32 24x24 table", "ds_read_b64 24x24 table",
    "ds_read_u16_d16(+_hi) 24x24 table"};
static const int op_insts[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 4, 2, 1, 1, 1, 1, 1};

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
#define I_PK_ADD_I16C(A, B) "v_pk_add_i16 " A ", " A ", " K " clamp\n\t"
#define I_PK_SUB_I16C(A, B) "v_pk_sub_i16 " A ", " A ", " K " clamp\n\t"
#define I_PK_MAX_I16(A, B) "v_pk_max_i16 " A ", " A ", " K "\n\t"
#define I_PK_MAX_U16(A, B) "v_pk_max_u16 " A ", " A ", " K "\n\t"
#define I_PK_ADD_U16(A, B) "v_pk_add_u16 " A ", " A ", " K "\n\t"
#define I_PK_ASHR(A, B) "v_pk_ashrrev_i16 " A ", 15, " A " op_sel_hi:[0,1]\n\t"
#define I_PK_LT_MASK(A, B) "v_pk_sub_i16 " B ", " A ", " K " clamp\n\tv_pk_ashrrev_i16 " A ", 15, " B " op_sel_hi:[0,1]\n\t"
#define I_PERM(A, B) "v_perm_b32 " A ", " A ", " B ", " K "\n\t"
#define I_BFI(A, B) "v_bfi_b32 " A ", " K ", " B ", " A "\n\t"
#define I_AND_OR(A, B) "v_and_or_b32 " A ", " A ", " K ", " B "\n\t"
#define I_BITOP3(A, B) "v_bitop3_b32 " A ", " A ", " K ", " B " bitop3:0xca\n\t"
#define I_AND_B32(A, B) "v_and_b32 " A ", " A ", " K "\n\t"
#define I_OR_B32(A, B) "v_or_b32 " A ", " A ", " K "\n\t"
#define I_DPP_WAVE_SHR(A, B) "v_mov_b32_dpp " A ", " B " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPP_ROW_SHR(A, B) "v_mov_b32_dpp " A ", " B " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_PK_TRANS(A, B) "v_pk_add_i16 " B ", " A ", " K " clamp\n\tv_pk_max_i16 " A ", " A ", " B "\n\t"
#define I_BIAS_TRANS(A, B) "v_add_u32 " B ", " A ", " K "\n\tv_pk_max_u16 " A ", " A ", " B "\n\t"
#define I_PK_SELECT(A, B) "v_pk_sub_i16 " B ", " A ", " K " clamp\n\tv_pk_ashrrev_i16 " B ", 15, " B " op_sel_hi:[0,1]\n\t" \
                          "v_bfi_b32 " B ", " B ", " K ", " A "\n\tv_pk_max_i16 " A ", " A ", " B "\n\t"
#define I_I32_TRANS(A, B) "v_add_u32 " B ", " A ", " K "\n\tv_max_i32 " A ", " A ", " B "\n\t"
// LDS: A = destination, B = byte address
#define I_LDS_B32(A, B) "ds_read_b32 " A ", " B "\n\t"
#define I_LDS_U16(A, B) "ds_read_u16_d16 " A ", " B "\n\t"
#define I_LDS_U16HI(A, B) "ds_read_u16_d16_hi " A ", " B "\n\t"
#define BLOCK8(I) I(A0, B0) I(A1, B1) I(A2, B2) I(A3, B3) I(A4, B4) I(A5, B5) I(A6, B6) I(A7, B7)
#define CHAIN8(I) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0) I(A0, B0)
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(k) : "vcc", "memory"
#define CASE(NAME) \
    if constexpr (OP == NAME) { if constexpr (NCH == 8) asm volatile(BLOCK8(I_##NAME) OPERANDS); else asm volatile(CHAIN8(I_##NAME) OPERANDS); }
#define LDSCASE(NAME, I) \
    if constexpr (OP == NAME) asm volatile(BLOCK8(I) "s_waitcnt lgkmcnt(0)\n\t" OPERANDS);

template <int OP, int NCH>
__device__ __forceinline__ void block(int (&a)[8], int (&b)[8], int k) {
    CASE(ADD_U32) CASE(MAX_I32) CASE(PK_ADD_I16C) CASE(PK_SUB_I16C) CASE(PK_MAX_I16) CASE(PK_MAX_U16) CASE(PK_ADD_U16) CASE(PK_ASHR)
    CASE(PK_LT_MASK) CASE(PERM) CASE(BFI) CASE(AND_OR) CASE(BITOP3) CASE(AND_B32) CASE(OR_B32) CASE(DPP_WAVE_SHR) CASE(DPP_ROW_SHR)
    CASE(PK_TRANS) CASE(BIAS_TRANS) CASE(PK_SELECT) CASE(I32_TRANS)
    LDSCASE(LDS_B32_LINEAR, I_LDS_B32) LDSCASE(LDS_B32_BCAST, I_LDS_B32) LDSCASE(LDS_B32_TABLE, I_LDS_B32)
    if constexpr (OP == LDS_B64_TABLE) {
        // four 8-byte reads (a pair of chains each)
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : "=&v"(*(long long *)&a[0]), "=&v"(*(long long *)&a[2]) : "v"(b[0]), "v"(b[1]) : "memory");
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\t"
                     : "=&v"(*(long long *)&a[4]), "=&v"(*(long long *)&a[6]) : "v"(b[2]), "v"(b[3]) : "memory");
    }
    if constexpr (OP == LDS_U16_D16_TABLE)
        asm volatile(I_LDS_U16(A0, B0) I_LDS_U16HI(A0, B1) I_LDS_U16(A1, B2) I_LDS_U16HI(A1, B3) I_LDS_U16(A2, B4) I_LDS_U16HI(A2, B5)
                     I_LDS_U16(A3, B6) I_LDS_U16HI(A3, B7) "s_waitcnt lgkmcnt(0)\n\t" OPERANDS);
}

template <int OP, int NCH>
__global__ __launch_bounds__(1024) void bench_kernel(long long *cycles, int *sink, int k) {
    extern __shared__ int lds[];
    int a[8], b[8];
    constexpr bool IS_LDS = OP >= LDS_B32_LINEAR;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        a[c] = threadIdx.x + c;
        if constexpr (OP == LDS_B32_LINEAR) b[c] = (int)((threadIdx.x & 63) * 4 + c * 256 + (threadIdx.x >> 6) * 2048);
        else if constexpr (OP == LDS_B32_BCAST) b[c] = c * 256 + (threadIdx.x >> 6) * 2048;
        else if constexpr (IS_LDS) {
            // 24 x query code (0..4) + target code (0..4), as the packed pass reads its substitution scores
            unsigned h = (threadIdx.x * 2654435761u + c * 40503u) >> 7;
            b[c] = (int)(((h % 5) * 24 + ((h / 5) % 5)) * (OP == LDS_B64_TABLE ? 8 : OP == LDS_U16_D16_TABLE ? 2 : 4));
        } else b[c] = k + c;
    }
    for (int x = threadIdx.x; x < 16 * 1024; x += blockDim.x) lds[x] = x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER / 4; it++) {
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

struct Row { std::string op; int nch; double waves_per_simd, cyc_per_inst_wave, inst_per_clk_simd, eff_clock_ghz; };

template <int OP, int NCH>
static Row run(int waves_per_cu, int n_cu, long long *d_cycles, int *d_sink) {
    const int blocks_per_cu = waves_per_cu > 16 ? 2 : 1;
    const int threads = 64 * waves_per_cu / blocks_per_cu;
    const int blocks = n_cu * blocks_per_cu;
    const size_t lds = (blocks_per_cu == 2 ? 70 : 96) * 1024;
    CHECK(hipFuncSetAttribute((const void *)bench_kernel<OP, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench_kernel<OP, NCH>), dim3(blocks), dim3(threads), lds, 0, d_cycles, d_sink, 3);
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
    const double insts = (double)ITER * 8 * op_insts[OP];
    Row r;
    r.op = op_name[OP]; r.nch = NCH; r.waves_per_simd = waves_per_cu / 4.0;
    r.cyc_per_inst_wave = mean / insts;
    r.inst_per_clk_simd = (waves_per_cu / 4.0) / r.cyc_per_inst_wave;
    r.eff_clock_ghz = mx / (ms * 1e-3) / 1e9;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

template <int OP>
static void sweep(std::vector<Row> &rows, int n_cu, long long *d_cycles, int *d_sink) {
    rows.push_back(run<OP, 1>(4, n_cu, d_cycles, d_sink));           // one dependent chain, 1 wave per SIMD: latency
    for (int w : {4, 8, 12, 16, 24, 32})                             // 1, 2, 3, 4, 6, 8 waves per SIMD, 8 independent chains
        rows.push_back(run<OP, 8>(w, n_cu, d_cycles, d_sink));
}

template <int... OPS>
static void sweep_all(std::vector<Row> &rows, int n_cu, long long *d_cycles, int *d_sink) { (sweep<OPS>(rows, n_cu, d_cycles, d_sink), ...); }

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    long long *d_cycles; int *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(long long) * n_cu * 32));
    CHECK(hipMalloc(&d_sink, 64));
    std::vector<Row> rows;
    sweep_all<ADD_U32, MAX_I32, PK_ADD_I16C, PK_SUB_I16C, PK_MAX_I16, PK_MAX_U16, PK_ADD_U16, PK_ASHR, PK_LT_MASK, PERM, BFI, AND_OR, BITOP3,
              AND_B32, OR_B32, DPP_WAVE_SHR, DPP_ROW_SHR, PK_TRANS, BIAS_TRANS, PK_SELECT, I32_TRANS,
              LDS_B32_LINEAR, LDS_B32_BCAST, LDS_B32_TABLE, LDS_B64_TABLE, LDS_U16_D16_TABLE>(rows, n_cu, d_cycles, d_sink);
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"clock_khz\": %d, \"iter\": %d,\n \"rows\": [\n",
           prop.name, prop.gcnArchName, n_cu, prop.clockRate, ITER);
    for (size_t i = 0; i < rows.size(); i++) {
        const Row &r = rows[i];
        printf("  {\"op\": \"%s\", \"independent_chains\": %d, \"waves_per_simd\": %.0f, \"cycles_per_inst_per_wave\": %.3f, "
               "\"wave_inst_per_clk_per_simd\": %.4f, \"cycles_per_inst_per_simd\": %.3f, \"effective_clock_ghz\": %.3f}%s\n",
               r.op.c_str(), r.nch, r.waves_per_simd, r.cyc_per_inst_wave, r.inst_per_clk_simd, 1.0 / r.inst_per_clk_simd, r.eff_clock_ghz,
               i + 1 < rows.size() ? "," : "");
    }
    printf(" ]}\n");
    return 0;
}
