// VALU issue-rate calibration for gfx950: wave-instructions per cycle per SIMD for the integer / packed-16 /
// DPP / permute instructions the mask-stage kernels are made of, as independent and dependent chains at
// 1..8 waves per SIMD.  Prints one JSON document; tools/microbench/run.sh stores it under profiles/.
//
//   build: hipcc -O2 --offload-arch=gfx950 valu_issue.hip -o valu_issue
//
// Method: a 256-thread workgroup = one wave per SIMD; `W` workgroups per CU are forced by a dynamic-LDS request of
// floor(160 KiB / W) per workgroup, grid = CUs x W (all resident at once).  Each wave executes ITERS x 64 copies of
// one instruction (inline asm, so nothing is folded), bracketed by s_memtime; the slowest wave's cycle count and the
// hipEvent wall time give   rate = W x ITERS x 64 / cycles   [wave-insts / cycle / SIMD].
// A SIMD that needs 4 cycles per wave64 instruction (16 lanes/clk) saturates at 0.25, a 2-cycle SIMD at 0.5.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// eight independent accumulators %0..%7, two read-only operands %8, %9
#define REP8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define DEP8(INS) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0)
#define X8(B) B B B B B B B B

#define I_PKMIN3(i)  "v_pk_minimum3_f16 %" #i ", %" #i ", %8, %9\n\t"
#define I_PKMAX3(i)  "v_pk_maximum3_f16 %" #i ", %" #i ", %8, %9\n\t"
#define I_PKMAD(i)   "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n\t"
#define I_PKADD(i)   "v_pk_add_u16 %" #i ", %" #i ", %8\n\t"
#define I_PKSUBI(i)  "v_pk_sub_i16 %" #i ", %" #i ", %8\n\t"
#define I_PKMAXU(i)  "v_pk_max_u16 %" #i ", %" #i ", %8\n\t"
#define I_PKMINU(i)  "v_pk_min_u16 %" #i ", %" #i ", %8\n\t"
#define I_PKLSHR(i)  "v_pk_lshrrev_b16 %" #i ", 1, %" #i "\n\t"
#define I_PERM(i)    "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_DPPROW(i)  "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPWAVE(i) "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADDDPP(i)  "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD(i)     "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_ADD3(i)    "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define I_ANDOR(i)   "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_LSHLOR(i)  "v_lshl_or_b32 %" #i ", %" #i ", 1, %9\n\t"
#define I_SAD(i)     "v_sad_u8 %" #i ", %" #i ", %8, %9\n\t"
#define I_MAD24(i)   "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n\t"
#define I_MULLO(i)   "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define I_ALIGN(i)   "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n\t"
#define I_BFE(i)     "v_bfe_u32 %" #i ", %" #i ", 1, 30\n\t"
#define I_MIN3U(i)   "v_min3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define I_FMA(i)     "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_PKFMA(i)   "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n\t"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define I_CMP16(i)   "v_cmp_gt_i16 vcc, %" #i ", %8\n\t"
#define I_CMP32S(i)  "v_cmp_lt_i32_e64 s[20:21], %" #i ", %8\n\t"
#define I_WRLANE(i)  "v_writelane_b32 %" #i ", s20, 5\n\t"
#define I_LSHL64(i)  "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define I_XOR(i)     "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define I_BFI(i)     "v_bfi_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_MOV(i)     "v_mov_b32 %" #i ", %8\n\t"
#define I_MBCNT(i)   "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n\t"
#define I_AND(i)     "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define I_OR(i)      "v_or_b32 %" #i ", %" #i ", %8\n\t"
#define I_SUB(i)     "v_sub_u32 %" #i ", %" #i ", %8\n\t"
#define I_MINU(i)    "v_min_u32 %" #i ", %" #i ", %8\n\t"
#define I_MAXU(i)    "v_max_u32 %" #i ", %" #i ", %8\n\t"
#define I_MAXI16(i)  "v_max_i16 %" #i ", %" #i ", %8\n\t"
#define I_ADD16(i)   "v_add_u16 %" #i ", %" #i ", %8\n\t"
#define I_MUL24(i)   "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define I_LSHR(i)    "v_lshrrev_b32 %" #i ", 1, %" #i "\n\t"
#define I_ASHR(i)    "v_ashrrev_i32 %" #i ", 1, %" #i "\n\t"
#define I_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define I_ADDCO(i)   "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n\t"
#define I_ADDF(i)    "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define I_PKADDF16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n\t"
#define I_PKMULLO(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n\t"
#define I_MADU16(i)  "v_mad_u16 %" #i ", %" #i ", %8, %9\n\t"
#define I_BFREV(i)   "v_bfrev_b32 %" #i ", %" #i "\n\t"
#define I_CNDADD(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_add_u32 %" #i ", %" #i ", %8\n\tv_add_u32 %" #i ", %" #i ", %9\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_CNDS(i)    "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define I_PERMADD(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n\t"

// mixed sequences (registers: %0..%7 accumulators, %8 %9 operands); names end in _xN = N instructions per macro
#define M_PERM_ADD_IND(i)  "v_perm_b32 %0, %0, %8, %9\n\tv_add_u32 %1, %1, %8\n\t"
#define M_ADD_ADD_PERM_PERM(i) "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_perm_b32 %2, %2, %8, %9\n\tv_perm_b32 %3, %3, %8, %9\n\t"
#define M_ADD_PERM_ADD_PERM(i) "v_add_u32 %0, %0, %8\n\tv_perm_b32 %2, %2, %8, %9\n\tv_add_u32 %1, %1, %8\n\tv_perm_b32 %3, %3, %8, %9\n\t"
#define M_MUL_SUB_SUB_OR(i) "v_mul_u32_u24 %0, %1, %8\n\tv_sub_u32 %2, %3, %0\n\tv_sub_u32 %4, %5, %0\n\tv_or_b32 %6, %2, %4\n\t"
#define M_ADD4_IND(i) "v_add_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_or_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %9\n\t"
#define M_ADD_LDS(i) "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tds_read_u16 %7, %9\n\t"
#define M_ADD_SALU(i) "v_add_u32 %0, %0, %8\n\ts_mov_b64 s[20:21], -1\n\tv_add_u32 %1, %1, %8\n\ts_mov_b64 s[20:21], 1\n\t"
#define M_VMOV_SGPR(i) "v_mov_b32 %0, s20\n\tv_mov_b32 %1, s21\n\tv_mov_b32 %2, s20\n\tv_mov_b32 %3, s21\n\t"
#define M_STEP_V(i) "v_perm_b32 %0, %0, %8, %9\n\tv_perm_b32 %1, %1, %8, %9\n\tv_perm_b32 %2, %2, %8, %9\n\tv_mul_u32_u24 %3, %4, %8\n\tv_sub_u32 %5, %6, %3\n\tv_sub_u32 %7, %4, %3\n\tv_or_b32 %5, %5, %7\n\tv_add_u32 %6, %6, %0\n\tv_sub_u32 %6, %6, %1\n\tv_add_u32 %4, %4, %2\n\tv_sub_u32 %4, %4, %1\n\tv_cmp_le_i32_e64 s[20:21], 0, %5\n\t"
#define OPS(F) \
    F(mix_perm_add_independent_x2insts, M_PERM_ADD_IND) F(mix_add_add_perm_perm_x4insts, M_ADD_ADD_PERM_PERM) \
    F(mix_add_perm_add_perm_x4insts, M_ADD_PERM_ADD_PERM) F(mix_mul_sub_sub_or_x4insts, M_MUL_SUB_SUB_OR) \
    F(mix_add_sub_or_and_x4insts, M_ADD4_IND) F(mix_4add_1ldsread_x5insts, M_ADD_LDS) F(mix_add_smov_add_smov_x4insts, M_ADD_SALU) \
    F(mix_4vmov_sgpr_x4insts, M_VMOV_SGPR) F(mix_walk_step_3perm_mul_7simple_cmp_x12insts, M_STEP_V)

extern "C" __global__ void k_dummy() {}

#define DEFINE_KERNEL(NAME, INS)                                                                                     \
    template <bool DEP>                                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(int iters, unsigned long long* cycles, unsigned* sink) {         \
        extern __shared__ unsigned char smem_[];                                                                     \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
                 a7 = a0 + 7;                                                                                        \
        const unsigned b = 0x00030005u + (threadIdx.x & 1), c = (threadIdx.x & 63) * 4u;                                         \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int it = 0; it < iters; ++it) {                                                                         \
            if (DEP)                                                                                                 \
                asm volatile(X8(DEP8(INS))                                                                           \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                             : "v"(b), "v"(c)                                                                        \
                             : "vcc", "s20", "s21");                                                                 \
            else                                                                                                     \
                asm volatile(X8(REP8(INS))                                                                           \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                             : "v"(b), "v"(c)                                                                        \
                             : "vcc", "s20", "s21");                                                                 \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                          \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) sink[0] = smem_[0];                                 \
    }
OPS(DEFINE_KERNEL)

struct Op {
    const char* name;
    void (*indep)(int, unsigned long long*, unsigned*);
    void (*dep)(int, unsigned long long*, unsigned*);
};
#define OP_ENTRY(NAME, INS) {#NAME, k_##NAME<false>, k_##NAME<true>},
static const Op g_ops[] = {OPS(OP_ENTRY)};

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    int clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, dev));
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    unsigned long long* d_cycles;
    unsigned* d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * cus * 8 * 4));
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(cus * 8 * 4);
    const int Ws[] = {1, 2, 3, 4, 5, 6, 8};
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"iters\": %d, \"insts_per_wave\": %d,\n",
                prop.name[0] ? prop.name : prop.gcnArchName, cus, clock_khz, iters, iters * 64);
    std::printf(" \"method\": \"W workgroups of 4 waves per CU (one wave per SIMD each), every wave issues iters*64 copies of one "
                "instruction; rate = W*insts / cycles of the slowest wave (s_memtime), wave-instructions per cycle per SIMD; "
                "ghz = insts*W*4*cus/(rate*... ) see wall_ms\",\n \"ops\": {\n");
    bool first = true;
    for (const Op& op : g_ops) {
        for (int dep = 0; dep < 2; ++dep) {
            auto fn = dep ? op.dep : op.indep;
            std::printf("%s  \"%s/%s\": {", first ? "" : ",\n", op.name, dep ? "dep" : "indep");
            first = false;
            bool f2 = true;
            for (int W : Ws) {
                const size_t lds = (160 * 1024) / W - 64;
                CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const int grid = cus * W;
                hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, 10, d_cycles, d_sink);   // warm-up
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, iters, d_cycles, d_sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
                unsigned long long mx = 0, sum = 0;
                for (int i = 0; i < grid * 4; ++i) { mx = std::max(mx, h[i]); sum += h[i]; }
                const double insts = (double)iters * 64.0;
                const double rate_max = W * insts / (double)mx, rate_avg = W * insts / ((double)sum / (grid * 4));
                // wall-clock rate: wave-insts per ns per SIMD -> divide by GHz to compare; report raw
                const double per_ns = W * insts / (ms * 1e6);
                std::printf("%s\"%d\": {\"rate\": %.4f, \"rate_avg_wave\": %.4f, \"wave_insts_per_ns_per_simd\": %.4f, \"wall_ms\": %.4f}",
                            f2 ? "" : ", ", W, rate_max, rate_avg, per_ns, ms);
                f2 = false;
            }
            std::printf("}");
            std::fflush(stdout);
        }
    }
    std::printf("\n }\n}\n");
    return 0;
}
