// Vector-memory issue cost on gfx950 by instruction shape, with cache-resident data: how many cycles of a CU's
// texture-address / data path one wave-instruction occupies.  Every wave sweeps its own 1 KiB (or, for the scattered
// shapes, 4 KiB) region again and again, so after the first pass nothing leaves the CU's L1 / the XCD's L2; 16 waves per
// CU (4 per SIMD) issue 8 independent loads (or stores) per loop trip.  Reported: ns and cycles (at the clock the runtime
// reports) per wave-instruction per CU -- the figure to multiply by a kernel's VMEM instructions per CU.
//
//   build: hipcc -O2 --offload-arch=gfx950 vmem_issue.hip -o vmem_issue ;  run: ./vmem_issue [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Mode {
    LD_U8,            // buffer_load_ubyte, lane i -> byte i               (the old top-hat row fetch)
    LD_U16,           // buffer_load_ushort, lane i -> bytes 2i..
    LD_B32,           // buffer_load_dword, lane i -> dword i
    LD_B32_SHARE4,    // buffer_load_dword, lanes 4k..4k+3 -> dword k      (the new top-hat row fetch)
    LD_B32_UNAL,      // buffer_load_dword at byte offset i                (unaligned, overlapping)
    LD_B64,           // buffer_load_dwordx2, lane i -> 8 bytes at 8i
    LD_B64_UNAL3,     // buffer_load_dwordx2 at byte offset 3i             (the undistort tap pair)
    LD_B64_TAP,       // buffer_load_dwordx2 at 4-byte aligned offsets 12i (a gather with some reuse: the warp taps)
    LD_B128,          // buffer_load_dwordx4, lane i -> 16 bytes at 16i
    ST_U8, ST_B32, ST_B64, ST_B128,
    LD_U8_ROWS,       // buffer_load_ubyte, lane i -> byte i of a NEW 1080-byte row per instruction (the old top-hat fetch: L1 misses)
    LD_B32_SHARE4_ROWS,   // buffer_load_dword, lanes 4k..4k+3 -> dword k of a new row per instruction (the new one)
    N_MODES
};
static const char* kNames[N_MODES] = {"load_ubyte", "load_ushort", "load_dword", "load_dword_4lanes_share", "load_dword_unaligned_1B_pitch",
                                      "load_dwordx2", "load_dwordx2_unaligned_3B_pitch", "load_dwordx2_12B_pitch", "load_dwordx4",
                                      "store_byte", "store_dword", "store_dwordx2", "store_dwordx4",
                                      "load_ubyte_new_row_each", "load_dword_4lanes_share_new_row_each"};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_vmem(uint8_t* __restrict__ buf, int iters, unsigned* sink, unsigned magic) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
    constexpr int REGION = 8192;
    uint8_t* base = buf + wave * REGION;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, REGION, 0x00027000);
    int voff = 0;
    switch (MODE) {
        case LD_U8: case ST_U8: case LD_B32_UNAL: voff = lane; break;
        case LD_U16: voff = 2 * lane; break;
        case LD_B32: case ST_B32: voff = 4 * lane; break;
        case LD_B32_SHARE4: voff = lane & ~3; break;
        case LD_B64: case ST_B64: voff = 8 * lane; break;
        case LD_B64_UNAL3: voff = 3 * lane; break;
        case LD_B64_TAP: voff = 12 * lane; break;
        case LD_B128: case ST_B128: voff = 16 * lane; break;
    }
    unsigned acc = magic ^ lane;
    if constexpr (MODE == LD_U8_ROWS || MODE == LD_B32_SHARE4_ROWS) {
        // every wave walks down its own 128-byte column of a big plane, one 1080-byte row per load: nothing is in the L1
        constexpr int PITCH = 1080, ROWS = 3600;           // 3.9 MB per wave, 16 GB in total would not fit: waves share planes of 64 MiB
        uint8_t* plane = buf + (wave % 16) * (size_t)(64u << 20) / 16 + (wave / 16 % 8) * 128;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(plane, 0, PITCH * ROWS, 0x00027000);
        const int vo = MODE == LD_U8_ROWS ? lane : (lane & ~3);
        int row = (int)(wave % 97);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int o = row * PITCH;
                row = row + 1 >= ROWS ? 0 : row + 1;
                if constexpr (MODE == LD_U8_ROWS) acc ^= __builtin_amdgcn_raw_buffer_load_b8(prs, vo, o, 0);
                else acc ^= __builtin_amdgcn_raw_buffer_load_b32(prs, vo, o, 0);
            }
        }
        if (acc == 0x12345678u) sink[0] = acc;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        const int so = (it & 3) * 1024;   // scalar offset changes every trip: nothing can be hoisted or merged
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = so + k * 64;    // each of the 8 instructions starts 64 bytes further: distinct addresses, same lines
            if constexpr (MODE == LD_U8) acc ^= __builtin_amdgcn_raw_buffer_load_b8(rs, voff, o, 0);
            else if constexpr (MODE == LD_U16) acc ^= __builtin_amdgcn_raw_buffer_load_b16(rs, voff, o, 0);
            else if constexpr (MODE == LD_B32 || MODE == LD_B32_SHARE4 || MODE == LD_B32_UNAL) acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, voff, o, 0);
            else if constexpr (MODE == LD_B64 || MODE == LD_B64_UNAL3 || MODE == LD_B64_TAP) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, o, 0); acc ^= v.x ^ v.y; }
            else if constexpr (MODE == LD_B128) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, o, 0); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            else if constexpr (MODE == ST_U8) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(acc + k), rs, voff, o, 0);
            else if constexpr (MODE == ST_B32) __builtin_amdgcn_raw_buffer_store_b32(acc + k, rs, voff, o, 0);
            else if constexpr (MODE == ST_B64) { u32x2 v = {acc + k, acc}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, o, 0); }
            else if constexpr (MODE == ST_B128) { u32x4 v = {acc + k, acc, acc, acc}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, o, 0); }
        }
        if (MODE >= ST_U8) acc = acc * 3 + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(uint8_t* buf, unsigned* sink, int cus, int iters, double ghz, bool last) {
    const int blocks = cus * 4;   // 16 waves per CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, iters / 8, sink, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, iters, sink, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double insts_per_cu = 16.0 * iters * 8.0;
    const double ns = best * 1e6 / insts_per_cu;
    std::printf("  \"%s\": {\"wall_ms\": %.4f, \"ns_per_wave_inst_per_cu\": %.3f, \"cycles_per_wave_inst_per_cu\": %.2f}%s\n", kNames[MODE], best, ns,
                ns * ghz, last ? "" : ",");
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
    int dev = 0, cus = 0, khz = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
    const double ghz = khz * 1e-6;
    const size_t bytes = std::max((size_t)cus * 16 * 8192, (size_t)(72u << 20));
    uint8_t* buf = nullptr;
    unsigned* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipMalloc(&sink, 64));
    std::printf("{\"cus\": %d, \"clock_ghz\": %.3f, \"waves_per_cu\": 16, \"iters\": %d, \"loads_or_stores_per_trip\": 8, \"modes\": {\n", cus, ghz, iters);
    run<LD_U8>(buf, sink, cus, iters, ghz, false);
    run<LD_U16>(buf, sink, cus, iters, ghz, false);
    run<LD_B32>(buf, sink, cus, iters, ghz, false);
    run<LD_B32_SHARE4>(buf, sink, cus, iters, ghz, false);
    run<LD_B32_UNAL>(buf, sink, cus, iters, ghz, false);
    run<LD_B64>(buf, sink, cus, iters, ghz, false);
    run<LD_B64_UNAL3>(buf, sink, cus, iters, ghz, false);
    run<LD_B64_TAP>(buf, sink, cus, iters, ghz, false);
    run<LD_B128>(buf, sink, cus, iters, ghz, false);
    run<ST_U8>(buf, sink, cus, iters, ghz, false);
    run<ST_B32>(buf, sink, cus, iters, ghz, false);
    run<ST_B64>(buf, sink, cus, iters, ghz, false);
    run<ST_B128>(buf, sink, cus, iters, ghz, false);
    run<LD_U8_ROWS>(buf, sink, cus, iters, ghz, false);
    run<LD_B32_SHARE4_ROWS>(buf, sink, cus, iters, ghz, true);
    std::printf("}}\n");
    return 0;
}
