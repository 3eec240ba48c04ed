// Minimal form of the code-generation case behind the register barrier in k_warp_split4 (csrc/k_frontend.hip).
//
// hipcc (ROCm 7.2, AMD clang 22) turns two "shift right by 15, clamp to 0..255" results that are packed into the low
// bytes of a word into ONE gfx950 instruction, v_ashr_pk_u8_i32, and ORs the third byte in with v_lshl_or_b32 -- which
// is only right if v_ashr_pk_u8_i32 leaves bits 31:16 of its destination zero.  This program packs three such values
//   (a) in plain C++ (the compiler is free to use the instruction),
//   (b) with the values made opaque first (what the product does),
// and (c) issues the instruction by hand on a destination preloaded with ones.  It prints one line per check;
// tests/test_gpu_toolchain_cases.py reads them.
//   build: hipcc -O3 --offload-arch=gfx950 ashr_pk_u8.hip -o ashr_pk_u8
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int KADD = 128 * (1 << 15) + (1 << 14);
__host__ __device__ __forceinline__ int lab_tail(int d) {      // the last two lines of lab_b_of() in k_frontend.hip
    const int v = (200 * d + KADD) >> 15;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// four clamped values packed into one word, the way k_warp_split4 packs its four Lab-b pixels.  ROCm 7.2 compiles the
// plain form to   v_ashr_pk_u8_i32 v2, v2, v3, 15 ; ... ; v_lshl_or_b32 v2, v3, 16, v2 ; v_lshl_or_b32 v2, v4, 24, v2
template <bool OPAQUE>
__global__ void k_pack4(const int* __restrict__ x, uint32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int b = lab_tail(x[4 * i + j]);
        if (OPAQUE) asm volatile("" : "+v"(b));
        o |= ((uint32_t)b & 255u) << (8 * j);
    }
    out[i] = o;
}

__global__ void k_raw(const int* __restrict__ x, uint32_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0xffffffffu;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 15" : "+v"(d) : "v"(x[4 * i]), "v"(x[4 * i + 1]));
    out[i] = d;
}

int main() {
    const int n = 1 << 16;
    std::vector<int> h(4 * n);
    std::srand(7);
    for (auto& v : h) v = (int)(std::rand() % 60001) - 30000;   // fY - fZ of two 15-bit table entries: negative, in-range and saturating results
    int* dx;
    uint32_t* dout;
    if (hipMalloc(&dx, h.size() * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) return 2;
    hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> got(n);
    auto expect = [&](int i) {
        return (uint32_t)lab_tail(h[4 * i]) | ((uint32_t)lab_tail(h[4 * i + 1]) << 8) | ((uint32_t)lab_tail(h[4 * i + 2]) << 16) |
               ((uint32_t)lab_tail(h[4 * i + 3]) << 24);
    };
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(k_pack4<false>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
        else hipLaunchKernelGGL(k_pack4<true>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
        hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < n; ++i)
            if (got[i] != expect(i)) { if (first < 0) first = i; ++bad; }
        std::printf("%s: %d of %d words wrong", variant == 0 ? "plain" : "opaque", bad, n);
        if (first >= 0) std::printf(" (first: inputs %d %d %d %d -> 0x%08x, expected 0x%08x)", h[4 * first], h[4 * first + 1], h[4 * first + 2], h[4 * first + 3], got[first], expect(first));
        std::printf("\n");
    }
    hipLaunchKernelGGL(k_raw, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int upper_zero = 0, upper_kept = 0, low_ok = 0;
    for (int i = 0; i < n; ++i) {
        auto c = [](int v) { v >>= 15; return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        low_ok += (got[i] & 0xffffu) == (c(h[4 * i]) | (c(h[4 * i + 1]) << 8));
        upper_zero += (got[i] >> 16) == 0;
        upper_kept += (got[i] >> 16) == 0xffffu;
    }
    std::printf("v_ashr_pk_u8_i32: low 16 bits as documented in %d of %d, bits 31:16 zero in %d, preserved from the destination in %d (e.g. 0x%08x)\n",
                low_ok, n, upper_zero, upper_kept, got[0]);
    return 0;
}
