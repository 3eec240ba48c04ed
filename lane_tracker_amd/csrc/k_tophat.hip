// Elliptical erode / dilate / top-hat for the 29x29 and 55x55 structuring elements
// (morphologyEx(MORPH_TOPHAT), lane_tracker.py:203-204, 210-211), gfx950.
//
// Exact decomposition of the flat ellipse: the footprint is one horizontal run per row, so
//     erode(y, x) = min over SE rows i of  H_{dx[i]}(y + i - r, x),
// where H_d is the horizontal window minimum of half-width d.  Runs nest, so all distinct H_d of
// one image row come from a short chain of "min of two shifted copies" steps (half-width
// 0 -> 1 -> 2 -> 4 -> 7 -> 14 for 55x55), and every final H_d is min(S[x-t], S[x+t]) of one chain
// plane.  That is ~20 min-ops per pixel for the horizontal part instead of 2337 taps.
//
// Mapping (one 64-lane wave = one task, no workgroup barriers):
//   * a wave owns a strip of 128 output columns and walks down a band of rows;
//   * lane l holds columns (x0 + l, x0 + 64 + l) packed as u16x2 in one VGPR, so every min/max
//     is one v_pk_min_u16 / v_pk_max_u16 for two pixels and a horizontal shift by t columns is a
//     shift by t lanes for both halves at once;
//   * the horizontal chain of the current row lives in a per-wave LDS scratch (6 planes x 128
//     dwords); LDS operations of one wave execute in order, so only compiler-level wavefront
//     fences separate the chain steps;
//   * the vertical combine never touches memory: K accumulators A[0..K-1] per lane hold the K
//     output rows in flight.  For every input row:  A[j] = min(A[j+1], H[slot(j)])  -- the shift
//     of the window is free because the instruction has a separate destination -- and A[0] is a
//     finished output row.
// Out-of-image taps are neutral (255 for erode, 0 for dilate), as in OpenCV's default border.
#include "lt_internal.h"

namespace lt {
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <bool DIL>
__device__ __forceinline__ uint32_t pk(uint32_t a, uint32_t b) {
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    const u16x2 r = DIL ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(uint32_t, r);
}

// order LDS traffic between lanes of one wave (hardware keeps a wave's DS ops in order; this
// only stops the compiler from moving a load above the store of another lane's value)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int PLANE = 160;  // dwords per chain plane: 128 entries + 16 dwords of margin on both sides, so that
constexpr int MARGIN = 16;  // every lane can run every chain step unpredicated (out-of-range entries are never consumed)

// ---- 55x55: half-widths 0,7,10,12,14,16,17,...,27 (17 distinct) -------------------------------------
struct SE55 {
    static constexpr int K = 55, R = 27, NH = 17, NPLANES = 6;
    // slot of the distinct half-width used by SE row j (rows 0..27, mirrored for 28..54)
    static constexpr int half_slot[28] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11, 12, 13, 13, 14, 14, 14,
                                          15, 15, 15, 16, 16, 16, 16, 16, 16};
    static constexpr int slot_width[17] = {0, 7, 10, 12, 14, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27};
    static constexpr int slot(int j) { return half_slot[j <= R ? j : K - 1 - j]; }

    template <bool DIL>
    static __device__ __forceinline__ void row_windows(uint32_t* s, int lane, uint32_t (&H)[NH]) {
        uint32_t* S0 = s + MARGIN;
        uint32_t* S1 = S0 + PLANE;
        uint32_t* S2 = S0 + 2 * PLANE;
        uint32_t* S4 = S0 + 3 * PLANE;
        uint32_t* S7 = S0 + 4 * PLANE;
        uint32_t* S14 = S0 + 5 * PLANE;
        const int pa = lane, pb = lane + 64;  // the two chain entries this lane computes
        // Valid ranges shrink step by step (S1 on [1,116], S2 [2,115], S4 [4,113], S7 [7,110],
        // S14 [14,103]); entries outside them hold garbage that no valid output ever reads.
        S1[pa] = pk<DIL>(pk<DIL>(S0[pa - 1], S0[pa + 1]), S0[pa]);
        S1[pb] = pk<DIL>(pk<DIL>(S0[pb - 1], S0[pb + 1]), S0[pb]);
        wave_lds_fence();
        S2[pa] = pk<DIL>(S1[pa - 1], S1[pa + 1]);
        S2[pb] = pk<DIL>(S1[pb - 1], S1[pb + 1]);
        wave_lds_fence();
        S4[pa] = pk<DIL>(S2[pa - 2], S2[pa + 2]);
        S4[pb] = pk<DIL>(S2[pb - 2], S2[pb + 2]);
        wave_lds_fence();
        S7[pa] = pk<DIL>(S4[pa - 3], S4[pa + 3]);
        S7[pb] = pk<DIL>(S4[pb - 3], S4[pb + 3]);
        wave_lds_fence();
        S14[pa] = pk<DIL>(S7[pa - 7], S7[pa + 7]);
        S14[pb] = pk<DIL>(S7[pb - 7], S7[pb + 7]);
        wave_lds_fence();
        const int p = R + lane;  // this lane's own columns
        H[0] = S0[p];
        H[1] = S7[p];
        H[2] = pk<DIL>(S7[p - 3], S7[p + 3]);
        H[3] = pk<DIL>(S7[p - 5], S7[p + 5]);
        H[4] = S14[p];
#pragma unroll
        for (int t = 2; t <= 13; ++t) H[3 + t] = pk<DIL>(S14[p - t], S14[p + t]);   // widths 16..27
    }
};

// ---- 29x29: half-widths 0,5,7,9,10,11,12,13,14 (9 distinct) -------------------------------------------
struct SE29 {
    static constexpr int K = 29, R = 14, NH = 9, NPLANES = 5;
    static constexpr int half_slot[15] = {0, 1, 2, 3, 4, 5, 5, 6, 7, 7, 7, 8, 8, 8, 8};
    static constexpr int slot_width[9] = {0, 5, 7, 9, 10, 11, 12, 13, 14};
    static constexpr int slot(int j) { return half_slot[j <= R ? j : K - 1 - j]; }

    template <bool DIL>
    static __device__ __forceinline__ void row_windows(uint32_t* s, int lane, uint32_t (&H)[NH]) {
        uint32_t* S0 = s + MARGIN;
        uint32_t* S1 = S0 + PLANE;
        uint32_t* S2 = S0 + 2 * PLANE;
        uint32_t* S4 = S0 + 3 * PLANE;
        uint32_t* S7 = S0 + 4 * PLANE;
        const int pa = lane, pb = lane + 64;  // entries 0..91 are real; valid: S1 [1,90], S2 [2,89], S4 [4,87], S7 [7,84]
        S1[pa] = pk<DIL>(pk<DIL>(S0[pa - 1], S0[pa + 1]), S0[pa]);
        S1[pb] = pk<DIL>(pk<DIL>(S0[pb - 1], S0[pb + 1]), S0[pb]);
        wave_lds_fence();
        S2[pa] = pk<DIL>(S1[pa - 1], S1[pa + 1]);
        S2[pb] = pk<DIL>(S1[pb - 1], S1[pb + 1]);
        wave_lds_fence();
        S4[pa] = pk<DIL>(S2[pa - 2], S2[pa + 2]);
        S4[pb] = pk<DIL>(S2[pb - 2], S2[pb + 2]);
        wave_lds_fence();
        S7[pa] = pk<DIL>(S4[pa - 3], S4[pa + 3]);
        S7[pb] = pk<DIL>(S4[pb - 3], S4[pb + 3]);
        wave_lds_fence();
        const int p = R + lane;
        H[0] = S0[p];
        H[1] = pk<DIL>(S4[p - 1], S4[p + 1]);
        H[2] = S7[p];
#pragma unroll
        for (int t = 2; t <= 7; ++t) H[1 + t] = pk<DIL>(S7[p - t], S7[p + t]);     // widths 9..14
    }
};

struct RunsGeom {
    int h, w, nstrips, nbands, band_rows, ntasks;
    size_t plane_stride;
};

// one chain entry = pixels (col_a, col_a + 64) of a row, packed.  The loads are unconditional on
// clamped addresses and masked afterwards, so that they can stay in flight across the chain
// (a guarded load compiles to branch + load + s_waitcnt vmcnt(0)).
template <class SE>
__device__ __forceinline__ uint32_t load_entry(const uint8_t* __restrict__ row, bool row_ok, int col_a, int w,
                                              uint32_t neutral) {
    const int col_b = col_a + 64;
    const uint32_t n8 = neutral & 0xffu;  // `neutral` is the packed pair; each half defaults to the 8-bit value
    const uint32_t ra = row[min(max(col_a, 0), w - 1)], rb = row[min(max(col_b, 0), w - 1)];
    const uint32_t a = (row_ok && col_a >= 0 && col_a < w) ? ra : n8;
    const uint32_t b = (row_ok && col_b >= 0 && col_b < w) ? rb : n8;
    return a | (b << 16);
}

// DIL = false: dst = erode(src);  DIL = true: dst = dilate(src), or minuend - dilate(src) (top-hat)
template <class SE, bool DIL>
__global__ __launch_bounds__(256) void k_morph_runs(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                   const uint8_t* __restrict__ minuend, RunsGeom g) {
    __shared__ uint32_t s_chain[4][SE::NPLANES * PLANE];
    constexpr int K = SE::K, R = SE::R, NH = SE::NH;
    constexpr uint32_t NEUTRAL = DIL ? 0u : 0x00ff00ffu;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wv;
    if (task >= g.ntasks) return;  // whole wave exits together; no block-level barriers are used
    const int strip = task % g.nstrips;
    const int band = (task / g.nstrips) % g.nbands;
    const int frame = task / (g.nstrips * g.nbands);
    const uint8_t* s = src + (size_t)frame * g.plane_stride;
    uint8_t* d = dst + (size_t)frame * g.plane_stride;
    const uint8_t* m = minuend ? minuend + (size_t)frame * g.plane_stride : nullptr;
    uint32_t* chain = s_chain[wv];
    const int x0 = strip * 128;
    const int yb0 = band * g.band_rows, yb1 = min(yb0 + g.band_rows, g.h);
    const int xa = x0 + lane, xb = xa + 64;

    uint32_t A[K];
#pragma unroll
    for (int j = 0; j < K; ++j) A[j] = NEUTRAL;

    const int y_first = yb0 - R, y_last = yb1 - 1 + R;
    const bool second = lane + 64 < 64 + 2 * R;  // does this lane own a second chain entry?
    // software prefetch of the next row's raw entries
    bool ok = y_first >= 0 && y_first < g.h;
    uint32_t e0 = load_entry<SE>(s + (size_t)max(y_first, 0) * g.w, ok, x0 - R + lane, g.w, NEUTRAL);
    uint32_t e1 = second ? load_entry<SE>(s + (size_t)max(y_first, 0) * g.w, ok, x0 - R + lane + 64, g.w, NEUTRAL) : NEUTRAL;
    for (int yy = y_first; yy <= y_last; ++yy) {
        chain[MARGIN + lane] = e0;
        chain[MARGIN + lane + 64] = e1;
        wave_lds_fence();
        const bool in_img = yy >= 0 && yy < g.h;
        // issue the next row's loads before the chain so that their latency overlaps it
        const int yn = yy + 1;
        const bool okn = yn >= 0 && yn < g.h && yn <= y_last;
        const uint8_t* rown = s + (size_t)min(max(yn, 0), g.h - 1) * g.w;
        e0 = load_entry<SE>(rown, okn, x0 - R + lane, g.w, NEUTRAL);
        e1 = second ? load_entry<SE>(rown, okn, x0 - R + lane + 64, g.w, NEUTRAL) : NEUTRAL;

        uint32_t H[NH];
        if (in_img) {
            SE::template row_windows<DIL>(chain, lane, H);
        } else {
#pragma unroll
            for (int i = 0; i < NH; ++i) H[i] = NEUTRAL;
        }
        wave_lds_fence();  // the chain planes are rewritten by the next iteration
#pragma unroll
        for (int j = 0; j < K - 1; ++j) A[j] = pk<DIL>(A[j + 1], H[SE::slot(j)]);
        A[K - 1] = H[SE::slot(K - 1)];
        const int y = yy - R;
        if (y >= yb0 && y < yb1) {
            const uint32_t va = A[0] & 0xffffu, vb = A[0] >> 16;
            const size_t o = (size_t)y * g.w;
            if (xa < g.w) {
                uint32_t v = va;
                if (m) { const uint32_t mm = m[o + xa]; v = mm > v ? mm - v : 0u; }
                d[o + xa] = (uint8_t)v;
            }
            if (xb < g.w) {
                uint32_t v = vb;
                if (m) { const uint32_t mm = m[o + xb]; v = mm > v ? mm - v : 0u; }
                d[o + xb] = (uint8_t)v;
            }
        }
    }
}

template <class SE>
bool table_matches(const EllipseSE& se) {
    if (se.k != SE::K) return false;
    for (int j = 0; j < SE::K; ++j)
        if (SE::slot_width[SE::slot(j)] != se.dx[j]) return false;
    return true;
}

template <class SE>
void launch_runs(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w, bool dilate,
                 size_t plane_stride, int n) {
    RunsGeom g;
    g.h = h;
    g.w = w;
    g.plane_stride = plane_stride;
    g.nstrips = (w + 127) / 128;
    // enough wave-tasks to fill 256 CUs x ~16 waves, but bands no shorter than ~2x the halo
    const int min_rows = 4 * SE::R;
    int nbands = 1;
    while ((long long)n * g.nstrips * nbands < 6144 && (h + nbands) / (nbands + 1) >= min_rows) ++nbands;
    g.band_rows = (h + nbands - 1) / nbands;
    g.nbands = (h + g.band_rows - 1) / g.band_rows;
    g.ntasks = n * g.nstrips * g.nbands;
    dim3 grid((g.ntasks + 3) / 4);
    if (dilate)
        hipLaunchKernelGGL((k_morph_runs<SE, true>), grid, dim3(256), 0, s, src, dst, minuend, g);
    else
        hipLaunchKernelGGL((k_morph_runs<SE, false>), grid, dim3(256), 0, s, src, dst, minuend, g);
}

}  // namespace

// true if the compiled-in run tables equal the structuring element OpenCV's formula gives
bool tophat_tables_match(const EllipseSE& se29, const EllipseSE& se55) {
    return table_matches<SE29>(se29) && table_matches<SE55>(se55);
}

void launch_morph_runs(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w, int k,
                       bool dilate, size_t plane_stride, int n) {
    if (n <= 0 || h <= 0 || w <= 0) return;
    if (k == 55)
        launch_runs<SE55>(s, src, dst, minuend, h, w, dilate, plane_stride, n);
    else
        launch_runs<SE29>(s, src, dst, minuend, h, w, dilate, plane_stride, n);
}

}  // namespace lt
