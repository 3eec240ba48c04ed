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
//
// The above is the first formulation, k_morph_runs (one row per iteration, kept for A/B measurements).  The kernel the
// library runs is k_morph_runs2 further down: two rows per iteration on v_pk_minimum3_f16 / v_pk_maximum3_f16, a three-step
// chain (0 -> 1 by DPP, -> 4 -> 13 through LDS), clamped borders instead of neutral fills, aligned dword loads, outputs
// regrouped into dword stores, pair strips; its comments describe each of these where it happens.
#include <cstdlib>

#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "lt_internal.h"

#ifndef LT_DPP_SELECT
#define LT_DPP_SELECT 0
#endif
#ifndef LT_FUSED55
#define LT_FUSED55 1
#endif

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
    int dpitch;                  // row pitch of the destination (w, or a 64-byte multiple for the threshold walks)
    size_t dst_stride;           // bytes per frame of the destination
    int nframes;                 // frames of the launch
    int nstrips_normal;          // strips handled one frame per wave; nstrips - 1 when the last strip is a PAIR strip (below)
    int n_normal;                // tasks of those strips: nframes * nstrips_normal * nbands; the tasks behind them are pair tasks
    int xcd;                     // 1: workgroups are renumbered so that each XCD (= each L2) owns one contiguous range of tasks
    uint8_t* copy_dst;           // COPYM kernels: the minuend is also stored here, with the destination's pitch and stride
};

// Per-lane column bookkeeping, loop invariant: clamped byte offsets of the (up to) four pixels a
// lane fetches per row, and which of them exist.  Loads are unconditional on the clamped offsets
// and masked afterwards, so they stay in flight across the chain (a guarded load compiles to
// branch + load + s_waitcnt vmcnt(0)).
struct LaneCols {
    uint32_t off[4];   // entry 0: (a, a+64); entry 1: (a+64, a+128); unsigned: the loads take the scalar-base + 32-bit-offset form
    uint32_t keep[2];  // per entry: 0xffff / 0xffff0000 bits set where the pixel exists
    uint32_t fill[2];  // per entry: neutral value in the halves that do not exist
};

__device__ __forceinline__ LaneCols lane_cols(int col0, int w, bool second, uint32_t n8) {
    LaneCols c;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ca = col0 + 64 * e, cb = ca + 64;
        const bool va = ca >= 0 && ca < w && (e == 0 || second), vb = cb >= 0 && cb < w && (e == 0 || second);
        c.off[2 * e] = (uint32_t)min(max(ca, 0), w - 1);
        c.off[2 * e + 1] = (uint32_t)min(max(cb, 0), w - 1);
        c.keep[e] = (va ? 0xffffu : 0u) | (vb ? 0xffff0000u : 0u);
        c.fill[e] = (va ? 0u : n8) | (vb ? 0u : (n8 << 16));
    }
    return c;
}

__device__ __forceinline__ uint32_t fetch_entry(const uint8_t* __restrict__ row, const LaneCols& c, int e) {
    const uint32_t a = row[c.off[2 * e]], b = row[c.off[2 * e + 1]];
    return ((a | (b << 16)) & c.keep[e]) | c.fill[e];
}

// The same entry through a buffer descriptor of the frame's plane: buffer_load_ubyte takes the descriptor, a 32-bit
// lane offset and a scalar row offset, so a load costs no VALU address arithmetic (the flat form pays one 64-bit
// add per load) and the row base is one s_mul instead of a 64-bit pointer computation.
__device__ __forceinline__ uint32_t fetch_entry(__amdgpu_buffer_rsrc_t plane, int row_off, const LaneCols& c, int e) {
    const uint32_t a = __builtin_amdgcn_raw_buffer_load_b8(plane, (int)c.off[2 * e], row_off, 0);
    const uint32_t b = __builtin_amdgcn_raw_buffer_load_b8(plane, (int)c.off[2 * e + 1], row_off, 0);
    return ((a | (b << 16)) & c.keep[e]) | c.fill[e];
}

// DIL = false: dst = erode(src);  DIL = true: dst = dilate(src), or minuend - dilate(src) (top-hat)
template <class SE, bool DIL>
__global__ __launch_bounds__(256) void k_morph_runs(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                   const uint8_t* __restrict__ minuend, RunsGeom g) {
    __shared__ uint32_t s_chain[4][SE::NPLANES * PLANE];
    constexpr int K = SE::K, R = SE::R, NH = SE::NH;
    constexpr uint32_t NEUTRAL = DIL ? 0u : 0x00ff00ffu;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // wave-uniform by construction; readfirstlane tells the compiler, so that the row loop, its bounds
    // and every row base address live in SGPRs and the loads take the scalar-base + lane-offset form
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (task >= g.ntasks) return;  // whole wave exits together; no block-level barriers are used
    const int strip = task % g.nstrips;
    const int band = (task / g.nstrips) % g.nbands;
    const int frame = task / (g.nstrips * g.nbands);
    const uint8_t* s = src + (size_t)frame * g.plane_stride;
    uint8_t* d = dst + (size_t)frame * g.dst_stride;
    const uint8_t* m = minuend ? minuend + (size_t)frame * g.plane_stride : nullptr;
    uint32_t* chain = s_chain[wv];
    const int x0 = strip * 128;
    const int yb0 = band * g.band_rows, yb1 = min(yb0 + g.band_rows, g.h);
    const int xa = x0 + lane, xb = xa + 64;
    const bool va = xa < g.w, vb = xb < g.w;
    const int oa = min(xa, g.w - 1), ob = min(xb, g.w - 1);   // clamped output / minuend columns

    uint32_t A[K];
#pragma unroll
    for (int j = 0; j < K; ++j) A[j] = NEUTRAL;

    const int y_first = yb0 - R, y_last = yb1 - 1 + R;
    const LaneCols cols = lane_cols(x0 - R + lane, g.w, lane + 64 < 64 + 2 * R, NEUTRAL & 0xffu);
    // software prefetch: the raw entries of the next input row and the minuend of the next output row
    const uint8_t* row = s + (size_t)min(max(y_first, 0), g.h - 1) * g.w;
    uint32_t e0 = fetch_entry(row, cols, 0), e1 = fetch_entry(row, cols, 1);
    uint32_t ma = 0, mb = 0;
    for (int yy = y_first; yy <= y_last; ++yy) {
        const bool in_img = yy >= 0 && yy < g.h;
        chain[MARGIN + lane] = in_img ? e0 : NEUTRAL;
        chain[MARGIN + lane + 64] = in_img ? e1 : NEUTRAL;
        wave_lds_fence();
        // issue the next iteration's loads before the chain so that their latency overlaps it
        row = s + (size_t)min(max(yy + 1, 0), g.h - 1) * g.w;
        e0 = fetch_entry(row, cols, 0);
        e1 = fetch_entry(row, cols, 1);
        const int y = yy - R;                  // the output row this iteration completes
        const uint32_t ma_cur = ma, mb_cur = mb;
        if (m) {
            const uint8_t* mrow = m + (size_t)min(max(y + 1, 0), g.h - 1) * g.w;
            ma = mrow[oa];
            mb = mrow[ob];
        }

        uint32_t H[NH];
        // rows outside the image enter the chain as all-neutral entries and come out neutral: no branch
        SE::template row_windows<DIL>(chain, lane, H);
        wave_lds_fence();  // the chain planes are rewritten by the next iteration
#pragma unroll
        for (int j = 0; j < K - 1; ++j) A[j] = pk<DIL>(A[j + 1], H[SE::slot(j)]);
        A[K - 1] = H[SE::slot(K - 1)];
        if (y >= yb0 && y < yb1) {
            uint32_t oa_v = A[0] & 0xffffu, ob_v = A[0] >> 16;
            if (m) {   // TOPHAT: src - open(src), saturating
                oa_v = ma_cur > oa_v ? ma_cur - oa_v : 0u;
                ob_v = mb_cur > ob_v ? mb_cur - ob_v : 0u;
            }
            const size_t o = (size_t)y * g.dpitch;
            if (va) d[o + xa] = (uint8_t)oa_v;
            if (vb) d[o + xb] = (uint8_t)ob_v;
        }
    }
}

// ================================================================================================
// Two input rows per iteration, 3-input packed min/max.
//
// gfx950 has v_pk_minimum3_f16 / v_pk_maximum3_f16.  A u8 value v as the 16-bit pattern 0x00vv is +0 or a positive
// f16 denormal, and non-negative f16 numbers order exactly like their bit patterns, so -- with f16 denormals kept, which
// is the mode HIP kernels run in and which the kernel sets in MODE itself -- the f16 min/max of such patterns is the
// integer min/max, bit for bit (the result is always one of the inputs).  -DLT_MORPH_BIAS=1 keeps the patterns normal
// numbers instead (0x0400 | v) at the price of one v_or per entry: 4 of the 83 / 131 VALU instructions per row pair, and
// these kernels are at 80-85 % of the VALU issue ceiling (profiles/r02_pmc_kernels.txt).  Processing rows (yy, yy+1) together turns two accumulator steps
//     A1[j] = min(A[j+1], Ha[s(j)]);  A2[j] = min(A1[j+1], Hb[s(j)])
// into one instruction  A2[j] = min3(A[j+2], Ha[s(j+1)], Hb[s(j)])  for two pixels: 27.5 instead of
// 54 accumulate ops per row, and half the loop overhead and LDS round trips per row.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
#ifndef LT_MORPH_BIAS
#define LT_MORPH_BIAS 0
#endif
constexpr uint32_t BIAS2 = LT_MORPH_BIAS ? 0x04000400u : 0u;

template <bool DIL>
__device__ __forceinline__ uint32_t op2(uint32_t a, uint32_t b) {
    const h16x2 x = __builtin_bit_cast(h16x2, a), y = __builtin_bit_cast(h16x2, b);
    return __builtin_bit_cast(uint32_t, DIL ? __builtin_elementwise_maximum(x, y) : __builtin_elementwise_minimum(x, y));
}
template <bool DIL>
__device__ __forceinline__ uint32_t op3(uint32_t a, uint32_t b, uint32_t c) {
    const h16x2 x = __builtin_bit_cast(h16x2, a), y = __builtin_bit_cast(h16x2, b), z = __builtin_bit_cast(h16x2, c);
    return __builtin_bit_cast(uint32_t, DIL ? __builtin_elementwise_maximum(x, __builtin_elementwise_maximum(y, z))
                                            : __builtin_elementwise_minimum(x, __builtin_elementwise_minimum(y, z)));
}
template <bool DIL>
__device__ __forceinline__ uint2 op2v(uint2 a, uint2 b) { return make_uint2(op2<DIL>(a.x, b.x), op2<DIL>(a.y, b.y)); }
template <bool DIL>
__device__ __forceinline__ uint2 op3v(uint2 a, uint2 b, uint2 c) {
    return make_uint2(op3<DIL>(a.x, b.x, c.x), op3<DIL>(a.y, b.y, c.y));
}

__device__ __forceinline__ uint32_t sub_sat16(uint32_t a, uint32_t b) {   // v_pk_sub_u16 clamp: max(a - b, 0) per half
    typedef unsigned short u16x2s __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2s, a), __builtin_bit_cast(u16x2s, b)));
}

// Chain of one row PAIR: entries are uint2 (.x = row yy, .y = row yy+1).  With 3-input ops the
// chain is shorter than in the one-row kernel: half-widths 0 -> 1 -> 4 -> 13 (55x55) or
// 0 -> 1 -> 4 (29x29); the first step is register-only (DPP).  A window of half-width d is
// the union of 2, 3 or 4 shifted windows of a chain plane:
//   from S4  (hw 4):  d = 4 + t  as {p-t, p+t}          for t <= 4,  as {p-t, p, p+t} for t <= 9,
//                     d = 14     as {p-10, p-3, p+3, p+10}  (29x29: = window 7 with S4[p -+ 10])
//   from S13 (hw 13): d = 13 + t as {p-t, p+t}          for t <= 13, as {p-t, p, p+t} for t = 14
// Valid entry ranges (55x55, entries 0..117): S1 [1,116], S4 [4,113], S13 [13,104]; the lane's own
// entries p = 27..90 read S13 at p +- 14 and S4 at p +- 8.  (29x29, entries 0..91): S1 [1,90],
// S4 [4,87]; p = 14..77 reads S4 at p +- 10.
// 55x55 only -- which stage of the fused finals (below) delivers half-width slot s, and the stage in which window
// row pair j (A[j] <- A[j+2], Ha[slot(j+1)], Hb[slot(j)]) has both of its slots
constexpr int fused_stage_of_slot(int s) { return s <= 3 ? 1 : s <= 7 ? 2 : s <= 11 ? 3 : s <= 14 ? 4 : 5; }
template <class SE>
constexpr int fused_stage_of_update(int j) {
    const int a = fused_stage_of_slot(SE::slot(j + 1)), b = fused_stage_of_slot(SE::slot(j));
    return a > b ? a : b;
}
template <class F, int... I>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// FUSE (55x55): the 35 window reads of a row pair held 74 VGPRs at once, next to the 55 of the vertical pipeline A --
// 161 VGPRs, three waves per SIMD.  Fused, the reads come in five stages ordered by half-width, and each stage is
// followed at once by the pipeline updates whose two half-widths it completes (outer rows of the ellipse first), so
// only one stage of reads and a few half-widths are alive at a time.
// Plane 0 holds only the 64 entries the half-width-0 window reads (entry R + lane = the lane's own columns): lane w stores
// its entry R.. as slot w (see row_pair), lane r reads slot (R + r) mod 64 at LDS address `s0_rd`.
template <class SE, bool DIL, bool FUSE = false>
__device__ __forceinline__ void row_windows2(uint2* s, int lane, uint2 e_pa, uint2 e_pb, uint32_t s0_rd, uint32_t (&Ha)[SE::NH], uint32_t (&Hb)[SE::NH],
                                             uint32_t* A = nullptr, uint32_t* out_ab = nullptr) {
    uint2* S0 = s + MARGIN;
    uint2* S4 = S0 + 2 * PLANE;
    uint2* SL = S0 + 3 * PLANE;   // S13 (55x55)
    const int pa = lane, pb = lane + 64;
    // Every LDS read of this function is a single-address ds_read_b64 issued by hand (256 B/clk in the LDS
    // pipe).  Left to the compiler they become ds_read2_b64, which moves the same bytes at half the rate,
    // and the kernel is LDS-pipe bound (SQ_WAIT_INST_LDS ~28 % with read2).  One s_waitcnt covers each
    // batch; the results are threaded through the wait statement so nothing is consumed before it.
#define LT_RD64(dst, base, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off) : "memory")
// Issue priority (LT_MORPH_PRIO bit 1, the default): a wave raises its priority (s_setprio 3) while it puts a stage's window
// reads into the LDS queue and drops it for the min / max work that follows -- the LDS pipe is the busier of the two co-bound
// units, so whoever can feed it goes first.  Measured per 256 frames (tools/ab_prio.sh, variant builds): 55x55 erode
// 0.490 -> 0.470 ms, top-hat 0.494 -> 0.474 ms; priority 2: 0.473 / 0.477; the other way round (-DLT_MORPH_PRIO=2, min / max
// first) 0.502 / 0.505; also around the chain steps, stage 1 and the 29x29 windows (bit 4): no gain, 29x29 unchanged.
#ifndef LT_MORPH_PRIO
#define LT_MORPH_PRIO 1
#endif
#ifndef LT_MORPH_PRIO_LEVEL
#define LT_MORPH_PRIO_LEVEL 3
#endif
#define LT_PRIO_READS() do { if (LT_MORPH_PRIO & 1) __builtin_amdgcn_s_setprio(LT_MORPH_PRIO_LEVEL); else if (LT_MORPH_PRIO == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define LT_PRIO_MATH()  do { if (LT_MORPH_PRIO & 1) __builtin_amdgcn_s_setprio(0); else if (LT_MORPH_PRIO == 2) __builtin_amdgcn_s_setprio(2); } while (0)
// bit 4: also around the reads of the chain steps, of stage 1 and of the 29x29 windows
#define LT_PRIO_READS2() do { if (LT_MORPH_PRIO & 4) __builtin_amdgcn_s_setprio(LT_MORPH_PRIO_LEVEL); } while (0)
#define LT_PRIO_MATH2()  do { if (LT_MORPH_PRIO & 4) __builtin_amdgcn_s_setprio(0); } while (0)
    const uint32_t cb = (uint32_t)(uintptr_t)(s + lane);   // low 32 bits of a flat LDS address = LDS offset
    auto pair = [](unsigned long long v) { return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); };
    // One chain step with three taps: DST[p] = op(SRC[p - D], SRC[p], SRC[p + D]) for p = pa, pb; entry p of
    // plane q lives (q * PLANE + MARGIN + p) * 8 bytes from the wave's chain base.
#define LT_STEP3(SRC, DST, D)                                                                                        \
    {                                                                                                                \
        unsigned long long a0, a1, a2, b0, b1, b2;                                                                   \
        LT_PRIO_READS2();                                                                                            \
        LT_RD64(a0, cb, ((SRC) * PLANE + MARGIN - (D)) * 8);                                                         \
        LT_RD64(a1, cb, ((SRC) * PLANE + MARGIN) * 8);                                                               \
        LT_RD64(a2, cb, ((SRC) * PLANE + MARGIN + (D)) * 8);                                                         \
        LT_RD64(b0, cb, ((SRC) * PLANE + MARGIN + 64 - (D)) * 8);                                                    \
        LT_RD64(b1, cb, ((SRC) * PLANE + MARGIN + 64) * 8);                                                          \
        LT_RD64(b2, cb, ((SRC) * PLANE + MARGIN + 64 + (D)) * 8);                                                    \
        LT_PRIO_MATH2();                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2)::"memory"); \
        (S0 + (DST) * PLANE)[pa] = op3v<DIL>(pair(a0), pair(a1), pair(a2));                                          \
        (S0 + (DST) * PLANE)[pb] = op3v<DIL>(pair(b0), pair(b1), pair(b2));                                          \
        wave_lds_fence();                                                                                            \
    }
    {
        // First chain step without LDS: the neighbours of entry p = lane (and p = lane + 64) sit in the adjacent
        // lanes' registers, one whole-wave DPP shift away; lane 63's right neighbour is entry 64 (lane 0's second
        // entry) and lane 0's left neighbour of entry 64 is entry 63 -- the rotates deliver exactly those.  Entries
        // 0 and 127 get a wrong neighbour, and are outside every window that is ever consumed.
        // no "old" operand (lanes without a source are overridden below): the move needs no copy of v into its destination first
        auto dpp = [](auto ctrl, uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, decltype(ctrl)::value, 0xf, 0xf, true); };
        using SHL = std::integral_constant<int, 0x130>; using ROL = std::integral_constant<int, 0x134>;
        using SHR = std::integral_constant<int, 0x138>; using ROR = std::integral_constant<int, 0x13c>;
        const uint2 u = make_uint2(dpp(ROL{}, e_pb.x), dpp(ROL{}, e_pb.y));      // lane i <- second entry of lane i + 1
        const uint2 v = make_uint2(dpp(ROR{}, e_pa.x), dpp(ROR{}, e_pa.y));      // lane i <- first entry of lane i - 1
#if LT_DPP_SELECT
        uint2 sl = make_uint2(dpp(SHL{}, e_pa.x), dpp(SHL{}, e_pa.y)), sr = make_uint2(dpp(SHR{}, e_pb.x), dpp(SHR{}, e_pb.y));
        // The shifts must run with every lane enabled (a DPP source lane that is masked off delivers nothing):
        // the empty statement keeps the compiler from sinking them under the lane == 63 / lane == 0 selects.
        asm volatile("" : "+v"(sl.x), "+v"(sl.y), "+v"(sr.x), "+v"(sr.y));
        const uint2 an = lane == 63 ? u : sl;
        const uint2 bp = lane == 0 ? v : sr;
        (S0 + PLANE)[pa] = op3v<DIL>(v, e_pa, an);
        (S0 + PLANE)[pb] = op3v<DIL>(bp, e_pb, u);
#else
        // The one-lane shifts take the rotated value as their "old" operand: the lane without a source (63 for the shift
        // left, 0 for the shift right) keeps it -- exactly the wrap-around entry -- so no select is needed (four
        // v_cndmask with an SGPR mask in a row are expensive, §5 issue rates).  bp is built on a copy of v (v is still
        // needed), an on u itself once u has been consumed.
        auto dpp_old = [](auto ctrl, uint32_t old, uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, decltype(ctrl)::value, 0xf, 0xf, false); };
        const uint2 bp = make_uint2(dpp_old(SHR{}, v.x, e_pb.x), dpp_old(SHR{}, v.y, e_pb.y));
        const uint2 s1b = op3v<DIL>(bp, e_pb, u);
        const uint2 an = make_uint2(dpp_old(SHL{}, u.x, e_pa.x), dpp_old(SHL{}, u.y, e_pa.y));
        (S0 + PLANE)[pa] = op3v<DIL>(v, e_pa, an);
        (S0 + PLANE)[pb] = s1b;
#endif
        wave_lds_fence();
    }
    LT_STEP3(1, 2, 3)
    const int p = SE::R + lane;
    auto lds_addr = [](const uint2* q) { return (uint32_t)(uintptr_t)q; };   // low 32 bits of a flat LDS address = LDS offset
    auto lo = [](unsigned long long v) { return (uint32_t)v; };
    auto hi = [](unsigned long long v) { return (uint32_t)(v >> 32); };
    if constexpr (SE::K == 55 && FUSE) {
        constexpr int K = SE::K, R = SE::R;
        constexpr int O4 = (2 * PLANE + MARGIN + R - 8) * 8, O13 = (3 * PLANE + MARGIN + R - 14) * 8;   // from cb
        uint32_t An[K];
        auto updates = [&](auto stage) {
            for_each_const([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (fused_stage_of_update<SE>(j) == decltype(stage)::value) {
                    An[j] = op3<DIL>(A[j + 2], Ha[SE::slot(j + 1)], Hb[SE::slot(j)]);
                    asm volatile("" : "+v"(An[j]));   // pins the update into its stage: the optimiser would sink it below every read
                }
            }, std::make_integer_sequence<int, K - 2>{});
        };
#define LT_G(i) (O13 + (i) * 8)
#define LT_PAIR(slot_, x, y) Ha[slot_] = op2<DIL>(lo(x), lo(y)); Hb[slot_] = op2<DIL>(hi(x), hi(y));
#define LT_WAIT8(n, a, b, c, d, e, f, g, h) \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "memory")
        // The reads of stage k+1 are in flight while stage k is reduced and consumed (LDS returns in order, so
        // "at most n outstanding" with n = the reads issued after a stage means that stage has arrived).
        // stage 1: half-widths 0, 7, 10, 12 -- planes S0 and S4, read under the last chain step
        unsigned long long r0, f0, f1, f2, f3, f4, f5, f6;
        LT_PRIO_READS2();
        LT_RD64(r0, s0_rd, 0);
        LT_RD64(f0, cb, O4 + 0); LT_RD64(f1, cb, O4 + 16); LT_RD64(f2, cb, O4 + 40); LT_RD64(f3, cb, O4 + 64);   // p-8, p-6, p-3, p
        LT_RD64(f4, cb, O4 + 88); LT_RD64(f5, cb, O4 + 112); LT_RD64(f6, cb, O4 + 128);                          // p+3, p+6, p+8
        LT_PRIO_MATH2();
        LT_STEP3(2, 3, 9)                                                                                        // waits for everything
        LT_WAIT8(0, r0, f0, f1, f2, f3, f4, f5, f6);
        // stage 2: 14, 16, 17, 18   (half-width 13 + q from S13[p - q], S13[p + q])
        LT_PRIO_READS();
        unsigned long long a2, b2, c2, d2, e2, f2_, g2, h2;
        LT_RD64(a2, cb, LT_G(13)); LT_RD64(b2, cb, LT_G(15)); LT_RD64(c2, cb, LT_G(11)); LT_RD64(d2, cb, LT_G(17));
        LT_RD64(e2, cb, LT_G(10)); LT_RD64(f2_, cb, LT_G(18)); LT_RD64(g2, cb, LT_G(9)); LT_RD64(h2, cb, LT_G(19));
        LT_PRIO_MATH();
        {
            Ha[0] = lo(r0); Hb[0] = hi(r0);
            LT_PAIR(1, f2, f4)
            Ha[2] = op3<DIL>(lo(f1), lo(f3), lo(f5)); Hb[2] = op3<DIL>(hi(f1), hi(f3), hi(f5));
            Ha[3] = op3<DIL>(lo(f0), lo(f3), lo(f6)); Hb[3] = op3<DIL>(hi(f0), hi(f3), hi(f6));
            out_ab[0] = op2<DIL>(A[1], Ha[SE::slot(0)]);                       // row y
            An[K - 2] = op2<DIL>(Ha[SE::slot(K - 1)], Hb[SE::slot(K - 2)]);
            An[K - 1] = Hb[SE::slot(K - 1)];
            asm volatile("" : "+v"(out_ab[0]), "+v"(An[K - 2]), "+v"(An[K - 1]));
            updates(std::integral_constant<int, 1>{});
        }
        // stage 3: 19, 20, 21, 22
        LT_PRIO_READS();
        unsigned long long a3, b3, c3, d3, e3, f3_, g3, h3;
        LT_RD64(a3, cb, LT_G(8)); LT_RD64(b3, cb, LT_G(20)); LT_RD64(c3, cb, LT_G(7)); LT_RD64(d3, cb, LT_G(21));
        LT_RD64(e3, cb, LT_G(6)); LT_RD64(f3_, cb, LT_G(22)); LT_RD64(g3, cb, LT_G(5)); LT_RD64(h3, cb, LT_G(23));
        LT_PRIO_MATH();
        LT_WAIT8(8, a2, b2, c2, d2, e2, f2_, g2, h2);
        {
            LT_PAIR(4, a2, b2) LT_PAIR(5, c2, d2) LT_PAIR(6, e2, f2_) LT_PAIR(7, g2, h2)
            updates(std::integral_constant<int, 2>{});
        }
        // stage 4: 23, 24, 25
        LT_PRIO_READS();
        unsigned long long a4, b4, c4, d4, e4, f4_;
        LT_RD64(a4, cb, LT_G(4)); LT_RD64(b4, cb, LT_G(24)); LT_RD64(c4, cb, LT_G(3)); LT_RD64(d4, cb, LT_G(25));
        LT_RD64(e4, cb, LT_G(2)); LT_RD64(f4_, cb, LT_G(26));
        LT_PRIO_MATH();
        LT_WAIT8(6, a3, b3, c3, d3, e3, f3_, g3, h3);
        {
            LT_PAIR(8, a3, b3) LT_PAIR(9, c3, d3) LT_PAIR(10, e3, f3_) LT_PAIR(11, g3, h3)
            updates(std::integral_constant<int, 3>{});
        }
        // stage 5: 26, 27 (= 13 + 14 as S13[p - 14], S13[p], S13[p + 14])
        LT_PRIO_READS();
        unsigned long long a5, b5, c5, d5, e5;
        LT_RD64(a5, cb, LT_G(1)); LT_RD64(b5, cb, LT_G(27)); LT_RD64(c5, cb, LT_G(0)); LT_RD64(d5, cb, LT_G(14)); LT_RD64(e5, cb, LT_G(28));
        LT_PRIO_MATH();
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a4), "+v"(b4), "+v"(c4), "+v"(d4), "+v"(e4), "+v"(f4_) :: "memory");
        {
            LT_PAIR(12, a4, b4) LT_PAIR(13, c4, d4) LT_PAIR(14, e4, f4_)
            updates(std::integral_constant<int, 4>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a5), "+v"(b5), "+v"(c5), "+v"(d5), "+v"(e5) :: "memory");
        {
            LT_PAIR(15, a5, b5)
            Ha[16] = op3<DIL>(lo(c5), lo(d5), lo(e5)); Hb[16] = op3<DIL>(hi(c5), hi(d5), hi(e5));
            updates(std::integral_constant<int, 5>{});
        }
#undef LT_WAIT8
#undef LT_PAIR
#undef LT_G
#pragma unroll
        for (int j = 0; j < K; ++j) A[j] = An[j];
        out_ab[1] = A[0];                                                      // row y + 1
    } else if (SE::K == 55) {
        LT_STEP3(2, 3, 9)
        const uint32_t a4 = lds_addr(S4 + p - 8), a13 = lds_addr(SL + p - 14);
        unsigned long long r0, f[7], g[29];
        LT_RD64(r0, s0_rd, 0);
        LT_RD64(f[0], a4, 0);    // p-8
        LT_RD64(f[1], a4, 16);   // p-6
        LT_RD64(f[2], a4, 40);   // p-3
        LT_RD64(f[3], a4, 64);   // p
        LT_RD64(f[4], a4, 88);   // p+3
        LT_RD64(f[5], a4, 112);  // p+6
        LT_RD64(f[6], a4, 128);  // p+8
        // g[i] = S13[p - 14 + i]; i = 1 and 27 (p -+ 13 ... ) are all used except p-+2 (i = 12, 16)
#define LT_G(i) LT_RD64(g[i], a13, (i) * 8)
        LT_G(0); LT_G(1); LT_G(2); LT_G(3); LT_G(4); LT_G(5); LT_G(6); LT_G(7); LT_G(8); LT_G(9); LT_G(10); LT_G(11);
        LT_G(13); LT_G(14); LT_G(15);
        LT_G(17); LT_G(18); LT_G(19); LT_G(20); LT_G(21); LT_G(22); LT_G(23); LT_G(24); LT_G(25); LT_G(26); LT_G(27); LT_G(28);
#undef LT_G
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(r0), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]),
                       "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),
                       "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[13]), "+v"(g[14]), "+v"(g[15])
                     :: "memory");
        asm volatile("" : "+v"(g[17]), "+v"(g[18]), "+v"(g[19]), "+v"(g[20]), "+v"(g[21]), "+v"(g[22]), "+v"(g[23]),
                          "+v"(g[24]), "+v"(g[25]), "+v"(g[26]), "+v"(g[27]), "+v"(g[28]) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        Ha[0] = lo(r0); Hb[0] = hi(r0);                                                                  // 0
        Ha[1] = op2<DIL>(lo(f[2]), lo(f[4])); Hb[1] = op2<DIL>(hi(f[2]), hi(f[4]));                      // 7
        Ha[2] = op3<DIL>(lo(f[1]), lo(f[3]), lo(f[5])); Hb[2] = op3<DIL>(hi(f[1]), hi(f[3]), hi(f[5]));  // 10
        Ha[3] = op3<DIL>(lo(f[0]), lo(f[3]), lo(f[6])); Hb[3] = op3<DIL>(hi(f[0]), hi(f[3]), hi(f[6]));  // 12
        Ha[4] = op2<DIL>(lo(g[13]), lo(g[15])); Hb[4] = op2<DIL>(hi(g[13]), hi(g[15]));                  // 14
#pragma unroll
        for (int q = 3; q <= 13; ++q) {                                                                  // 16..26
            Ha[2 + q] = op2<DIL>(lo(g[14 - q]), lo(g[14 + q]));
            Hb[2 + q] = op2<DIL>(hi(g[14 - q]), hi(g[14 + q]));
        }
        Ha[16] = op3<DIL>(lo(g[0]), lo(g[14]), lo(g[28])); Hb[16] = op3<DIL>(hi(g[0]), hi(g[14]), hi(g[28]));   // 27
    } else {
        // 29x29: every window straight from S4 (half-width 4 + t = S4[p - t] u S4[p + t] for t <= 4, with S4[p] in the
        // middle for t <= 9, and 14 = 7 u S4[p -+ 10]).  A further chain plane S7 would save nothing in VALU and costs a
        // 128-entry LDS store per row pair: the LDS pipe, where a 16-byte store takes 13 cycles, is this kernel's bound.
        const uint32_t a4 = lds_addr(S4 + p - 10);
        unsigned long long r0, g[21];
        LT_PRIO_READS2();
        LT_RD64(r0, s0_rd, 0);
#define LT_G(i) LT_RD64(g[i], a4, (i) * 8)
        LT_G(9); LT_G(11); LT_G(7); LT_G(13); LT_G(10); LT_G(5); LT_G(15); LT_G(4); LT_G(16); LT_G(3); LT_G(17); LT_G(2); LT_G(18);
        LT_G(1); LT_G(19); LT_G(0); LT_G(20);
#undef LT_G
        LT_PRIO_MATH2();
        // two waits: the first eight reads (half-widths 0 .. 9) are reduced while the other ten are still in flight (erode 0.293 ->
        // 0.289 ms, top-hat 0.295 -> 0.289 ms per 256 frames against one wait for all eighteen)
        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(r0), "+v"(g[9]), "+v"(g[11]), "+v"(g[7]), "+v"(g[13]), "+v"(g[10]), "+v"(g[5]), "+v"(g[15]) :: "memory");
        Ha[0] = lo(r0); Hb[0] = hi(r0);
        Ha[1] = op2<DIL>(lo(g[9]), lo(g[11])); Hb[1] = op2<DIL>(hi(g[9]), hi(g[11]));
        Ha[2] = op2<DIL>(lo(g[7]), lo(g[13])); Hb[2] = op2<DIL>(hi(g[7]), hi(g[13]));
        Ha[3] = op3<DIL>(lo(g[5]), lo(g[10]), lo(g[15])); Hb[3] = op3<DIL>(hi(g[5]), hi(g[10]), hi(g[15]));
        asm volatile("" : "+v"(Ha[0]), "+v"(Hb[0]), "+v"(Ha[1]), "+v"(Hb[1]), "+v"(Ha[2]), "+v"(Hb[2]), "+v"(Ha[3]), "+v"(Hb[3]));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[16]), "+v"(g[17]), "+v"(g[18]), "+v"(g[19]), "+v"(g[20])
                     :: "memory");
#pragma unroll
        for (int t = 6; t <= 9; ++t) {
            Ha[t - 2] = op3<DIL>(lo(g[10 - t]), lo(g[10]), lo(g[10 + t]));
            Hb[t - 2] = op3<DIL>(hi(g[10 - t]), hi(g[10]), hi(g[10 + t]));
        }
        Ha[8] = op3<DIL>(Ha[2], lo(g[0]), lo(g[20])); Hb[8] = op3<DIL>(Hb[2], hi(g[0]), hi(g[20]));

    }
#undef LT_STEP3
#undef LT_RD64
}

// WIDE (w % 4 == 0, planes 4-byte aligned): one byte store per lane and instruction is the expensive way to write a
// plane (dropping 7/8 of the byte stores: -15 % on the 29x29 kernels, while regrouping the byte LOADS the same way
// gained nothing).  The wave's 2 x 128 output bytes of a row pair are regrouped through 256 bytes of LDS (two 16-bit
// writes, one 64-bit read per lane) so that every lane holds four adjacent pixels of one row: ONE dword store and, for
// the top-hat, ONE dword load of the minuend per row pair instead of 4 + 4.  The read-back and the store of a row
// pair happen one iteration later, under the next pair's window update, so the LDS round trip is off the critical path.
//
// Borders without masks.  OpenCV ignores taps outside the image; here rows and columns are CLAMPED instead, which is the
// same thing for this structuring element: a clamped tap repeats an edge pixel that the true footprint already holds
// (a horizontal window that sticks out on the left contains column 0 because its centre is inside the image; an SE row
// that lies above the image repeats row 0 with ITS half-width, and the SE row that really lands on row 0 is nearer to the
// centre, i.e. at least as wide).  So no neutral fill, no per-lane keep masks and no row-inside-image selects.
// Outputs and the minuend go through running per-lane offsets into buffer descriptors that cover exactly the band's rows
// (destination) or the plane (minuend): rows outside them, and lanes right of the image, are out of range, which the
// hardware turns into dropped stores / zero loads -- no compares, no exec masking, no address clamps in the loop.
// TOPHAT: src - open(src) can never be negative (an opening is anti-extensive, also with ignored borders), so the
// saturating subtract of four bytes is one plain 32-bit subtract (no byte ever borrows).
//
// PAIR strip.  A wave's 64 lanes x 2 halves cover 128 columns, and the width is rarely a multiple of that: at 1080 columns the
// ninth strip has 56 -- its high halves and 8 lanes idle, 6.25 % of every launch.  When the last strip is at most 64
// columns wide it is therefore processed for TWO frames at once: the low halves carry frame f, the high halves frame
// f + 1, both at column x0 + lane (the chain shifts both halves alike, so nothing else changes: entry e = columns
// (x0 - R + e) of the two frames).  Input: four dwords per row instead of three (columns a, b of both frames); output:
// lanes 0..15 of each half-wave store frame f, lanes 16..31 frame f + 1 (explicit row predicate instead of the band
// descriptor, which cannot clip two frames).  An odd last frame is processed against itself and not stored twice.
// COPYM (top-hat, WIDE): the minuend dword every lane holds for its store is stored a second time into g.copy_dst, a plane
// with the destination's pitch -- the raw Lab-b plane in the layout the threshold walks read (the greenery mask of
// filter_lane_points, lane_tracker.py:224, is a bilateral threshold of the RAW plane).  One more store per row pair in a
// kernel whose memory pipe idles.
template <class SE, bool DIL, bool WIDE, bool TH, bool PAIR, bool COPYM = false>
__device__ __forceinline__ void morph_task(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint8_t* __restrict__ minuend,
                                           const RunsGeom& g, uint2* chain, uint8_t* s_out_w, int lane, int strip, int band, int frame) {
    static_assert(WIDE || !PAIR, "pair strips exist for the WIDE kernels only");
    static_assert(!COPYM || (WIDE && TH), "the minuend copy rides on the dword store of the WIDE top-hat");
    constexpr int K = SE::K, R = SE::R, NH = SE::NH;
    constexpr uint32_t NEUTRAL = (DIL ? 0u : 0x00ff00ffu) | BIAS2;
    const uint8_t* s = src + (size_t)frame * g.plane_stride;
    uint8_t* d = dst + (size_t)frame * g.dst_stride;
    const uint8_t* m = TH ? minuend + (size_t)frame * g.plane_stride : nullptr;
    const bool has_b = PAIR && frame + 1 < g.nframes;                                    // wave-uniform
    const bool lane_b = PAIR && (lane & 16);                                             // this lane stores frame f + 1
    const int x0 = strip * 128;
    const int yb0 = band * g.band_rows, yb1 = min(yb0 + g.band_rows, g.h);
    const int xa = x0 + lane, xb = xa + 64;
    const bool va = xa < g.w, vb = xb < g.w;
    const int oa = min(xa, g.w - 1), ob = min(xb, g.w - 1);

    const int wcol = x0 + 4 * (lane & (PAIR ? 15 : 31)), wcol_c = min(wcol, g.w - 4);   // WIDE: this lane's four output columns
    const uint32_t row_sel = lane < 32 ? 0x06040200u : 0x07050301u;                     // bytes of row 0 / row 1 of a [2*col + row] group

    uint32_t A[K];
#pragma unroll
    for (int j = 0; j < K; ++j) A[j] = NEUTRAL;   // flushed out of the pipeline before the band's first row

    const int y_first = yb0 - R, y_last = yb1 - 1 + R;
    // the three source columns of a lane (entry 0 = columns (a, b), entry 1 = (b, c)), clamped into the image
    const int c0 = x0 - R + lane;
    const uint32_t col_a = (uint32_t)min(max(c0, 0), g.w - 1), col_b = (uint32_t)min(max(c0 + 64, 0), g.w - 1), col_c = (uint32_t)min(max(c0 + 128, 0), g.w - 1);
    constexpr int RSRC_RAW = 0x00027000;   // untyped 32-bit buffer, no swizzle; out-of-range loads return 0, stores are dropped
    const int plane_bytes = g.h * g.w;
    const int b_src = has_b ? (int)g.plane_stride : 0;   // byte offset of frame f + 1 in the source / minuend (0: the odd last frame against itself)
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(s), 0, plane_bytes + b_src, RSRC_RAW);
    // the destination descriptor covers the band's rows only: whatever a lane computes above or below them is dropped
    // (PAIR: the band of frame f through the band of frame f + 1; rows are tested explicitly there)
    const __amdgpu_buffer_rsrc_t dst_rs = __builtin_amdgcn_make_buffer_rsrc(d + (size_t)yb0 * g.dpitch, 0, (yb1 - yb0) * g.dpitch + (has_b ? (int)g.dst_stride : 0), RSRC_RAW);
    const __amdgpu_buffer_rsrc_t min_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(TH ? m : s), 0, plane_bytes + b_src, RSRC_RAW);
    const __amdgpu_buffer_rsrc_t cp_rs = __builtin_amdgcn_make_buffer_rsrc(
        (COPYM ? g.copy_dst + (size_t)frame * g.dst_stride : d) + (size_t)yb0 * g.dpitch, 0, (yb1 - yb0) * g.dpitch + (has_b ? (int)g.dst_stride : 0), RSRC_RAW);
    auto row_ptr = [&](int y) { return __mul24(min(max(y, 0), g.h - 1), g.w); };   // byte offset of the (clamped) row: wave-uniform
    // Software prefetch: the three pixels (a, b, c) of each row of the NEXT pair are requested at the top of a pair, and
    // combined into its two entries (a | b << 16, b | c << 16) by the LAST statements of the pair.  The combine is a
    // volatile statement: left to the scheduler it is hoisted up to the loads and every pair waits out the memory latency
    // right there.  WIDE (rows are 4-byte aligned): a pixel arrives as the aligned dword that holds it, and ONE v_perm_b32
    // per entry, with a per-lane selector, picks the two pixels into the 16-bit halves and zeroes the rest.  Byte loads
    // cost more: the compiler carries their results as i8 and re-extends each with a v_and before any 32-bit use, and
    // (x << 16) | y becomes a shift and an SDWA or.
    struct Raw { uint32_t a, b, c, d; };   // PAIR: a, b of frame f and c, d = a, b of frame f + 1
    // v_perm_b32 D, S0, S1, sel: selector byte 0..3 = that byte of S1, 4..7 = byte of S0, 0x0c = constant 0
    const uint32_t sel_ab = 0x0c000c00u | ((4u + (col_b & 3u)) << 16) | (col_a & 3u);
    const uint32_t sel_bc = 0x0c000c00u | ((4u + (col_c & 3u)) << 16) | (col_b & 3u);
    const uint32_t sel_pa = 0x0c000c00u | ((4u + (col_a & 3u)) << 16) | (col_a & 3u);
    const uint32_t sel_pb = 0x0c000c00u | ((4u + (col_b & 3u)) << 16) | (col_b & 3u);
    auto load_row = [&](int y) __attribute__((always_inline)) {
        const int ro = row_ptr(y);
        Raw r;
        r.d = 0;
        if (PAIR) {
            r.a = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_a & ~3u), ro, 0);
            r.b = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_b & ~3u), ro, 0);
            r.c = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_a & ~3u), ro + b_src, 0);
            r.d = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_b & ~3u), ro + b_src, 0);
        } else if (WIDE) {
            r.a = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_a & ~3u), ro, 0);
            r.b = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_b & ~3u), ro, 0);
            r.c = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_c & ~3u), ro, 0);
        } else {
            r.a = __builtin_amdgcn_raw_buffer_load_b8(src_rs, (int)col_a, ro, 0);
            r.b = __builtin_amdgcn_raw_buffer_load_b8(src_rs, (int)col_b, ro, 0);
            r.c = __builtin_amdgcn_raw_buffer_load_b8(src_rs, (int)col_c, ro, 0);
        }
        return r;
    };
    auto combine = [&](const Raw& r, uint32_t& e0, uint32_t& e1) __attribute__((always_inline)) {
        if (PAIR) {   // entry e = (frame f, frame f + 1) at one column: the same byte of the two frames' dwords
            asm volatile("v_perm_b32 %0, %4, %2, %6\n\tv_perm_b32 %1, %5, %3, %7" : "=&v"(e0), "=&v"(e1) : "v"(r.a), "v"(r.b), "v"(r.c), "v"(r.d), "v"(sel_pa), "v"(sel_pb));
        } else if (WIDE) {
            asm volatile("v_perm_b32 %0, %3, %2, %5\n\tv_perm_b32 %1, %4, %3, %6" : "=&v"(e0), "=&v"(e1) : "v"(r.a), "v"(r.b), "v"(r.c), "v"(sel_ab), "v"(sel_bc));
        } else {
            e0 = r.a | (r.b << 16);
            e1 = r.b | (r.c << 16);
            asm volatile("" : "+v"(e0), "+v"(e1));
        }
        if (LT_MORPH_BIAS) { e0 |= BIAS2; e1 |= BIAS2; }
    };
    uint32_t ea0, ea1, eb0, eb1;   // the entries of the next row pair
    combine(load_row(y_first), ea0, ea1);
    combine(load_row(y_first + 1), eb0, eb1);
    uint32_t ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;   // !WIDE: minuend (xa, xb) of output rows y and y+1
    // WIDE: running byte offsets of this lane's dword -- row y + (lane >> 5) of the minuend, row y - 2 + (lane >> 5) of the
    // band (relative to its first row), for the output row y of the current iteration; lanes right of the image stay out of range
    uint32_t m_off = (uint32_t)(__mul24(yb0 - 2 * R + (lane >> 5), g.w) + wcol_c + (lane_b ? b_src : 0));
    uint32_t st_off = wcol < g.w && (!lane_b || has_b) ? (uint32_t)(__mul24(-2 * R - 2 + (lane >> 5), g.dpitch) + wcol + (lane_b ? (int)g.dst_stride : 0))
                                                      : 0x80000000u;
    uint32_t mcur = 0;
    const uint32_t s0_rd = (uint32_t)(uintptr_t)(chain + MARGIN + ((R + lane) & 63));
    const uint32_t out_wr = (uint32_t)(uintptr_t)(s_out_w + (WIDE ? 2 * lane : 0));         // LDS offsets
    const uint32_t out_rd = (uint32_t)(uintptr_t)(s_out_w + (WIDE ? 8 * (lane & 31) : 0));
    auto out_issue = [&](unsigned long long& q) { asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(out_rd) : "memory"); };
    auto out_finish = [&](unsigned long long q, uint32_t mp, int yp) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q) :: "memory");
        uint32_t v = __builtin_amdgcn_perm((uint32_t)(q >> 32), (uint32_t)q, row_sel);   // this lane's row of 4 columns x 2 rows
        if (TH) v = mp - v;   // TOPHAT: src - open(src) >= 0 in every byte
        if (PAIR) {           // two frames behind one descriptor: the band is tested per lane
            const int row = yp + (lane >> 5);
            if (row >= yb0 && row < yb1) {
                __builtin_amdgcn_raw_buffer_store_b32(v, dst_rs, (int)st_off, 0, 0);
                if (COPYM) __builtin_amdgcn_raw_buffer_store_b32(mp, cp_rs, (int)st_off, 0, 0);
            }
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(v, dst_rs, (int)st_off, 0, 0);
            if (COPYM) __builtin_amdgcn_raw_buffer_store_b32(mp, cp_rs, (int)st_off, 0, 0);
        }
    };
    auto row_pair = [&](int yy) __attribute__((always_inline)) {
        const uint2 e_pa = make_uint2(ea0, eb0);
        const uint2 e_pb = make_uint2(ea1, eb1);
        chain[MARGIN + lane] = lane < R ? e_pb : e_pa;   // entry (lane < R ? lane + 64 : lane): what lane (lane - R) mod 64 owns
        wave_lds_fence();
        const Raw ra = load_row(yy + 2), rb = load_row(yy + 3);
        const int y = yy - R;                       // output rows y and y+1 complete in this iteration
        const uint32_t ca0 = ma0, cb0 = mb0, ca1 = ma1, cb1 = mb1;
        const uint32_t mprev = mcur;    // WIDE: minuend of rows y-2, y-1 (stored in this iteration)
        if (TH) {
            if (WIDE) {   // lane <-> row y + (lane >> 5), columns x0 + 4 (lane & 31) .. + 3: the layout of the dword store
                mcur = __builtin_amdgcn_raw_buffer_load_b32(min_rs, (int)m_off, 0, 0);
            } else {
                const uint8_t* r0 = m + (size_t)min(max(y + 2, 0), g.h - 1) * g.w;
                const uint8_t* r1 = m + (size_t)min(max(y + 3, 0), g.h - 1) * g.w;
                ma0 = r0[oa]; mb0 = r0[ob];
                ma1 = r1[oa]; mb1 = r1[ob];
            }
        }
        uint32_t Ha[NH], Hb[NH];
        constexpr bool FUSE = LT_FUSED55 && K == 55;
        uint32_t out_ab[2];
        row_windows2<SE, DIL, FUSE>(chain, lane, e_pa, e_pb, s0_rd, Ha, Hb, A, out_ab);
        wave_lds_fence();   // the chain planes are rewritten by the next iteration
        const bool prev_out = WIDE && y - 1 >= yb0 && y - 2 < yb1;   // rows y-2, y-1 wait regrouped in s_out
        unsigned long long q = 0;
        if (prev_out) out_issue(q);
        if (WIDE) __builtin_amdgcn_sched_barrier(0);
        if (!FUSE) {
            out_ab[0] = op2<DIL>(A[1], Ha[SE::slot(0)]);                              // row y
#pragma unroll
            for (int j = 0; j < K - 2; ++j) A[j] = op3<DIL>(A[j + 2], Ha[SE::slot(j + 1)], Hb[SE::slot(j)]);
            A[K - 2] = op2<DIL>(Ha[SE::slot(K - 1)], Hb[SE::slot(K - 2)]);
            A[K - 1] = Hb[SE::slot(K - 1)];
            out_ab[1] = A[0];                                                         // row y + 1
        }
        const uint32_t out_a = out_ab[0], out_b = out_ab[1];
        if (WIDE) {
            __builtin_amdgcn_sched_barrier(0);   // the window update above stays between the read-back and its use
            // unconditional: outside the band the store falls outside dst_rs.  Under a branch the compiler cannot count
            // it, and the wait for the next pair's bytes at the loop top becomes vmcnt(0) -- a wait for this store
            out_finish(q, mprev, y - 2);
            if (y + 1 >= yb0 && y < yb1) {       // wave-uniform; read back and stored while the next row pair computes
                const uint32_t W = __builtin_amdgcn_perm(out_b, out_a, 0x06020400u);   // [row y: xa, row y+1: xa, y: xb, y+1: xb]
                asm volatile("ds_write_b16 %0, %1\n\tds_write_b16_d16_hi %0, %1 offset:128" :: "v"(out_wr), "v"(W) : "memory");
            }
            m_off += 2u * (uint32_t)g.w;
            st_off += 2u * (uint32_t)g.dpitch;
        } else
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int yo = y + rr;
            if (yo >= yb0 && yo < yb1) {
                const uint32_t o2 = rr == 0 ? out_a : out_b;
                uint32_t oa_v = o2 & 0xffu, ob_v = (o2 >> 16) & 0xffu;
                if (TH) {   // TOPHAT: src - open(src)
                    const uint32_t qa = rr == 0 ? ca0 : ca1, qb = rr == 0 ? cb0 : cb1;
                    oa_v = qa - oa_v;
                    ob_v = qb - ob_v;
                }
                const size_t o = (size_t)yo * g.dpitch;
                if (va) d[o + xa] = (uint8_t)oa_v;
                if (vb) d[o + xb] = (uint8_t)ob_v;
            }
        }
        combine(ra, ea0, ea1);
        combine(rb, eb0, eb1);
    };
    // 55x55: two row pairs per loop iteration.  The vertical pipeline A moves by two registers per row pair; with one
    // pair per iteration the staged update order (outer rows first) left 11 register copies at the loop's back edge,
    // with two the second pair lands in the first one's registers.  An odd pair count runs one pair past the band
    // (loads are clamped, stores fall outside the band's descriptor).
    constexpr int PAIRS = (LT_FUSED55 && K == 55) ? 2 : 1;
    for (int yy = y_first; yy <= y_last; yy += 2 * PAIRS) {
        row_pair(yy);
        if (PAIRS == 2) row_pair(yy + 2);
    }
    if (WIDE) {   // the last row pair is still in s_out (outside the band if the loop ran past it: dropped by the descriptor)
        const int npairs = (y_last - y_first) / 2 + 1, pairs_run = (npairs + PAIRS - 1) / PAIRS * PAIRS;
        const int y_tail = y_first + 2 * (pairs_run - 1) - R;     // output row y of the last pair the loop ran
        unsigned long long q;
        out_issue(q);
        out_finish(q, mcur, y_tail);
    }
}

// Waves per workgroup.  The waves of a workgroup share nothing (no barrier, separate LDS slices); the workgroup is only the
// unit the dispatcher places.  1, 2 and 4 measure the same (-DLT_MORPH_WPB=n).
#ifndef LT_MORPH_WPB
#define LT_MORPH_WPB 4
#endif
// -DLT_MORPH_MAX_WAVES=n: cap the kernel at n waves per SIMD (co-residency experiments: the registers of the waves it does
// not take stay free for another stream's kernels)
#ifdef LT_MORPH_MAX_WAVES
#define LT_MORPH_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(1, LT_MORPH_MAX_WAVES)))
#else
#define LT_MORPH_WAVES_ATTR
#endif
template <class SE, bool DIL, bool WIDE, bool TH, bool COPYM = false>
__global__ __launch_bounds__(64 * LT_MORPH_WPB) LT_MORPH_WAVES_ATTR void k_morph_runs2(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                    const uint8_t* __restrict__ minuend, RunsGeom g) {
    __shared__ uint2 s_chain[LT_MORPH_WPB][4 * PLANE];   // S0, S1, S4, S13
    __shared__ __attribute__((aligned(8))) uint8_t s_out[WIDE ? LT_MORPH_WPB : 1][WIDE ? 256 : 8];   // [2*col + row] of a row pair
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8) and every XCD has its own L2.  Tasks are numbered
    // strip-fastest, so in launch order the two halves of a band's strips, and the bands above and below, run on eight different
    // L2s and each fetches the shared halo (2R columns / rows) from HBM on its own.  Renumbered, an XCD owns one contiguous
    // range of tasks: neighbours in the image are neighbours in time on one L2.  Bijective for any grid; speed only.
    uint32_t bid = blockIdx.x;
    if (g.xcd) {
        const uint32_t total = gridDim.x, q = total >> 3, r = total & 7u, xcd = bid & 7u, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int task = __builtin_amdgcn_readfirstlane((int)bid * LT_MORPH_WPB + wv);   // wave-uniform: keeps the loop scalar
    if (task >= g.ntasks) return;
    // Without the bias the pixel patterns 0x00vv are f16 denormals: min/max must not flush them.  FP16 denormals are on
    // in the kernel descriptor the compiler writes (tests/test_isa_guards.py checks it); set MODE.FP_DENORM[3:2] anyway.
    if (!LT_MORPH_BIAS) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3);
    // pair tasks first: they are as long as any other task, and at the end of the launch they would run on a half-empty chip
    const int n_pair = g.ntasks - g.n_normal;
    if (!WIDE || task >= n_pair) {
        const int t = task - n_pair;
        const int strip = t % g.nstrips_normal;
        const int band = (t / g.nstrips_normal) % g.nbands;
        const int frame = t / (g.nstrips_normal * g.nbands);
        morph_task<SE, DIL, WIDE, TH, false, COPYM>(src, dst, minuend, g, s_chain[wv], s_out[wv], lane, strip, band, frame);
    } else if constexpr (WIDE) {
        morph_task<SE, DIL, WIDE, TH, true, COPYM>(src, dst, minuend, g, s_chain[wv], s_out[wv], lane, g.nstrips_normal, task % g.nbands, 2 * (task / g.nbands));
    }
}

// ================================================================================================
// One or two frames per launch (process(): one frame per call, the host waiting for its record).
//
// k_morph_runs2 at one frame is a few hundred waves, each alone on its SIMD, walking band_rows + 2R rows one dependent row
// pair after the other (0.85 us per pair at one wave per SIMD: profiles/NOTES_r05.md D.4) -- and the walk cannot be shorter
// than the halo, 2R rows, however short the band.  Here the walk of a (strip, band) task is split over the Q waves of a
// workgroup BY ROW PAIRS: wave q takes the pairs q, q + Q, q + 2Q, ... of the task's rows.  Each wave runs the horizontal part
// of its own pairs only (nothing is computed twice) and keeps a vertical pipeline of its own, which now moves by 2Q rows per
// step instead of two:
//     A'[j] = op(A[j + 2Q], Ha[slot(j + 1)], Hb[slot(j)]),     j = -1 .. K - 1
// so that after a step the wave's positions -1 .. 2Q - 2 are final AS FAR AS ITS OWN ROWS GO: partial results of 2Q output
// rows.  The min / max over the Q waves' partials is the output row.  Partials travel through a ring of rows in LDS (one
// dword per lane, row and wave); after the step's barrier the rows all Q waves have delivered are combined, two rows per wave,
// and stored.  A task's dependent chain is (band_rows + 2R) / 2Q steps instead of (band_rows + 2R) / 2.
//
// Which rows are complete after step t.  Offsets o count rows from the first output row any wave can complete,
// yb0 - 2R.  Wave q's pair of step t is rows y_first + 2 (t Q + q) (+1), and it delivers the offsets 2Q t + 2q .. 2Q t + 2q + 2Q - 1.
// So once every wave has finished step t the offsets 2Q t .. 2Q t + 2Q - 1 are there from all of them (the first 2q of
// them from wave q's step t - 1).  Offsets below 2R lie above the band and are never stored, which covers t = 0.  The ring
// holds 3 x 2Q rows: a wave that is already writing step t + 1 (offsets < 2Q (t + 3) - 2) while another one still combines step t
// touches none of its rows, and nobody gets further ahead than that (one barrier per step).
//
// Borders are clamped as in k_morph_runs2; the argument there needs only that an SE row above the image is narrower than the
// one that lands on row 0, which does not depend on who accumulates which rows.
template <class SE, bool DIL, bool TH, int Q>
__device__ __forceinline__ void morph_one_task(const int task, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                               const uint8_t* __restrict__ minuend, const RunsGeom& g, uint2 (*s_chain)[4 * PLANE],
                                               uint32_t (*s_ring)[3 * 2 * Q][64]) {
    constexpr int K = SE::K, R = SE::R, NH = SE::NH, S = 2 * Q, RING = 3 * S;
    static_assert(S - 1 <= K - 1, "the pipeline must be longer than a step");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (!LT_MORPH_BIAS) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3);   // f16 denormals kept (see k_morph_runs2)
    const int strip = task % g.nstrips;
    const int band = (task / g.nstrips) % g.nbands;
    const int frame = task / (g.nstrips * g.nbands);
    const uint8_t* s = src + (size_t)frame * g.plane_stride;
    uint8_t* d = dst + (size_t)frame * g.dst_stride;
    const uint8_t* m = TH ? minuend + (size_t)frame * g.plane_stride : nullptr;
    uint2* chain = s_chain[wv];
    const int x0 = strip * 128;
    const int yb0 = band * g.band_rows, yb1 = min(yb0 + g.band_rows, g.h);
    const int xa = x0 + lane, xb = xa + 64;
    const bool va = xa < g.w, vb = xb < g.w;
    const int oa = min(xa, g.w - 1), ob = min(xb, g.w - 1);
    const int y_first = yb0 - R, y_base = yb0 - 2 * R;
    const int npairs = (yb1 - yb0 + 2 * R + 1) / 2, nsteps = (npairs + Q - 1) / Q;
    // the three source columns of a lane (entry 0 = columns (a, b), entry 1 = (b, c)), clamped into the image; rows are 4-byte
    // aligned (the launcher checks): a pixel arrives as the aligned dword that holds it and one v_perm_b32 per entry picks the two
    // pixels into the 16-bit halves (as in morph_task)
    const int c0 = x0 - R + lane;
    const uint32_t col_a = (uint32_t)min(max(c0, 0), g.w - 1), col_b = (uint32_t)min(max(c0 + 64, 0), g.w - 1), col_c = (uint32_t)min(max(c0 + 128, 0), g.w - 1);
    constexpr int RSRC_RAW = 0x00027000;
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(s), 0, g.h * g.w, RSRC_RAW);
    auto row_off = [&](int y) { return __mul24(min(max(y, 0), g.h - 1), g.w); };
    const uint32_t sel_ab = 0x0c000c00u | ((4u + (col_b & 3u)) << 16) | (col_a & 3u);
    const uint32_t sel_bc = 0x0c000c00u | ((4u + (col_c & 3u)) << 16) | (col_b & 3u);
    struct Raw { uint32_t a, b, c; };
    auto load_row = [&](int y) __attribute__((always_inline)) {
        const int ro = row_off(y);
        Raw r;
        r.a = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_a & ~3u), ro, 0);
        r.b = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_b & ~3u), ro, 0);
        r.c = __builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(col_c & ~3u), ro, 0);
        return r;
    };
    auto combine = [&](const Raw& r, uint32_t& e0, uint32_t& e1) __attribute__((always_inline)) {
        asm volatile("v_perm_b32 %0, %3, %2, %5\n\tv_perm_b32 %1, %4, %3, %6" : "=&v"(e0), "=&v"(e1) : "v"(r.a), "v"(r.b), "v"(r.c), "v"(sel_ab), "v"(sel_bc));
        if (LT_MORPH_BIAS) { e0 |= BIAS2; e1 |= BIAS2; }
    };
    constexpr uint32_t NEUTRAL = (DIL ? 0u : 0x00ff00ffu) | BIAS2;
    uint32_t A[K + 1];                       // position j = -1 .. K - 1 at index j + 1
#pragma unroll
    for (int p = 0; p <= K; ++p) A[p] = NEUTRAL;
    uint32_t ea0, ea1, eb0, eb1;             // the entries of this wave's next row pair
    {
        const int yy = y_first + 2 * wv;
        combine(load_row(yy), ea0, ea1);
        combine(load_row(yy + 1), eb0, eb1);
    }
    const uint32_t s0_rd = (uint32_t)(uintptr_t)(chain + MARGIN + ((R + lane) & 63));
    for (int t = 0; t < nsteps; ++t) {
        const int yy = y_first + 2 * (t * Q + wv);            // this wave's pair of the step: rows yy, yy + 1
        const uint2 e_pa = make_uint2(ea0, eb0);
        const uint2 e_pb = make_uint2(ea1, eb1);
        chain[MARGIN + lane] = lane < R ? e_pb : e_pa;        // plane 0 holds the 64 entries the half-width-0 window reads (row_windows2)
        wave_lds_fence();
        const Raw ra = load_row(yy + S), rb = load_row(yy + S + 1);
        // the two output rows this wave combines and stores after the barrier, and their minuend
        const int o0 = S * t + 2 * wv, yo = y_base + o0;
        uint32_t m0a = 0, m0b = 0, m1a = 0, m1b = 0;
        if (TH) {
            const uint8_t* r0 = m + (size_t)min(max(yo, 0), g.h - 1) * g.w;
            const uint8_t* r1 = m + (size_t)min(max(yo + 1, 0), g.h - 1) * g.w;
            m0a = r0[oa]; m0b = r0[ob];
            m1a = r1[oa]; m1b = r1[ob];
        }
        uint32_t Ha[NH], Hb[NH];
        row_windows2<SE, DIL, false>(chain, lane, e_pa, e_pb, s0_rd, Ha, Hb);
        wave_lds_fence();                                     // the chain planes are rewritten by the next step
        uint32_t An[K + 1];
#pragma unroll
        for (int p = 0; p <= K; ++p) {
            const int j = p - 1;
            const bool prev = j + S <= K - 1, ha = j <= K - 2, hb = j >= 0;
            // (the slot indices are clamped for the compiler's sake where the operand does not exist)
            const uint32_t va_ = Ha[SE::slot(j + 1 <= K - 1 ? j + 1 : K - 1)], vb_ = Hb[SE::slot(j >= 0 ? j : 0)];
            const uint32_t vp = A[prev ? p + S : K];
            An[p] = prev && ha && hb ? op3<DIL>(vp, va_, vb_) : prev && ha ? op2<DIL>(vp, va_) : ha && hb ? op2<DIL>(va_, vb_) : vb_;
        }
        // positions -1 .. S - 2 are this wave's share of the output rows at offsets o0 .. o0 + S - 1
        {
            int slot = o0 % RING;
#pragma unroll
            for (int q = 0; q < S; ++q) {
                s_ring[wv][slot][lane] = An[q];
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
        }
#pragma unroll
        for (int p = S; p <= K; ++p) A[p] = An[p];
        __syncthreads();
        {
            const int sl0 = o0 % RING, sl1 = sl0 + 1 == RING ? 0 : sl0 + 1;
            uint32_t v0 = s_ring[0][sl0][lane], v1 = s_ring[0][sl1][lane];
#pragma unroll
            for (int q = 1; q < Q; ++q) {
                v0 = op2<DIL>(v0, s_ring[q][sl0][lane]);
                v1 = op2<DIL>(v1, s_ring[q][sl1][lane]);
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int y = yo + rr;
                if (y >= yb0 && y < yb1) {                    // wave-uniform
                    const uint32_t v = rr == 0 ? v0 : v1;
                    uint32_t pa = v & 0xffu, pb = (v >> 16) & 0xffu;
                    if (TH) {                                 // src - open(src), never negative
                        pa = (rr == 0 ? m0a : m1a) - pa;
                        pb = (rr == 0 ? m0b : m1b) - pb;
                    }
                    const size_t o = (size_t)y * g.dpitch;
                    if (va) d[o + xa] = (uint8_t)pa;
                    if (vb) d[o + xb] = (uint8_t)pb;
                }
            }
        }
        combine(ra, ea0, ea1);
        combine(rb, eb0, eb1);
    }
}

template <class SE, bool DIL, bool TH, int Q>
__global__ __launch_bounds__(64 * Q) void k_morph_one(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      const uint8_t* __restrict__ minuend, RunsGeom g) {
    __shared__ uint2 s_chain[Q][4 * PLANE];   // S0, S1, S4, S13 per wave
    __shared__ uint32_t s_ring[Q][3 * 2 * Q][64];
    morph_one_task<SE, DIL, TH, Q>((int)blockIdx.x, src, dst, minuend, g, s_chain, s_ring);
}

// The same step of BOTH top-hats of a frame in ONE launch (process(): the 55x55 erode of the Lab-b plane and the 29x29 erode of the
// R plane; then the two top-hats): the planes do not depend on each other, one frame's tasks of either fill a fraction of the
// chip, and as two launches on two streams they cost a fork, a join (11-12 us of signalling on the frame's critical path:
// profiles/r05_process_timeline.txt) and two more launches.  The 55x55 tasks -- the longer walks -- come first in the grid.
template <bool DIL, bool TH, int Q>
__global__ __launch_bounds__(64 * Q) void k_morph_one_pair(const uint8_t* __restrict__ src55, uint8_t* __restrict__ dst55, const uint8_t* __restrict__ min55, RunsGeom g55,
                                                           const uint8_t* __restrict__ src29, uint8_t* __restrict__ dst29, const uint8_t* __restrict__ min29, RunsGeom g29) {
    __shared__ uint2 s_chain[Q][4 * PLANE];
    __shared__ uint32_t s_ring[Q][3 * 2 * Q][64];
    const int task = (int)blockIdx.x;      // (workgroup-uniform: the branch below costs nothing)
    if (task < g55.ntasks) morph_one_task<SE55, DIL, TH, Q>(task, src55, dst55, min55, g55, s_chain, s_ring);
    else morph_one_task<SE29, DIL, TH, Q>(task - g55.ntasks, src29, dst29, min29, g29, s_chain, s_ring);
}

template <class SE, int Q>
void launch_one(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, bool dilate, const RunsGeom& g) {
    const dim3 grid(g.ntasks), block(64 * Q);
    if (dilate && minuend) hipLaunchKernelGGL((k_morph_one<SE, true, true, Q>), grid, block, 0, s, src, dst, minuend, g);
    else if (dilate) hipLaunchKernelGGL((k_morph_one<SE, true, false, Q>), grid, block, 0, s, src, dst, minuend, g);
    else hipLaunchKernelGGL((k_morph_one<SE, false, false, Q>), grid, block, 0, s, src, dst, minuend, g);
}

// Bands of a one- or two-frame launch of k_morph_one: about `want_wgs` workgroups (default: two per CU), none shorter than 8 rows
static void one_frame_bands(RunsGeom& g, int n, int want_wgs) {
    int cus1 = 256, dev1 = 0;
    (void)hipGetDevice(&dev1);
    (void)hipDeviceGetAttribute(&cus1, hipDeviceAttributeMultiprocessorCount, dev1);
    const int target = want_wgs > 0 ? want_wgs : 2 * cus1;
    int nb = std::max(1, target / std::max(1, n * g.nstrips));
    int rows = std::max((g.h + nb - 1) / nb, 8);
    rows = (rows + 1) & ~1;
    g.band_rows = rows;
    g.nbands = (g.h + rows - 1) / rows;
    g.nstrips_normal = g.nstrips;
    g.ntasks = g.n_normal = n * g.nstrips * g.nbands;
    g.xcd = 0;
}

template <class SE>
bool table_matches(const EllipseSE& se) {
    if (se.k != SE::K) return false;
    for (int j = 0; j < SE::K; ++j)
        if (SE::slot_width[SE::slot(j)] != se.dx[j]) return false;
    return true;
}

template <class SE>
bool launch_runs(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w, bool dilate,
                 size_t plane_stride, int n, int dpitch, size_t dst_stride, uint8_t* copy_dst) {
    RunsGeom g;
    g.copy_dst = copy_dst;
    g.h = h;
    g.w = w;
    g.plane_stride = plane_stride;
    g.dpitch = dpitch > 0 ? dpitch : w;
    g.dst_stride = dpitch > 0 ? dst_stride : plane_stride;
    g.nstrips = (w + 127) / 128;
    g.nframes = n;
    static const bool one_row = [] { const char* e = LT_EXP_ENV("LT_MORPH_ONE_ROW"); return e && e[0] == '1'; }();
    static const bool narrow = [] { const char* e = LT_EXP_ENV("LT_MORPH_WIDE"); return e && e[0] == '0'; }();
    const bool wide = !narrow && (w & 3) == 0 && (plane_stride & 3) == 0 && w >= 4 && (g.dpitch & 3) == 0 && (g.dst_stride & 3) == 0 &&
                      ((uintptr_t)dst & 3) == 0 && ((uintptr_t)minuend & 3) == 0;
    // One or two frames: the walk of a task split over the waves of a workgroup (k_morph_one).  LT_MORPH_ONE=0: k_morph_runs2 as
    // for any other frame count (A/B, tests); =8: eight waves per task.  LT_MORPH_ONE_WGS: workgroups per launch to aim for.
    static const int one_q = [] { const char* e = LT_EXP_ENV("LT_MORPH_ONE"); return e ? std::atoi(e) : 4; }();
    if (n <= 2 && (one_q == 4 || one_q == 8) && !one_row && wide && !copy_dst && ((uintptr_t)src & 3) == 0) {
        static const int want_wgs = [] { const char* e = LT_EXP_ENV("LT_MORPH_ONE_WGS"); return e ? std::atoi(e) : 0; }();
        one_frame_bands(g, n, want_wgs);
        if (one_q == 8) launch_one<SE, 8>(s, src, dst, minuend, dilate, g);
        else launch_one<SE, 4>(s, src, dst, minuend, dilate, g);
        return true;
    }
    // Band count: every task walks band_rows + 2R rows, and the chip holds `slots` waves at once, so
    // the makespan is ~ ceil(tasks / slots) * (band_rows + 2R).  Pick the band count that minimises
    // it (a grid of 2.25 rounds costs 3 rounds); bands no shorter than 2R keep the halo overhead sane.
    int dev = 0, cus = 256, blocks_per_cu = 3;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    {
        const void* fn = one_row ? (dilate ? (const void*)k_morph_runs<SE, true> : (const void*)k_morph_runs<SE, false>)
                                 : wide ? (dilate ? (minuend ? (const void*)k_morph_runs2<SE, true, true, true> : (const void*)k_morph_runs2<SE, true, true, false>)
                                                  : (const void*)k_morph_runs2<SE, false, true, false>)
                                        : (dilate ? (minuend ? (const void*)k_morph_runs2<SE, true, false, true> : (const void*)k_morph_runs2<SE, true, false, false>)
                                                  : (const void*)k_morph_runs2<SE, false, false, false>);
        int nb = 0;
        const int threads = one_row ? 256 : 64 * LT_MORPH_WPB;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, 0) == hipSuccess && nb > 0) blocks_per_cu = nb * threads / 256;
        if (blocks_per_cu < 1) blocks_per_cu = 1;
    }
    const long long slots = (long long)cus * blocks_per_cu * 4, simds = (long long)cus * 4;
    int best_nb = 1;
    double best_cost = 1e300;
    for (int nb = 1; nb <= 256; ++nb) {
        const int rows = (h + nb - 1) / nb;
        const int real_nb = (h + rows - 1) / rows;
        const long long tasks = (long long)n * g.nstrips * real_nb;
        // full grids: bands no shorter than 2R keep the halo overhead sane; a grid that cannot fill the chip
        // anyway (a single frame) is latency-bound, so it may trade redundant halo rows for shorter walks -- down to a few
        // rows per task while every wave still has a SIMD to itself (one 1100-row frame: 15 -> 10 rows, 29.6 -> 27.2 us per
        // 55x55 launch; a second wave per SIMD costs more than the shorter walk gains: 31.5 us at 6 rows)
        if (nb > 1 && rows < (tasks <= simds ? 4 : tasks <= slots ? (SE::R + 1) / 2 : 2 * SE::R)) break;
        const double rounds = (double)((tasks + slots - 1) / slots);
        // small grids cannot fill the chip: prefer more, shorter tasks there
        const double cost = tasks <= simds ? (double)(rows + 2 * SE::R) : tasks < slots ? 1.15 * (rows + 2 * SE::R) : rounds * (rows + 2 * SE::R);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_nb = nb; }
    }
    // Measured on the full 256-frame grid (tools/nb_sweep.sh, tools/nb55_sweep.sh): all four kernels are fastest with
    // 4 bands of 275 rows (the model above picks 5 and 2 for 29x29 and 5 for 55x55: it trusts an occupancy figure
    // whose extra waves add no throughput -- the kernels are bound by VALU issue and the LDS pipe, not by latency).
    if ((long long)n * g.nstrips * 4 >= slots && h / 4 >= 2 * SE::R) best_nb = 4;
    // Launches of a slot slice (85 of 256 frames on three streams) share the chip with the other slices' kernels:
    // there, too, few long walks beat a grid sized to fill the chip on its own (tools/nb_value_sweep.sh: 3 bands
    // 66.8 k frames/s, 4 bands 66.5 k, the model's 5-6 bands 65.5 k).
    else if ((long long)n * g.nstrips * 3 >= slots / 4 && h / 3 >= 2 * SE::R) best_nb = 3;
#ifdef LT_EXPERIMENTS
    {   // measurement override: LT_MORPH_NB_<k><E|D>=<bands>, e.g. LT_MORPH_NB_55D=5
        char name[32];
        std::snprintf(name, sizeof name, "LT_MORPH_NB_%d%c", SE::K, dilate ? 'D' : 'E');
        const char* e = std::getenv(name);
        if (e && std::atoi(e) > 0) best_nb = std::atoi(e);
    }
#endif
    g.band_rows = (h + best_nb - 1) / best_nb;
    g.nbands = (h + g.band_rows - 1) / g.band_rows;
    // the last strip as a PAIR strip (two frames per wave) when it is at most 64 columns wide
    static const bool no_pair = [] { const char* e = LT_EXP_ENV("LT_MORPH_PAIR"); return e && e[0] == '0'; }();   // A/B
    const bool pair_strip = wide && !one_row && !no_pair && w - 128 * (g.nstrips - 1) <= 64 &&
                            (long long)plane_stride + (long long)h * w < (1ll << 31) && (long long)g.dst_stride + (long long)h * g.dpitch < (1ll << 31);
    g.nstrips_normal = pair_strip ? g.nstrips - 1 : g.nstrips;
    g.n_normal = n * g.nstrips_normal * g.nbands;
    g.ntasks = g.n_normal + (pair_strip ? (n + 1) / 2 * g.nbands : 0);
    // measured (tools/ab_fetch.sh LT_MORPH_XCD): the renumbering makes all four launches 1-5 % SLOWER (0.295 -> 0.311 ms
    // erode 29x29, 0.493 -> 0.508 ms erode 55x55 per 256 frames), so it is off unless asked for
    static const bool want_xcd = [] { const char* e = LT_EXP_ENV("LT_MORPH_XCD"); return e && e[0] == '1'; }();
    g.xcd = want_xcd ? 1 : 0;
    dim3 grid((g.ntasks + 3) / 4);
    const dim3 grid2((g.ntasks + LT_MORPH_WPB - 1) / LT_MORPH_WPB), block2(64 * LT_MORPH_WPB);
    if (copy_dst && one_row) return false;
    if (one_row) {   // previous formulation (one row per iteration, u16 min/max), kept for A/B measurements
        if (dilate) hipLaunchKernelGGL((k_morph_runs<SE, true>), grid, dim3(256), 0, s, src, dst, minuend, g);
        else hipLaunchKernelGGL((k_morph_runs<SE, false>), grid, dim3(256), 0, s, src, dst, minuend, g);
    } else {
        static const int extra_lds = [] { const char* e = LT_EXP_ENV("LT_MORPH_EXTRA_LDS"); return e ? std::atoi(e) : 0; }();   // occupancy experiments
        const bool th = dilate && minuend != nullptr;
        if (copy_dst) {   // the minuend copy exists for the 55x55 WIDE top-hat only (the Lab-b plane)
            if constexpr (SE::K == 55) {
                if (!wide || !th || ((uintptr_t)copy_dst & 3)) return false;
                hipLaunchKernelGGL((k_morph_runs2<SE, true, true, true, true>), grid2, block2, extra_lds, s, src, dst, minuend, g);
                return true;
            } else
                return false;
        }
#define LT_LAUNCH(DIL_, WIDE_, TH_) hipLaunchKernelGGL((k_morph_runs2<SE, DIL_, WIDE_, TH_>), grid2, block2, extra_lds, s, src, dst, minuend, g)
        if (wide) {
            if (th) LT_LAUNCH(true, true, true);
            else if (dilate) LT_LAUNCH(true, true, false);
            else LT_LAUNCH(false, true, false);
        } else {
            if (th) LT_LAUNCH(true, false, true);
            else if (dilate) LT_LAUNCH(true, false, false);
            else LT_LAUNCH(false, false, false);
        }
#undef LT_LAUNCH
    }
    return copy_dst == nullptr;
}

}  // namespace

// true if the compiled-in run tables equal the structuring element OpenCV's formula gives
bool tophat_tables_match(const EllipseSE& se29, const EllipseSE& se55) {
    return table_matches<SE29>(se29) && table_matches<SE55>(se55);
}

bool launch_morph_runs(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w, int k,
                       bool dilate, size_t plane_stride, int n, int dpitch, size_t dst_stride, uint8_t* copy_dst) {
    if (n <= 0 || h <= 0 || w <= 0) return true;
    if (k == 55)
        return launch_runs<SE55>(s, src, dst, minuend, h, w, dilate, plane_stride, n, dpitch, dst_stride, copy_dst);
    return launch_runs<SE29>(s, src, dst, minuend, h, w, dilate, plane_stride, n, dpitch, dst_stride, copy_dst);
}

// One or two frames: the same step (erode, or dilate / top-hat) of the 55x55 chain of one plane and of the 29x29 chain of another
// in ONE launch (k_morph_one_pair).  false: not launched (a geometry k_morph_one does not take) -- the caller launches them one by one.
bool launch_morph_one_pair(hipStream_t s, const uint8_t* src55, uint8_t* dst55, const uint8_t* min55, const uint8_t* src29, uint8_t* dst29,
                           const uint8_t* min29, int h, int w, bool dilate, size_t plane_stride, int n, int dpitch, size_t dst_stride) {
    if (n <= 0 || n > 2 || h <= 0 || w < 4) return false;
    RunsGeom g;
    g.copy_dst = nullptr;
    g.h = h;
    g.w = w;
    g.plane_stride = plane_stride;
    g.dpitch = dpitch > 0 ? dpitch : w;
    g.dst_stride = dpitch > 0 ? dst_stride : plane_stride;
    g.nstrips = (w + 127) / 128;
    g.nframes = n;
    auto aligned = [](const void* q) { return ((uintptr_t)q & 3) == 0; };
    if ((w & 3) || (plane_stride & 3) || (g.dpitch & 3) || (g.dst_stride & 3) || !aligned(src55) || !aligned(dst55) || !aligned(min55) ||
        !aligned(src29) || !aligned(dst29) || !aligned(min29) || (dilate && (min55 == nullptr) != (min29 == nullptr)))
        return false;
    RunsGeom g55 = g, g29 = g;
    // workgroups per plane (default: two per CU each) and waves per task: measurement switches of the experiments build
    static const int wgs55 = [] { const char* e = LT_EXP_ENV("LT_PAIR_WGS55"); return e ? std::atoi(e) : 0; }();
    static const int wgs29 = [] { const char* e = LT_EXP_ENV("LT_PAIR_WGS29"); return e ? std::atoi(e) : 0; }();
    static const int pair_q = [] { const char* e = LT_EXP_ENV("LT_PAIR_Q"); return e ? std::atoi(e) : 4; }();
    // Measured (tools/pair_sweep.sh, wall time of one frame's mask chain + band search + record, three runs each): two
    // workgroups per CU for each plane -- what a launch of one plane alone takes -- 92.9-94.6 us; one per CU for the 55x55 plane
    // and one and a half for the 29x29 plane 82.9-83.9 us (256 / 256: 82.9-85.3; 192 / any: 87-93; 384 / any: 88-92).
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    one_frame_bands(g55, n, wgs55 > 0 ? wgs55 : cus);
    one_frame_bands(g29, n, wgs29 > 0 ? wgs29 : cus + cus / 2);
    const dim3 grid(g55.ntasks + g29.ntasks);
#define LT_PAIR(Q_)                                                                                                                              \
    do {                                                                                                                                         \
        const dim3 block(64 * Q_);                                                                                                               \
        if (dilate && min55) hipLaunchKernelGGL((k_morph_one_pair<true, true, Q_>), grid, block, 0, s, src55, dst55, min55, g55, src29, dst29, min29, g29);   \
        else if (dilate) hipLaunchKernelGGL((k_morph_one_pair<true, false, Q_>), grid, block, 0, s, src55, dst55, min55, g55, src29, dst29, min29, g29);      \
        else hipLaunchKernelGGL((k_morph_one_pair<false, false, Q_>), grid, block, 0, s, src55, dst55, min55, g55, src29, dst29, min29, g29);                 \
    } while (0)
#ifdef LT_EXPERIMENTS
    if (pair_q == 8) { LT_PAIR(8); return true; }
#endif
    (void)pair_q;
    LT_PAIR(4);
#undef LT_PAIR
    return true;
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_tophat() {} }
void preload_k_tophat(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_tophat, dim3(1), dim3(1), 0, s); }

}  // namespace lt

