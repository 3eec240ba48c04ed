// Geometric front end + colour split, gfx950.
//
//   k_undistort_rows   cv2.undistort            lane_tracker.py:832   (only the rows the warp reads)
//   k_warp_split       cv2.warpPerspective      lane_tracker.py:834
//                      + img[:,:,0]             lane_tracker.py:207
//                      + cvtColor(RGB2LAB)[:,:,2] lane_tracker.py:208
//
// Both resamplings are OpenCV's 8-bit bilinear remap: integer tap (sx,sy) + 5-bit fractions,
// exact 15-bit weights, constant-0 border, (sum + 2^14) >> 15.  They cannot be composed into one
// resampling because the intermediate is rounded to u8, so the undistorted rows are materialised
// once (one RGBX dword per pixel, rows x W x 4 bytes per frame: it stays in L2 for the warp).
// The bird's-eye RGB image itself is never written: the warp emits the R plane and the Lab-b
// plane directly.
//
// Layout of the undistorted rows: slots 2p and 2p+1 are INTERLEAVED per pixel (lt_internal.h: und_slot_base; pixel i
// of slot s is dword und_slot_base(s) + 2 i).  Both remap kernels are bound by the NUMBER of vector-memory
// instructions they issue (profiles/r02_vmem_issue.json: ~20 cycles of a CU's memory pipe per wave-instruction whatever
// its width), and with this layout the two horizontally adjacent taps of a bilinear sample are 16 contiguous bytes that
// hold them for two frames: one dwordx4 load instead of two dwordx2 loads.
#include <algorithm>
#include <cstdlib>

#include "lt_internal.h"

namespace lt {
namespace {

// Every product in this file fits 24 bits, so the multiplies are written with __mul24: a plain 32-bit
// `*` becomes v_mul_lo_u32, which issues at a quarter of the v_mul_u32_u24 / v_mad_u32_u24 rate.
__device__ __forceinline__ int bilerp(int v00, int v01, int v10, int v11, int fx, int fy) {
    const int gx = 32 - fx, gy = 32 - fy;
    // (sum_i w_i v_i + 2^14) >> 15 with w = {gx gy, fx gy, gx fy, fx fy} * 32  ==  (gy h0 + fy h1 + 512) >> 10
    const int h0 = __mul24(v00, gx) + __mul24(v01, fx), h1 = __mul24(v10, gx) + __mul24(v11, fx);
    return (__mul24(h0, gy) + __mul24(h1, fy) + 512) >> 10;
}

// Workgroups are dealt to the 8 XCDs round-robin in linear launch order (x fastest, then y, then z), and each XCD
// has its own L2.  `xcd_block` turns the hardware's linear id into the id this workgroup works on, such that every
// XCD owns one contiguous range of the launch: neighbouring blocks sample overlapping source rows, and with the
// round-robin order each of the 8 L2s fetched its own copy of them (rocprofv3 FETCH_SIZE of the warp: 2.5x the
// source bytes).  Bijective for any launch size; speed only, never correctness.  remap == 0: identity.
__device__ __forceinline__ uint32_t xcd_block(uint32_t lid, uint32_t total, int remap) {
    if (!remap) return lid;
    const uint32_t q = total >> 3, r = total & 7u, xcd = lid & 7u, k = lid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// One thread per undistorted pixel.  Frames are RGB interleaved (3 B/px); the output is one RGBX
// dword per pixel, so that the warp fetches a whole tap with a single aligned load.  All loads are
// unconditional on clamped addresses and masked afterwards (a guarded load serialises on its own
// s_waitcnt).

// ALIGNED4 (frames are 4-byte aligned and so is their stride): the 8-byte window is fetched as the three aligned dwords
// that hold it and shifted into place with two v_alignbyte_b32 -- an unaligned dwordx2 at 3-byte pitch occupies the
// memory pipe for ~32 cycles, an aligned dwordx3 for ~20.
template <bool ALIGNED4>
__global__ __launch_bounds__(256) void k_undistort_rows(const uint8_t* __restrict__ frames, size_t frame_stride,
                                                       const int16_t* __restrict__ uxy,
                                                       const uint16_t* __restrict__ ufrac, FrontEndGeom g,
                                                       uint32_t* __restrict__ und, size_t und_px, int first_slot, int n, int fpb,
                                                       int remap) {
    const uint32_t per_z = gridDim.x * gridDim.y;
    const uint32_t id = xcd_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, per_z * gridDim.z, remap);
    const uint32_t bz = id / per_z, by = (id - bz * per_z) / gridDim.x, bx = id - bz * per_z - by * gridDim.x;
    const int x = bx * blockDim.x + threadIdx.x;
    const int row = by;  // relative to g.r0
    if (x >= g.img_w) return;
    const int z0 = bz * fpb, z1 = min(z0 + fpb, n);   // fpb frames per thread: table entry and offsets are frame-independent
    const size_t o = (size_t)row * g.img_w + x;
    const int sx = uxy[o * 2], sy = uxy[o * 2 + 1];
    const int f = ufrac[o], fx = f & 31, fy = f >> 5;
    const bool y0 = sy >= 0 && sy < g.img_h, y1 = sy + 1 >= 0 && sy + 1 < g.img_h;
    const bool x0 = sx >= 0 && sx < g.img_w, x1 = sx + 1 >= 0 && sx + 1 < g.img_w;
    const int cy0 = min(max(sy, 0), g.img_h - 1), cy1 = min(max(sy + 1, 0), g.img_h - 1);
    // The two taps of a row are 6 consecutive bytes (RGB RGB) at byte offset 3*sx: one 8-byte window per row instead
    // of six byte loads (the frame buffer is padded by 8 bytes for the overrun).  When sx or sx+1 is outside the
    // frame the row is fetched from the clamped column and the taps are masked.
    struct __attribute__((packed, aligned(1))) Row8 { uint64_t v; };
    const int cxl = min(max(sx, 0), g.img_w - 2);          // leftmost column of the 6-byte window we fetch
    // 32-bit offsets (a frame is far below 2^31 bytes) keep the address math on full-rate 24-bit multiplies
    const uint32_t off0 = (uint32_t)((__mul24(cy0, g.img_w) + cxl) * 3), off1 = (uint32_t)((__mul24(cy1, g.img_w) + cxl) * 3);
    // column sx sits at byte 3*(sx-cxl) of the window when it is inside the frame; sx+1 three bytes later
    const int sh0 = 24 * (min(max(sx, 0), g.img_w - 1) - cxl), sh1 = 24 * (min(max(sx + 1, 0), g.img_w - 1) - cxl);
    const uint32_t m00 = (y0 && x0) ? 255u : 0u, m01 = (y0 && x1) ? 255u : 0u;
    const uint32_t m10 = (y1 && x0) ? 255u : 0u, m11 = (y1 && x1) ? 255u : 0u;
    const uint32_t gx = 32u - (uint32_t)fx, gy = 32u - (uint32_t)fy;
    const uint32_t w00 = (gx * gy) & 0x7ffu, w01 = ((uint32_t)fx * gy) & 0x7ffu, w10 = (gx * (uint32_t)fy) & 0x7ffu, w11 = ((uint32_t)fx * (uint32_t)fy) & 0x7ffu;
    const uint8_t* src = frames + (size_t)z0 * frame_stride;
    constexpr int RSRC_RAW = 0x00027000;   // untyped 32-bit buffer, no swizzle
    const int nz = z1 - z0, fstride = (int)frame_stride;
    // +8: the window of the frame's last pixel ends in the padding behind it
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, nz * fstride + 8, RSRC_RAW);
    // the slots this thread writes: pairs [pair0, pair1] of the interleaved buffer
    const int pair0 = (first_slot + z0) >> 1, pair1 = (first_slot + z1 - 1) >> 1, pair_b = (int)(und_px * 8);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(und + (size_t)pair0 * 2 * und_px, 0, (pair1 - pair0 + 1) * pair_b, RSRC_RAW);
    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
    auto window = [&](uint32_t off, int frame_off) {
        if constexpr (ALIGNED4) {
            const u32x3 d = __builtin_amdgcn_raw_buffer_load_b96(frs, (int)(off & ~3u), frame_off, 0);
            const uint32_t lo = __builtin_amdgcn_alignbyte(d.y, d.x, off & 3u), hi = __builtin_amdgcn_alignbyte(d.z, d.y, off & 3u);
            return (uint64_t)lo | ((uint64_t)hi << 32);
        } else {
            return reinterpret_cast<const Row8*>(src + (size_t)(unsigned)frame_off + off)->v;
        }
    };
    uint64_t q0 = window(off0, 0), q1 = window(off1, 0);
    for (int z = z0; z < z1; ++z) {
        // the next frame's taps are in flight while this one is blended
        const int nfo = (min(z + 1, z1 - 1) - z0) * fstride;
        const uint64_t n0 = window(off0, nfo), n1 = window(off1, nfo);
        const uint32_t a0 = (uint32_t)(q0 >> sh0), a1 = (uint32_t)(q0 >> sh1);
        const uint32_t b0 = (uint32_t)(q1 >> sh0), b1 = (uint32_t)(q1 >> sh1);
        uint32_t out = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const uint32_t v00 = (a0 >> (8 * ch)) & m00, v01 = (a1 >> (8 * ch)) & m01;
            const uint32_t v10 = (b0 >> (8 * ch)) & m10, v11 = (b1 >> (8 * ch)) & m11;
            // (sum_i w_i p_i + 2^9) >> 10: the same integer as bilerp(); with the 11-bit weights visible every product is a
            // v_mul_u32_u24 / v_mad_u32_u24 (the two-stage form was re-associated into quarter-rate 32- and 64-bit multiplies)
            out |= ((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + 512u) >> 10) << (8 * ch);
        }
        const int slot = first_slot + z;   // wave-uniform: pair and parity go into the scalar offset of the store
        __builtin_amdgcn_raw_buffer_store_b32(out, urs, (int)o * 8, __builtin_amdgcn_readfirstlane(((slot >> 1) - pair0) * pair_b + (slot & 1) * 4), 0);
        q0 = n0;
        q1 = n1;
    }
}

struct LabLut {
    const uint16_t* gamma_tab;
    const uint16_t* cbrt_tab;
};

__device__ __forceinline__ int lab_b_of(int r, int g, int b, const uint16_t* gt, const uint16_t* ct,
                                        const int32_t* C) {
    const int R = gt[r], G = gt[g], B = gt[b];
    int iy = (__mul24(R, C[3]) + __mul24(G, C[4]) + __mul24(B, C[5]) + (1 << 11)) >> 12;   // R,G,B <= 2040, C < 4096
    int iz = (__mul24(R, C[6]) + __mul24(G, C[7]) + __mul24(B, C[8]) + (1 << 11)) >> 12;
    iy = iy > 3071 ? 3071 : iy;
    iz = iz > 3071 ? 3071 : iz;
    const int fY = ct[iy], fZ = ct[iz];
    const int v = (__mul24(200, fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__device__ __forceinline__ void stage_lab_tables(uint16_t* s_gamma, uint16_t* s_cbrt, int32_t* s_coef,
                                                 const uint16_t* gamma_tab, const uint16_t* cbrt_tab,
                                                 const int32_t* coeffs) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_gamma[i] = gamma_tab[i];
    for (int i = threadIdx.x; i < 3072; i += blockDim.x) s_cbrt[i] = cbrt_tab[i];
    if (threadIdx.x < 9) s_coef[threadIdx.x] = coeffs[threadIdx.x];
    __syncthreads();
}

// exact 8-bit remap blend of the three channels of four RGBX taps:
//   sum_i w_i p_i with w = {(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx fy} * 32, then (s + 2^14) >> 15,
// which equals ((32-fy) h0 + fy h1 + 512) >> 10 with h = (32-fx) p_left + fx p_right  (same integers).
__device__ __forceinline__ void blend_taps(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11, int fx, int fy,
                                           int (&rgb)[3]) {
    const int gx = 32 - fx, gy = 32 - fy;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int h0 = __mul24((int)((t00 >> (8 * ch)) & 255u), gx) + __mul24((int)((t01 >> (8 * ch)) & 255u), fx);
        const int h1 = __mul24((int)((t10 >> (8 * ch)) & 255u), gx) + __mul24((int)((t11 >> (8 * ch)) & 255u), fx);
        rgb[ch] = (__mul24(h0, gy) + __mul24(h1, fy) + 512) >> 10;
    }
}

__device__ __forceinline__ void warp_pixel(const uint32_t* __restrict__ src, const FrontEndGeom& g, int sx, int sy,
                                           int f, const uint16_t* gt, const uint16_t* ct, const int32_t* C, int& r_out,
                                           int& b_out) {
    const int fx = f & 31, fy = f >> 5;
    // taps outside the camera frame are 0; in-frame taps always fall inside rows [r0, r0+nrows)
    const int ry0 = sy - g.r0, ry1 = sy + 1 - g.r0;
    const bool y0 = sy >= 0 && sy < g.img_h && ry0 >= 0 && ry0 < g.nrows;
    const bool y1 = sy + 1 >= 0 && sy + 1 < g.img_h && ry1 >= 0 && ry1 < g.nrows;
    const bool x0 = sx >= 0 && sx < g.img_w, x1 = sx + 1 >= 0 && sx + 1 < g.img_w;
    const int cy0 = min(max(ry0, 0), g.nrows - 1), cy1 = min(max(ry1, 0), g.nrows - 1);
    const int cx0 = min(max(sx, 0), g.img_w - 1), cx1 = min(max(sx + 1, 0), g.img_w - 1);
    const int o0 = __mul24(cy0, g.img_w), o1 = __mul24(cy1, g.img_w);
    uint32_t t00 = src[2 * (o0 + cx0)], t01 = src[2 * (o0 + cx1)];   // pixels of one slot are two dwords apart
    uint32_t t10 = src[2 * (o1 + cx0)], t11 = src[2 * (o1 + cx1)];
    t00 = (y0 && x0) ? t00 : 0u;
    t01 = (y0 && x1) ? t01 : 0u;
    t10 = (y1 && x0) ? t10 : 0u;
    t11 = (y1 && x1) ? t11 : 0u;
    int rgb[3];
    blend_taps(t00, t01, t10, t11, fx, fy, rgb);
    r_out = rgb[0];
    b_out = lab_b_of(rgb[0], rgb[1], rgb[2], gt, ct, C);
}

// Four adjacent bird's-eye pixels per thread: bilinear samples of the RGBX undistorted rows, then
// one dword store to the R plane and one to the Lab-b plane.  `quads` = pixels / 4 (w % 4 == 0).
// The remap table entry of a quad is the same for every frame, so a thread keeps it (and the tap
// offsets derived from it) in registers and walks `ppb` consecutive slot PAIRS with it: the two taps of a row are one
// 16-byte load {left.even, left.odd, right.even, right.odd} that serves both slots of the pair.
// Slots [first_slot, first_slot + n); `und` is the base of the whole buffer, planeR / planeB point at first_slot's planes.

// Latency is hidden by occupancy, not by prefetch: with the taps of the next slot pair in flight (two sets of 8 x 16 bytes)
// the kernel needs 97 VGPRs = 4 waves per SIMD; with one set it fits 64 = 8 waves and runs 6 % faster alone and 3 % faster
// end to end, where it shares the CUs with the other slices' kernels (-DLT_WARP_PREFETCH=1 -DLT_WARP_WAVES=5 for the A/B).
#ifndef LT_WARP_PREFETCH
#define LT_WARP_PREFETCH 0
#endif
#ifndef LT_WARP_WAVES
#define LT_WARP_WAVES 8
#endif
#ifndef LT_WARP_DOT
#define LT_WARP_DOT 0   // measured: 11 % fewer VALU instructions (324 -> 288 per 8 pixels), the same 0.537 ms per 256 frames -- the kernel is co-bound by the vector-memory pipe
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LT_WARP_WAVES, LT_WARP_WAVES))) void k_warp_split4(const uint32_t* __restrict__ und, size_t und_px, int first_slot,
                                                    const int16_t* __restrict__ wxy,
                                                    const uint16_t* __restrict__ wfrac, FrontEndGeom g,
                                                    const uint16_t* __restrict__ gamma_tab,
                                                    const uint16_t* __restrict__ cbrt_tab,
                                                    const int32_t* __restrict__ coeffs, uint8_t* __restrict__ planeR,
                                                    uint8_t* __restrict__ planeB, size_t plane_stride, int n, int ppb,
                                                    int remap) {
    __shared__ uint16_t s_gamma[256];
    __shared__ uint16_t s_cbrt[3072];
    __shared__ int32_t s_coef[9];
    // gamma LUT and the Y / Z rows of the matrix folded into one table per channel: s_yz[ch][v] = gamma[v] * (C[3+ch], C[6+ch]),
    // the rounding constant of DESCALE(., 12) added to the red entries -- one 8-byte LDS read per channel replaces a 16-bit
    // read, two multiplies / multiply-adds and the rounding add (the kernel is at 80 % of the VALU issue ceiling)
    __shared__ uint2 s_yz[3][256];
    const size_t quads = ((size_t)g.warp_h * g.warp_w) >> 2;
    const uint32_t id = xcd_block(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x * gridDim.z, remap);
    const uint32_t bz = id / gridDim.x, bx = id - bz * gridDim.x;
    const size_t qi = (size_t)bx * blockDim.x + threadIdx.x;
    const size_t qc = qi < quads ? qi : quads - 1;
    // the table entry is requested before the Lab tables are staged: both latencies overlap
    const uint4 xy = reinterpret_cast<const uint4*>(wxy)[qc];        // 4 x (sx, sy) int16 pairs
    const uint2 fr = reinterpret_cast<const uint2*>(wfrac)[qc];      // 4 x u16
    {
        const int v = threadIdx.x;   // 256 threads: one table row each
        const uint32_t gv = gamma_tab[v];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            s_yz[ch][v] = make_uint2(gv * (uint32_t)coeffs[3 + ch] + (ch == 0 ? 2048u : 0u), gv * (uint32_t)coeffs[6 + ch] + (ch == 0 ? 2048u : 0u));
    }
    stage_lab_tables(s_gamma, s_cbrt, s_coef, gamma_tab, cbrt_tab, coeffs);
    if (qi >= quads) return;
    // this walk: pairs [pa, pb) of the buffer, i.e. slots [2 pa, 2 pb) clipped to [first_slot, first_slot + n)
    const int pair_lo = first_slot >> 1, pair_hi = (first_slot + n + 1) >> 1;
    const int pa = pair_lo + (int)bz * ppb, pb = min(pa + ppb, pair_hi);
    const int s_lo = max(2 * pa, first_slot), s_hi = min(2 * pb, first_slot + n);   // slots this walk writes
    if (s_hi <= s_lo) return;
    uint8_t* const outR = planeR + (size_t)(s_lo - first_slot) * plane_stride;
    uint8_t* const outB = planeB + (size_t)(s_lo - first_slot) * plane_stride;
    const uint32_t xyv[4] = {xy.x, xy.y, xy.z, xy.w};
    const uint32_t frv[4] = {fr.x & 0xffffu, fr.x >> 16, fr.y & 0xffffu, fr.y >> 16};
    // 87 % of the bird's-eye view samples strictly inside the staged rows: skip every border test there
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sx = (int16_t)(xyv[i] & 0xffffu), sy = (int16_t)(xyv[i] >> 16);
        inside = inside && sx >= 0 && sx + 1 < g.img_w && sy >= g.r0 && sy + 1 < g.r0 + g.nrows && sy + 1 < g.img_h;
    }
    if (inside) {
        // The four tap weights of a pixel (products of the 5-bit fractions, <= 1024) are frame-independent.  They
        // are masked to 11 bits on purpose: the compiler folds (a gx + b fx) gy into a (gx gy) + b (fx gy) anyway,
        // and unless it can see that the weight products are small it multiplies with v_mul_lo_u32 (quarter rate)
        // instead of v_mul_u32_u24 -- six of them per pixel in the previous version of this loop.
        uint32_t w00[4], w01[4], w10[4], w11[4];
        // LT_WARP_DOT: the two-stage form of the same integer, two rows at a time in packed 16-bit lanes --
        //   (h0, h1) = (top.left, bottom.left) * gx + (top.right, bottom.right) * fx      v_pk_mul_lo_u16 + v_pk_mad_u16  (<= 8160)
        //   value    = (h0 * gy + h1 * fy + 512) >> 10                                       v_dot2_u32_u16 + shift
        // with one v_perm_b32 per tap pair to put a channel of the top and the bottom tap into the two lanes: 6 instructions per
        // channel instead of 4 field extracts + 4 multiply-adds + add + shift.  gx2 / fx2 hold the weight in both lanes.
        uint32_t gx2[4], fx2[4], gyfy[4];
        int toff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sx = (int16_t)(xyv[i] & 0xffffu), sy = (int16_t)(xyv[i] >> 16);
            toff[i] = (__mul24(sy - g.r0, g.img_w) + sx) * 8;            // byte offset of the left tap inside a pair
            const uint32_t fx = frv[i] & 31u, fy = frv[i] >> 5, gx = 32u - fx, gy = 32u - fy;
            w00[i] = (gx * gy) & 0x7ffu;
            w01[i] = (fx * gy) & 0x7ffu;
            w10[i] = (gx * fy) & 0x7ffu;
            w11[i] = (fx * fy) & 0x7ffu;
            gx2[i] = gx | (gx << 16);
            fx2[i] = fx | (fx << 16);
            gyfy[i] = gy | (fy << 16);
        }
        // Taps and outputs go through buffer descriptors of this walk: descriptor + 32-bit lane offset + scalar
        // pair / frame offset, so no access pays a 64-bit VALU address add.
        constexpr int RSRC_RAW = 0x00027000;   // untyped 32-bit buffer, no swizzle
        const int pair_b = (int)(und_px * 8);
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(und + (size_t)pa * 2 * und_px), 0, (pb - pa) * pair_b, RSRC_RAW);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(outR, 0, (s_hi - s_lo) * (int)plane_stride, RSRC_RAW);
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(outB, 0, (s_hi - s_lo) * (int)plane_stride, RSRC_RAW);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int row_b = g.img_w * 8, out_off = (int)qi * 4;
        u32x4 tapA[8], tapB[8];   // [0..3] top row, [4..7] bottom row of the four pixels; A / B alternate between pairs
        auto fetch = [&](u32x4 (&t)[8], int p) __attribute__((always_inline)) {
            const int po = (min(p, pb - 1) - pa) * pair_b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t[i] = __builtin_amdgcn_raw_buffer_load_b128(urs, toff[i], po, 0);
                t[4 + i] = __builtin_amdgcn_raw_buffer_load_b128(urs, toff[i] + row_b, po, 0);
            }
        };
        auto blend_pair = [&](const u32x4 (&t)[8], int p) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int slot = 2 * p + f;
                if (slot < s_lo || slot >= s_hi) continue;     // wave-uniform: the odd slot in front of / behind the range
                uint32_t oR = 0, oB = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t ta = t[i][f], tb = t[i][2 + f], ba = t[4 + i][f], bb = t[4 + i][2 + f];
                    int rgb[3];
#if LT_WARP_DOT
                    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        // lanes (top, bottom) of channel ch: bytes ch of the top tap and of the bottom tap, zero-extended
                        const uint32_t selc = 0x0c000c00u | (uint32_t)ch | ((uint32_t)(4 + ch) << 16);
                        const u16x2 L = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(ba, ta, selc));
                        const u16x2 Rr = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(bb, tb, selc));
                        const u16x2 H = L * __builtin_bit_cast(u16x2, gx2[i]) + Rr * __builtin_bit_cast(u16x2, fx2[i]);
                        rgb[ch] = (int)(__builtin_amdgcn_udot2(H, __builtin_bit_cast(u16x2, gyfy[i]), 512u, false) >> 10);
                    }
#else
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)     // (sum_i w_i p_i + 2^9) >> 10: the same integer as the two-stage blend
                        rgb[ch] = (int)((((ta >> (8 * ch)) & 255u) * w00[i] + ((tb >> (8 * ch)) & 255u) * w01[i] +
                                         ((ba >> (8 * ch)) & 255u) * w10[i] + ((bb >> (8 * ch)) & 255u) * w11[i] + 512u) >> 10);
#endif
                    int r = rgb[0], b;
                    {
                        const uint2 yr = s_yz[0][rgb[0]], yg = s_yz[1][rgb[1]], yb = s_yz[2][rgb[2]];
                        int iy = (int)((yr.x + yg.x + yb.x) >> 12), iz = (int)((yr.y + yg.y + yb.y) >> 12);   // sums < 2^24
                        iy = iy > 3071 ? 3071 : iy;
                        iz = iz > 3071 ? 3071 : iz;
                        const int fY = s_cbrt[iy], fZ = s_cbrt[iz];
                        const int v = (__mul24(200, fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15;
                        b = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    // Opaque to the optimiser on purpose: with the value ranges visible, hipcc (ROCm 7.2) folded the
                    // four byte inserts into a 16-bit combine that leaked bits 16+ of an unshifted Lab value into the
                    // third pixel (caught by the parity test); the barrier costs nothing at run time.
#ifndef LT_CASE_WARP_NO_BARRIER   // tools/toolchain_cases.sh builds the kernel without it to check whether the case still exists
                    asm volatile("" : "+v"(r), "+v"(b));
#endif
                    oR |= ((uint32_t)r & 255u) << (8 * i);
                    oB |= ((uint32_t)b & 255u) << (8 * i);
                }
                __builtin_amdgcn_raw_buffer_store_b32(oR, rrs, out_off, (slot - s_lo) * (int)plane_stride, 0);
                __builtin_amdgcn_raw_buffer_store_b32(oB, brs, out_off, (slot - s_lo) * (int)plane_stride, 0);
            }
        };
        // two pairs per trip, the tap registers alternate: the taps of the next pair are in flight while this one is
        // blended, and nothing is copied at the back edge
#if LT_WARP_PREFETCH
        fetch(tapA, pa);
        for (int p = pa; p < pb; p += 2) {
            fetch(tapB, p + 1);
            blend_pair(tapA, p);
            if (p + 1 >= pb) break;
            fetch(tapA, p + 2);
            blend_pair(tapB, p + 1);
        }
#else
        for (int p = pa; p < pb; ++p) {
            fetch(tapA, p);
            blend_pair(tapA, p);
        }
        (void)tapB;
#endif
        return;
    }
    // 13 % of the bird's-eye view (the bottom corner triangles) samples entirely outside the camera frame: every tap
    // is the constant border 0, so R = 0 and Lab-b = b(0,0,0) for every frame -- no loads, no blend.
    bool none = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sx = (int16_t)(xyv[i] & 0xffffu), sy = (int16_t)(xyv[i] >> 16), ry0 = sy - g.r0, ry1 = ry0 + 1;
        const bool y0 = sy >= 0 && sy < g.img_h && ry0 >= 0 && ry0 < g.nrows;
        const bool y1 = sy + 1 >= 0 && sy + 1 < g.img_h && ry1 >= 0 && ry1 < g.nrows;
        const bool x0 = sx >= 0 && sx < g.img_w, x1 = sx + 1 >= 0 && sx + 1 < g.img_w;
        none = none && !((y0 || y1) && (x0 || x1));
    }
    if (none) {
        const uint32_t oB = (uint32_t)lab_b_of(0, 0, 0, s_gamma, s_cbrt, s_coef) * 0x01010101u;
        for (int slot = s_lo; slot < s_hi; ++slot) {
            reinterpret_cast<uint32_t*>(outR + (size_t)(slot - s_lo) * plane_stride)[qi] = 0u;
            reinterpret_cast<uint32_t*>(outB + (size_t)(slot - s_lo) * plane_stride)[qi] = oB;
        }
        return;
    }
    for (int slot = s_lo; slot < s_hi; ++slot) {
        const uint32_t* src = und + und_slot_base(und_px, slot);
        uint32_t oR = 0, oB = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sx = (int16_t)(xyv[i] & 0xffffu), sy = (int16_t)(xyv[i] >> 16);
            int r, b;
            warp_pixel(src, g, sx, sy, (int)frv[i], s_gamma, s_cbrt, s_coef, r, b);
            asm volatile("" : "+v"(r), "+v"(b));
            oR |= ((uint32_t)r & 255u) << (8 * i);
            oB |= ((uint32_t)b & 255u) << (8 * i);
        }
        reinterpret_cast<uint32_t*>(outR + (size_t)(slot - s_lo) * plane_stride)[qi] = oR;
        reinterpret_cast<uint32_t*>(outB + (size_t)(slot - s_lo) * plane_stride)[qi] = oB;
    }
}

// any width: one pixel per thread
__global__ __launch_bounds__(256) void k_warp_split1(const uint32_t* __restrict__ und, size_t und_px, int first_slot,
                                                    const int16_t* __restrict__ wxy,
                                                    const uint16_t* __restrict__ wfrac, FrontEndGeom g,
                                                    const uint16_t* __restrict__ gamma_tab,
                                                    const uint16_t* __restrict__ cbrt_tab,
                                                    const int32_t* __restrict__ coeffs, uint8_t* __restrict__ planeR,
                                                    uint8_t* __restrict__ planeB, size_t plane_stride) {
    __shared__ uint16_t s_gamma[256];
    __shared__ uint16_t s_cbrt[3072];
    __shared__ int32_t s_coef[9];
    stage_lab_tables(s_gamma, s_cbrt, s_coef, gamma_tab, cbrt_tab, coeffs);
    const size_t npix = (size_t)g.warp_h * g.warp_w;
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= npix) return;
    int r, b;
    warp_pixel(und + und_slot_base(und_px, first_slot + (int)blockIdx.z), g, wxy[o * 2], wxy[o * 2 + 1], wfrac[o], s_gamma, s_cbrt,
               s_coef, r, b);
    planeR[(size_t)blockIdx.z * plane_stride + o] = (uint8_t)r;
    planeB[(size_t)blockIdx.z * plane_stride + o] = (uint8_t)b;
}

// filter_lane_points() entry on an already-warped RGB image (lane_tracker.py:207-208)
__global__ __launch_bounds__(256) void k_split_bev(const uint8_t* __restrict__ bev, size_t bev_stride, int npix,
                                                  const uint16_t* __restrict__ gamma_tab,
                                                  const uint16_t* __restrict__ cbrt_tab,
                                                  const int32_t* __restrict__ coeffs, uint8_t* __restrict__ planeR,
                                                  uint8_t* __restrict__ planeB, size_t plane_stride) {
    __shared__ uint16_t s_gamma[256];
    __shared__ uint16_t s_cbrt[3072];
    __shared__ int32_t s_coef[9];
    stage_lab_tables(s_gamma, s_cbrt, s_coef, gamma_tab, cbrt_tab, coeffs);
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (size_t)npix) return;
    const uint8_t* p = bev + (size_t)blockIdx.z * bev_stride + o * 3;
    const int r = p[0], gg = p[1], b = p[2];
    planeR[(size_t)blockIdx.z * plane_stride + o] = (uint8_t)r;
    planeB[(size_t)blockIdx.z * plane_stride + o] = (uint8_t)lab_b_of(r, gg, b, s_gamma, s_cbrt, s_coef);
}

__global__ __launch_bounds__(256) void k_undistorted_to_rgb(const uint32_t* __restrict__ und, size_t und_px, int first_slot,
                                                           int nrows, int w, uint8_t* __restrict__ out) {
    const size_t plane = (size_t)nrows * w;
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= plane) return;
    const uint32_t v = und[und_slot_base(und_px, first_slot + (int)blockIdx.z) + 2 * o];
    uint8_t* dst = out + (size_t)blockIdx.z * plane * 3 + o * 3;
    dst[0] = (uint8_t)v;
    dst[1] = (uint8_t)(v >> 8);
    dst[2] = (uint8_t)(v >> 16);
}

}  // namespace

// Frames one thread walks with its remap-table entry: as many as possible (the table read, the tap address
// arithmetic and one memory round trip are paid once per walk) while the launch still has a few groups of
// frames, so that small batches keep their parallelism.
static int xcd_remap() {
    static const int v = [] { const char* e = LT_EXP_ENV("LT_XCD_REMAP"); return e ? std::atoi(e) : 1; }();
    return v;
}

static int frames_per_thread(int n) {
    static const int cap = [] { const char* e = LT_EXP_ENV("LT_FRONTEND_FPB"); int v = e ? std::atoi(e) : 16; return v < 1 ? 1 : v; }();
    int fpb = n / 8;
    return fpb < 1 ? 1 : (fpb > cap ? cap : fpb);
}

void launch_undistort_rows(hipStream_t s, const uint8_t* frames, size_t frame_stride, const int16_t* uxy,
                           const uint16_t* ufrac, FrontEndGeom g, uint32_t* und, size_t und_px, int first_slot, int n) {
    if (n <= 0 || g.nrows <= 0) return;
    const int fpb = frames_per_thread(n);
    dim3 grid((g.img_w + 255) / 256, g.nrows, (n + fpb - 1) / fpb);
    static const bool unaligned = [] { const char* e = LT_EXP_ENV("LT_UNDISTORT_UNALIGNED"); return e && e[0] == '1'; }();   // A/B
    if (!unaligned && ((uintptr_t)frames & 3) == 0 && (frame_stride & 3) == 0 && (size_t)fpb * frame_stride < (1u << 30))
        hipLaunchKernelGGL(k_undistort_rows<true>, grid, dim3(256), 0, s, frames, frame_stride, uxy, ufrac, g, und, und_px, first_slot, n, fpb, xcd_remap());
    else
        hipLaunchKernelGGL(k_undistort_rows<false>, grid, dim3(256), 0, s, frames, frame_stride, uxy, ufrac, g, und, und_px, first_slot, n, fpb, xcd_remap());
}

void launch_warp_split(hipStream_t s, const uint32_t* und, size_t und_px, int first_slot, const int16_t* wxy,
                       const uint16_t* wfrac, FrontEndGeom g, const uint16_t* gamma_tab, const uint16_t* cbrt_tab,
                       const int32_t* coeffs, uint8_t* planeR, uint8_t* planeB, size_t plane_stride, int n) {
    if (n <= 0 || g.nrows <= 0) return;
    const size_t npix = (size_t)g.warp_h * g.warp_w;
    if ((g.warp_w & 3) == 0 && (plane_stride & 3) == 0) {
        const int pairs = ((first_slot + n + 1) >> 1) - (first_slot >> 1);
        const int ppb = std::max(1, frames_per_thread(n) / 2);
        dim3 grid((unsigned)(((npix >> 2) + 255) / 256), 1, (pairs + ppb - 1) / ppb);
        hipLaunchKernelGGL(k_warp_split4, grid, dim3(256), 0, s, und, und_px, first_slot, wxy, wfrac, g, gamma_tab, cbrt_tab,
                           coeffs, planeR, planeB, plane_stride, n, ppb, xcd_remap());
    } else {
        dim3 grid((unsigned)((npix + 255) / 256), 1, n);
        hipLaunchKernelGGL(k_warp_split1, grid, dim3(256), 0, s, und, und_px, first_slot, wxy, wfrac, g, gamma_tab, cbrt_tab,
                           coeffs, planeR, planeB, plane_stride);
    }
}

void launch_split_bev(hipStream_t s, const uint8_t* bev, size_t bev_stride, int npix, const uint16_t* gamma_tab,
                      const uint16_t* cbrt_tab, const int32_t* coeffs, uint8_t* planeR, uint8_t* planeB,
                      size_t plane_stride, int n) {
    if (n <= 0) return;
    dim3 grid((unsigned)((npix + 255) / 256), 1, n);
    hipLaunchKernelGGL(k_split_bev, grid, dim3(256), 0, s, bev, bev_stride, npix, gamma_tab, cbrt_tab, coeffs, planeR,
                       planeB, plane_stride);
}

void launch_undistorted_to_rgb(hipStream_t s, const uint32_t* und, size_t und_px, int first_slot, int nrows, int w, uint8_t* out,
                               int n) {
    if (n <= 0 || nrows <= 0) return;
    dim3 grid((unsigned)(((size_t)nrows * w + 255) / 256), 1, n);
    hipLaunchKernelGGL(k_undistorted_to_rgb, grid, dim3(256), 0, s, und, und_px, first_slot, nrows, w, out);
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_frontend() {} }
void preload_k_frontend(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_frontend, dim3(1), dim3(1), 0, s); }

}  // namespace lt

