// Lane-pixel search + polynomial fit, gfx950.  One 256-thread workgroup per frame.
//
//   k_sws_fit    LaneTracker.sliding_window_search + fit_poly   lane_tracker.py:242-447, 502-509
//   k_band_fit   LaneTracker.band_search + fit_poly             lane_tracker.py:449-509
//
// Column histograms are LDS atomics, the first/last-argmax of the box-filtered histogram and the
// per-row stream compaction are wavefront ballots/shuffles (64 lanes), and the 2nd-degree fit is a
// per-lane 3x3 normal-equations solve on exact int64 moments taken about the image centre.
// Lane pixels are emitted in the reference's order (level-major, then row-major inside a window;
// row-major for the band search) as packed (y << 16) | x.
#include <algorithm>
#include <cstdlib>

#include "lt_internal.h"

namespace lt {
namespace {

constexpr int NT = 256;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Cross-lane steps on DPP (a few cycles each) instead of ds_bpermute shuffles (an LDS round trip each): the
// recurrence is one wave walking a dependent chain, so every round trip is on the critical path.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, BOUND);
}
// max over the wave, returned wave-uniform: rotate-and-max inside each row of 16, then four readlanes
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
    v = max(v, dpp_u32<0x121, 0xf, false>(v, v));   // row_ror:1
    v = max(v, dpp_u32<0x122, 0xf, false>(v, v));   // row_ror:2
    v = max(v, dpp_u32<0x124, 0xf, false>(v, v));   // row_ror:4
    v = max(v, dpp_u32<0x128, 0xf, false>(v, v));   // row_ror:8
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row (row_shr 1, 2, 4, 8 with zero fill),
// then the row totals with row_bcast:15 (into rows 1, 3) and row_bcast:31 (into rows 2, 3)
__device__ __forceinline__ unsigned wave_inclusive_sum_dpp(unsigned v) {
    v += dpp_u32<0x111, 0xf, true>(0u, v);
    v += dpp_u32<0x112, 0xf, true>(0u, v);
    v += dpp_u32<0x114, 0xf, true>(0u, v);
    v += dpp_u32<0x118, 0xf, true>(0u, v);
    v += dpp_u32<0x142, 0xa, false>(0u, v);
    v += dpp_u32<0x143, 0xc, false>(0u, v);
    return v;
}

// sum of a 64-bit value over the wave (wave-uniform result, two's complement): three limbs of 22/22/20
// bits, so that each limb's 64-lane sum fits 32 bits and can use the DPP scan
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    const unsigned long long u = (unsigned long long)v;
    const unsigned s0 = wave_inclusive_sum_dpp((unsigned)(u & 0x3fffffu));
    const unsigned s1 = wave_inclusive_sum_dpp((unsigned)((u >> 22) & 0x3fffffu));
    const unsigned s2 = wave_inclusive_sum_dpp((unsigned)(u >> 44));
    const unsigned long long t0 = (unsigned)__builtin_amdgcn_readlane((int)s0, 63), t1 = (unsigned)__builtin_amdgcn_readlane((int)s1, 63),
                             t2 = (unsigned)__builtin_amdgcn_readlane((int)s2, 63);
    return (long long)(t0 + (t1 << 22) + (t2 << 44));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// Moments of the lane pixels about (y0, x0), exact in int64:
//   m[0..4] = sum (y-y0)^k, k = 0..4 ;  m[5..7] = sum (x-x0) (y-y0)^k, k = 0..2
struct Moments {
    long long m[8];
    __device__ void clear() {
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = 0;
    }
    __device__ void add(int y, int x, int y0, int x0) {
        const long long dy = y - y0, dx = x - x0, dy2 = dy * dy;
        m[0] += 1; m[1] += dy; m[2] += dy2; m[3] += dy2 * dy; m[4] += dy2 * dy2;
        m[5] += dx; m[6] += dx * dy; m[7] += dx * dy2;
    }
};

// Least-squares parabola x = a y^2 + b y + c from the centred moments (np.polyfit(y, x, 2)).
// The normal equations are solved in f64 on variables scaled to O(1); returns false when the
// system is rank deficient (fewer than 3 distinct y), which the host then handles like NumPy.
__device__ bool solve_poly2(const long long* m, int distinct_rows, double y0, double x0, double sy, double* out) {
    out[0] = out[1] = out[2] = 0.0;
    if (m[0] <= 0) return false;
    if (distinct_rows < 3) return false;
    const double is = 1.0 / sy, is2 = is * is, is3 = is2 * is, is4 = is2 * is2;
    // u = (y - y0)/sy ; x' = x - x0 ;  [S4 S3 S2; S3 S2 S1; S2 S1 S0] [A B C]' = [T2 T1 T0]'
    const double S0 = (double)m[0], S1 = (double)m[1] * is, S2 = (double)m[2] * is2, S3 = (double)m[3] * is3,
                 S4 = (double)m[4] * is4;
    const double T0 = (double)m[5], T1 = (double)m[6] * is, T2 = (double)m[7] * is2;
    // symmetric positive definite: Cholesky  G = L L'
    const double l00 = sqrt(S4);
    if (!(l00 > 0.0)) return false;
    const double l10 = S3 / l00, l20 = S2 / l00;
    const double d1 = S2 - l10 * l10;
    if (!(d1 > 0.0)) return false;
    const double l11 = sqrt(d1);
    const double l21 = (S1 - l20 * l10) / l11;
    const double d2 = S0 - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) return false;
    const double l22 = sqrt(d2);
    const double z0 = T2 / l00, z1 = (T1 - l10 * z0) / l11, z2 = (T0 - l20 * z0 - l21 * z1) / l22;
    const double C = z2 / l22, B = (z1 - l21 * C) / l11, A = (z0 - l10 * B - l20 * C) / l00;
    // x - x0 = A u^2 + B u + C with u = (y - y0)/sy
    const double a = A * is2, b = B * is;
    out[0] = a;
    out[1] = b - 2.0 * a * y0;
    out[2] = a * y0 * y0 - b * y0 + C + x0;
    return true;
}

// exclusive prefix of an LDS int array of n entries into out[0..n] (out[n] = total), by wave 0
__device__ void wave0_exclusive_scan(const unsigned* in, unsigned* out, int n) {
    if (wave_id() != 0) return;
    const int lane = lane_id(), chunk = (n + 63) / 64;
    const int a = min(lane * chunk, n), b = min(a + chunk, n);
    unsigned s = 0;
    for (int i = a; i < b; ++i) s += in[i];
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += t;
    }
    unsigned run = incl - s;
    for (int i = a; i < b; ++i) {
        const unsigned v = in[i];
        out[i] = run;
        run += v;
    }
    if (lane == 63) out[n] = incl;
}

// ---- column sums of every search band, all frames at once -------------------------------------------
// np.sum(img[r0:r1, :], axis=0) (lane_tracker.py:290, 310, 350) for band 0 = the start slice
// [y_start, img_height) and bands 1..nlevels-1 = the level rows.  One thread per 4 columns
// (dword loads, unconditional and batched), sums[frame][band][w] as u32.
constexpr int BS_RG = 8;   // row groups per workgroup: the start slice is hundreds of rows tall, and a single
                           // frame must not spend them one after the other in one thread
template <bool VEC4>
__global__ __launch_bounds__(64 * BS_RG) void k_band_sums(const uint8_t* __restrict__ masks, size_t mask_stride,
                                                         SearchGeom g, uint32_t* __restrict__ sums) {
    __shared__ uint4 part[BS_RG][64];
    const int band = blockIdx.y, frame = blockIdx.z, tx = threadIdx.x, ty = threadIdx.y;
    const int r0 = max(band == 0 ? max(g.y_start, 0) : g.img_height - (1 + band) * g.wh, 0);
    const int r1 = band == 0 ? g.img_height : g.img_height - band * g.wh;
    const uint8_t* m = masks + (size_t)frame * mask_stride;
    uint32_t* out = sums + ((size_t)frame * g.nbands + band) * g.w;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (VEC4) {
        const int nq = g.w >> 2, q = min(blockIdx.x * 64 + tx, nq - 1);
        const uint32_t* col = reinterpret_cast<const uint32_t*>(m) + q;
        for (int y = r0 + ty; y < r1; y += BS_RG * 8) {      // rows y, y + 8, ..., all in flight
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(size_t)min(y + BS_RG * u, g.h - 1) * nq];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (y + BS_RG * u < r1) { a0 += v[u] & 255u; a1 += (v[u] >> 8) & 255u; a2 += (v[u] >> 16) & 255u; a3 += v[u] >> 24; }
        }
    } else {
        const int x = min(blockIdx.x * 64 + tx, g.w - 1);
        for (int y = r0 + ty; y < r1; y += BS_RG) a0 += m[(size_t)y * g.w + x];
    }
    part[ty][tx] = make_uint4(a0, a1, a2, a3);
    __syncthreads();
    if (ty != 0) return;
#pragma unroll
    for (int j = 1; j < BS_RG; ++j) {
        const uint4 p = part[j][tx];
        a0 += p.x; a1 += p.y; a2 += p.z; a3 += p.w;
    }
    if (VEC4) {
        const int q = blockIdx.x * 64 + tx;
        if (q < (g.w >> 2)) reinterpret_cast<uint4*>(out)[q] = make_uint4(a0, a1, a2, a3);
    } else {
        const int x = blockIdx.x * 64 + tx;
        if (x < g.w) out[x] = a0;
    }
}

// The same sums from the opened bit plane: counts of set pixels (the searches only compare and maximise the
// sums, so the factor 255 of a {0,255} mask drops out).  The rows of a band are contiguous in the bit plane: a
// workgroup copies them to LDS (coalesced, a few instructions), then each wave takes a 64-column word, a lane
// per column, and walks the rows with broadcast LDS reads (all lanes read the same word).
constexpr int BSB_ROWS = 256;   // rows staged at a time (35 KB at 17 words per row)
constexpr int BSB_NT = 1024;    // 16 waves: the start slice of a single frame is one workgroup's job
__global__ __launch_bounds__(BSB_NT) void k_band_sums_bits(MaskBits mb, SearchGeom g, uint32_t* __restrict__ sums) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* sw = reinterpret_cast<unsigned long long*>(smem);
    const int band = blockIdx.y, frame = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = max(band == 0 ? max(g.y_start, 0) : g.img_height - (1 + band) * g.wh, 0);
    const int r1 = band == 0 ? g.img_height : g.img_height - band * g.wh;
    const unsigned long long* fb = mb.bits + (size_t)frame * mb.bits_stride;
    uint32_t* out = sums + ((size_t)frame * g.nbands + band) * g.w;
    constexpr int NW = BSB_NT / 64;
    const int wpr = mb.wpr, jpw = (wpr + NW - 1) / NW;     // words per row; words handled by each wave
    uint32_t acc[4];                                       // jpw <= 4 (w <= 4096)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = 0;
    for (int c0 = r0; c0 < r1; c0 += BSB_ROWS) {
        const int nr = min(BSB_ROWS, r1 - c0), nw = nr * wpr;
        __syncthreads();
        for (int i = threadIdx.x; i < nw; i += BSB_NT) sw[i] = fb[(size_t)c0 * wpr + i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = wv + NW * q;
            if (q >= jpw || j >= wpr) break;
            uint32_t a = 0;
            int r = 0;
            for (; r + 8 <= nr; r += 8) {                      // eight broadcast reads in flight
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = sw[(r + u) * wpr + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (uint32_t)(v[u] >> lane) & 1u;
            }
            for (; r < nr; ++r) a += (uint32_t)(sw[r * wpr + j] >> lane) & 1u;
            acc[q] += a;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = wv + NW * q, x = j * 64 + lane;
        if (q < jpw && j < wpr && x < g.w) out[x] = acc[q];
    }
}

// copy n band sums (global) into LDS, coalesced and batched
__device__ void load_sums(const uint32_t* __restrict__ src, int n, unsigned* dst) {
    for (int base = threadIdx.x; base < n; base += NT * 4) {
        unsigned v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[min(base + u * NT, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * NT < n) dst[base + u * NT] = v[u];
    }
    __syncthreads();
}

// first/last argmax of conv[a:b), conv = np.convolve(ones(ww), sums[0:ncnt)) ('full'), from the
// exclusive prefix array; every wave computes the same answer redundantly.  false <=> np.any false.
__device__ bool box_argmax(const unsigned* prefix, int ncnt, int ww, int a, int b, int& first, int& last) {
    const int lane = lane_id();
    unsigned best = 0;
    for (int k = a + lane; k < b; k += 64) {
        const int lo = max(k - ww + 1, 0), hi = min(k + 1, ncnt);
        const unsigned v = hi > lo ? prefix[hi] - prefix[lo] : 0u;
        best = max(best, v);
    }
    best = wave_max_u32(best);
    if (best == 0) return false;
    int f = 0x7fffffff, l = -1;
    for (int k = a + lane; k < b; k += 64) {
        const int lo = max(k - ww + 1, 0), hi = min(k + 1, ncnt);
        const unsigned v = hi > lo ? prefix[hi] - prefix[lo] : 0u;
        if (v == best) { f = min(f, k); l = max(l, k); }
    }
    first = wave_min_i32(f) - a;
    last = wave_max_i32(l) - a;
    return true;
}

struct Roi {
    int active, a, b;  // columns [a, b), already clipped; active = 0: no window on this level
};

// Non-zero pixels of columns [a, b) of one row, in ascending x, through aligned dword loads.
// EMIT = false: returns the count.  EMIT = true: writes packed (y<<16)|x from pix[idx], adds moments.
template <bool EMIT, bool VEC4>
__device__ __forceinline__ int roi_row(const uint8_t* __restrict__ row, int a, int b, int y, uint32_t* pix, int idx,
                                       int maxpix, Moments& mom, int y0c, int x0c) {
    int cnt = 0;
    if (VEC4) {
        const int xa = a & ~3, nd = (b - xa + 3) >> 2;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(row + xa);
        for (int d0 = 0; d0 < nd; d0 += 4) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[min(d0 + u, nd - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (d0 + u >= nd) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = xa + (d0 + u) * 4 + j;
                    if (((v[u] >> (8 * j)) & 255u) != 0 && x >= a && x < b) {
                        if (EMIT) {
                            if (idx + cnt < maxpix) pix[idx + cnt] = ((uint32_t)y << 16) | (uint32_t)x;
                            mom.add(y, x, y0c, x0c);
                        }
                        ++cnt;
                    }
                }
            }
        }
    } else {
        for (int x = a; x < b; ++x)
            if (row[x] != 0) {
                if (EMIT) {
                    if (idx + cnt < maxpix) pix[idx + cnt] = ((uint32_t)y << 16) | (uint32_t)x;
                    mom.add(y, x, y0c, x0c);
                }
                ++cnt;
            }
    }
    return cnt;
}

// roi.nonzero() for the (up to) two windows of one level, rows [r0, r1): one thread per
// (side, row) counts its row, offsets are a short serial sum, then the same thread emits its
// pixels in order.  n_out[side] is advanced; moments and distinct-row counts accumulate.
template <bool VEC4>
__device__ void extract_windows(const uint8_t* mask, int w, int r0, int r1, const Roi* roi, unsigned* rowcnt,
                                unsigned* rowoff, int wh_cap, uint32_t* pix, int maxpix, int* n_out, Moments* mom,
                                int* distinct, int y0c, int x0c) {
    const int nrows = r1 - r0, pairs = 2 * nrows;
    for (int pi = threadIdx.x; pi < pairs; pi += NT) {
        const int s = pi >= nrows ? 1 : 0, ry = pi - s * nrows;
        unsigned c = 0;
        if (roi[s].active && roi[s].b > roi[s].a)
            c = roi_row<false, VEC4>(mask + (size_t)(r0 + ry) * w, roi[s].a, roi[s].b, r0 + ry, nullptr, 0, 0, mom[s], y0c, x0c);
        rowcnt[s * wh_cap + ry] = c;
    }
    __syncthreads();
    for (int pi = threadIdx.x; pi < pairs + 2; pi += NT) {   // exclusive offsets; entry nrows = total
        const int s = pi >= nrows + 1 ? 1 : 0, ry = pi - s * (nrows + 1);
        unsigned acc = 0;
        for (int q = 0; q < ry; ++q) acc += rowcnt[s * wh_cap + q];
        rowoff[s * (wh_cap + 1) + ry] = acc;
    }
    __syncthreads();
    for (int pi = threadIdx.x; pi < pairs; pi += NT) {
        const int s = pi >= nrows ? 1 : 0, ry = pi - s * nrows;
        if (roi[s].active && rowcnt[s * wh_cap + ry] != 0)
            roi_row<true, VEC4>(mask + (size_t)(r0 + ry) * w, roi[s].a, roi[s].b, r0 + ry, pix + (size_t)s * maxpix,
                                n_out[s] + (int)rowoff[s * (wh_cap + 1) + ry], maxpix, mom[s], y0c, x0c);
    }
    for (int s = 0; s < 2; ++s) {   // uniform bookkeeping (every thread keeps the same copy)
        if (!roi[s].active) continue;
        int d = 0;
        for (int ry = 0; ry < nrows; ++ry) d += rowcnt[s * wh_cap + ry] != 0;
        distinct[s] += d;
        n_out[s] += (int)rowoff[s * (wh_cap + 1) + nrows];
    }
    __syncthreads();
}

// carry (LDS, 8 doubles, optional): the six coefficients of this fit and, in [6], 1.0 when the next frame of a chained band
// search may build on them (both lanes found, both fits regular) -- k_band_chain2
__device__ void reduce_and_fit(Moments* mom, const int* distinct, long long* s_mom, int h, int w, int n_left,
                               int n_right, bool detected, int mode, lt_lane_record* rec, int pix_format = 0,
                               double* carry = nullptr) {
    // block-reduce the per-thread moments: wave shuffle, then LDS atomics
    for (int i = threadIdx.x; i < 16; i += NT) s_mom[i] = 0;
    __syncthreads();
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long v = wave_sum_i64(mom[s].m[k]);
            if (lane_id() == 0 && v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&s_mom[s * 8 + k]), (unsigned long long)v);
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        lt_lane_record r;
        const double y0 = (double)(h / 2), x0 = (double)(w / 2), sy = (double)(h > 1 ? h : 2) * 0.5;
        unsigned flags = 0;
        for (int k = 0; k < 3; ++k) r.left_coeffs[k] = r.right_coeffs[k] = 0.0;
        if (detected) {
            if (!solve_poly2(s_mom, distinct[0], y0, x0, sy, r.left_coeffs)) flags |= 1u;
            if (!solve_poly2(s_mom + 8, distinct[1], y0, x0, sy, r.right_coeffs)) flags |= 2u;
        }
        r.n_left = n_left;
        r.n_right = n_right;
        r.detected = detected ? 1 : 0;
        r.fit_flags = (uint8_t)flags;
        r.mode = (uint8_t)mode;
        r._pad = (uint8_t)pix_format;   // 0: packed (y << 16) | x lists; 1: per-row column masks (k_sws_fit2)
        r.frame = rec->frame;  // keep the caller's tag
        *rec = r;
        if (carry) {
            for (int k = 0; k < 3; ++k) { carry[k] = r.left_coeffs[k]; carry[3 + k] = r.right_coeffs[k]; }
            carry[6] = detected && flags == 0 ? 1.0 : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
template <bool VEC4>
__global__ __launch_bounds__(NT) void k_sws_fit(const uint8_t* __restrict__ masks, size_t mask_stride, SearchGeom g,
                                               const uint32_t* __restrict__ band_sums,
                                               uint32_t* __restrict__ pix_all, int32_t* __restrict__ cent_all,
                                               lt_lane_record* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* sums = reinterpret_cast<unsigned*>(smem);          // w
    unsigned* prefix = sums + g.w;                               // w + 1
    unsigned* rowcnt = prefix + g.w + 1;                         // 2 * wh
    unsigned* rowoff = rowcnt + 2 * g.wh;                        // 2 * (wh + 1)
    long long* s_mom = reinterpret_cast<long long*>(smem + (((size_t)(2 * g.w + 1 + 4 * g.wh + 2) * 4 + 15) & ~(size_t)15));

    const int frame = blockIdx.x;
    const uint8_t* mask = masks + (size_t)frame * mask_stride;
    const uint32_t* fsums = band_sums + (size_t)frame * g.nbands * g.w;   // [band][w], band 0 = start slice
    uint32_t* pix = pix_all + (size_t)frame * 2 * g.maxpix;
    int32_t* cent = cent_all + (size_t)frame * 2 * (g.maxlev + 2);   // [side][0] = count, then entries
    const int W = g.w, ww = g.ww, wh = g.wh, hw = g.hw, H1 = g.img_height;
    const int y0c = g.h / 2, x0c = g.w / 2;

    Moments mom[2];
    mom[0].clear();
    mom[1].clear();
    int n_out[2] = {0, 0}, distinct[2] = {0, 0};
    // per-side search state, identical in every thread  (lane_tracker.py:335-343)
    int c[2], ns[2] = {0, 0}, lo[2], hi[2], ndiff[2] = {0, 0}, last_diff[2] = {0, 0}, ncent[2] = {0, 0},
        nroi[2] = {0, 0};
    Roi roi[2];

    // ---- level 0 (:290-332)
    for (int s = 0; s < 2; ++s) {
        const int c0 = s == 0 ? g.ignore_sides : g.img_center;
        const int c1 = s == 0 ? g.img_center : W - g.ignore_sides;
        const int off = c0;
        bool found = false;
        int first = 0, last = 0;
        if (c1 > c0 && H1 > g.y_start) {
            load_sums(fsums + c0, c1 - c0, sums);                    // :290 / :310
            wave0_exclusive_scan(sums, prefix, c1 - c0);
            __syncthreads();
            found = box_argmax(prefix, c1 - c0, ww, 0, (c1 - c0) + ww - 1, first, last);
            __syncthreads();
        }
        roi[s].active = 0;
        if (found) {
            c[s] = ((first + last) >> 1) - hw + off;                 // :296-297 / :316-317
            const int a = c[s] - hw, b = min(c[s] + hw, W);
            roi[s].active = 1;
            roi[s].a = a < 0 ? b : a;                                // negative start: NumPy slice is empty
            roi[s].b = b;
            nroi[s]++;
        } else {
            c[s] = s == 0 ? g.def_left : g.def_right;                // :308 / :328
        }
        if (threadIdx.x == 0) cent[s * (g.maxlev + 2) + 1 + ncent[s]] = c[s];
        ncent[s]++;
        lo[s] = -g.search_range;
        hi[s] = g.search_range;
    }
    if (H1 - wh >= 0)
        extract_windows<VEC4>(mask, W, H1 - wh, H1, roi, rowcnt, rowoff, wh, pix, g.maxpix, n_out, mom, distinct, y0c, x0c);

    // ---- levels 1 .. nlevels-1 (:346-430)
    for (int level = 1; level < g.nlevels; ++level) {
        const int r0 = H1 - (1 + level) * wh, r1 = H1 - level * wh;
        load_sums(fsums + (size_t)level * W, W, sums);               // :350
        wave0_exclusive_scan(sums, prefix, W);
        __syncthreads();
        const int conv_len = W + ww - 1;                             // :351
        for (int s = 0; s < 2; ++s) {                                // left first, then right (:354, :395)
            roi[s].active = 0;
            if (ns[s] >= g.limit) continue;
            const int lo_i = max(c[s] + lo[s] + hw, 0);              // :356
            const int hi_i = min(c[s] + hi[s] + hw, W);              // :357
            const int a = min(lo_i, conv_len);                       // conv[lo_i:hi_i], Python slice rules
            const int b = hi_i < 0 ? max(conv_len + hi_i, 0) : min(hi_i, conv_len);
            int first = 0, last = 0;
            const bool found = b > a && box_argmax(prefix, W, ww, a, b, first, last);   // :360
            if (found) {
                const int newc = ((first + last + 1) >> 1) + lo_i - hw;   // ceil, :363-364
                if (threadIdx.x == 0 && ncent[s] < g.maxlev + 1) cent[s * (g.maxlev + 2) + 1 + ncent[s]] = newc;
                ncent[s]++;
                last_diff[s] = newc - c[s];                          // :366
                ndiff[s]++;
                c[s] = newc;
                ns[s] = 0;                                           // :368
                const int ra = c[s] - hw, rb = min(c[s] + hw, W);
                roi[s].active = 1;
                roi[s].a = ra < 0 ? rb : ra;
                roi[s].b = rb;
                nroi[s]++;
                const int t = (int)(g.mu * (double)last_diff[s]);    // :380-381, truncation toward zero
                lo[s] += t;
                hi[s] += t;
            } else {
                const int o = 1 - s;
                if (ndiff[o] > 0 && ns[o] == 0) c[s] += last_diff[o]; // :385-387 / :423-425
                if (threadIdx.x == 0 && ncent[s] < g.maxlev + 1) cent[s * (g.maxlev + 2) + 1 + ncent[s]] = c[s];
                ncent[s]++;
                ns[s]++;                                             // :390
                if (ns[s] >= g.limit) ncent[s] -= min(g.limit > 0 ? g.limit : ncent[s], ncent[s]);   // :391-392
            }
        }
        __syncthreads();
        if (roi[0].active || roi[1].active)
            extract_windows<VEC4>(mask, W, r0, r1, roi, rowcnt, rowoff, wh, pix, g.maxpix, n_out, mom, distinct, y0c, x0c);
    }
    if (threadIdx.x == 0) {
        cent[0] = ncent[0];
        cent[g.maxlev + 2] = ncent[1];
    }
    const bool detected = nroi[0] > 0 && nroi[1] > 0 && n_out[0] > 0 && n_out[1] > 0;   // :432-447
    reduce_and_fit(mom, distinct, s_mom, g.h, g.w, n_out[0], n_out[1], detected, 0, recs + frame);
}

// ===================================================================================================
// k_sws_fit2: the same search with the dependent chain taken off global memory.
//
// k_sws_fit walks the levels one by one and pays, per level, a global round trip for the band sums, a
// 1080-wide scan, and two more dependent round trips for the count and emit passes of the windows --
// about 13 us per level, 340 us per frame, almost all of it latency.  Here
//   A  every band sum of the frame is brought into LDS at once (levels as u16; the start slice as u32);
//   B  wave 0 alone runs the recurrence over the levels on LDS data (a scan over just the columns a
//      level can reach, two argmax passes), recording each level's window [a, b) per side;
//   C  all windows are then read in parallel, coalesced, as 16-byte pieces dealt to the 256 threads; the
//      non-zero flags of a piece are OR-ed into one 64-bit column mask per window row (LDS atomics); a row's
//      count and moments follow from its mask in closed form, and the masks themselves are what is stored:
//      the reference's pixel lists are expanded from them on demand (lt_download_pixels).
// Needs w % 4 == 0, window width <= 64, window height * 255 <= 65535; launch_sws_fit falls back to k_sws_fit
// otherwise.

// n / d for 0 <= n < 2^31 / d with a precomputed multiplier (exact: n * (magic * d - 2^32) < 2^32)
__device__ __forceinline__ unsigned div_magic(int d) { return d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int div_by(int n, int d, unsigned magic) { return d > 1 ? (int)__umulhi((unsigned)n, magic) : n; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS pointers keep their address space through the out-of-line call (a generic pointer would turn every
// access into a flat_load / flat_store)
#define LT_LDS __attribute__((address_space(3)))
typedef LT_LDS unsigned lds_u32;
typedef LT_LDS const unsigned lds_cu32;
typedef LT_LDS const uint16_t lds_cu16;
typedef LT_LDS int lds_i32;
// ... and a GLOBAL pointer its own: the centroid list lives in global memory, and through a generic pointer every store to it was a
// flat_store -- which counts in lgkmcnt as well as vmcnt, so that each of the recurrence's waits for an LDS read also waited for the
// last centroid's round trip to L2 (39 of the one-frame kernel's 57 us were phase B; LT_SWS2_PROBE, NOTES_r06 E.8)
typedef __attribute__((address_space(1))) int32_t glb_i32;
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// first/last argmax of conv[a:b), conv = np.convolve(ones(ww), src[0:ncnt)) ('full').  Only the entries a
// level can reach are touched: the exclusive prefix of src[pl .. ph) is built 64 entries at a time (one
// entry per lane, DPP scan) into q, conv[k] = q[hi] - q[lo], one wave-max, and the positions of the maxima
// come from ballots.  q: scratch of >= ph - pl + 1 entries.  false <=> np.any false.  All arguments and
// results are wave-uniform.
template <class SRC>
__device__ __forceinline__ bool box_argmax_window(SRC src, int ncnt, int ww, int a, int b, lds_u32* q, int& first, int& last) {
    const int pl = max(a - ww + 1, 0), ph = min(b, ncnt);
    if (ph <= pl) return false;                      // every window sum is empty -> np.any false
    const int lane = lane_id(), n = ph - pl;
    unsigned run = 0;
    if (n <= 128) {                                  // the levels above the start slice (n = reach + window width ~ 70): ONE pass, two
        const int i0 = 2 * lane, i1 = i0 + 1;        // entries per lane -- a second pass of LDS read -> scan -> LDS write is 0.15 us of latency
        const unsigned v0 = i0 < n ? (unsigned)src[pl + i0] : 0u, v1 = i1 < n ? (unsigned)src[pl + i1] : 0u;
        const unsigned t = v0 + v1;
        const unsigned incl = wave_inclusive_sum_dpp(t);
        if (i0 < n) q[i0] = incl - t;
        if (i1 < n) q[i1] = incl - v1;
        run = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    } else
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const unsigned v = i < n ? (unsigned)src[pl + i] : 0u;
        const unsigned incl = wave_inclusive_sum_dpp(v);
        if (i < n) q[i] = run + incl - v;
        run += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) q[n] = run;
    if (run == 0) return false;                      // nothing in reach
    wave_sync();
    auto window = [&](int k) {
        const int lo = max(k - ww + 1, 0), hi = min(k + 1, ncnt);
        return (k < b && hi > lo) ? q[hi - pl] - q[lo - pl] : 0u;
    };
    int f, l;
    if (b - a <= 64) {                               // the usual case: one candidate per lane
        const unsigned v = window(a + lane);
        const unsigned best = wave_max_u32_dpp(v);
        const unsigned long long hit = __ballot(a + lane < b && v == best);
        f = a + (int)__builtin_ctzll(hit);
        l = a + 63 - (int)__builtin_clzll(hit);
    } else {
        unsigned best = 0;
        for (int base = a; base < b; base += 64) best = max(best, window(base + lane));
        best = wave_max_u32_dpp(best);
        f = -1;
        l = -1;
        for (int base = a; base < b; base += 64) {
            const unsigned long long hit = __ballot(base + lane < b && window(base + lane) == best);
            if (hit) {
                if (f < 0) f = base + (int)__builtin_ctzll(hit);
                l = base + 63 - (int)__builtin_clzll(hit);
            }
        }
    }
    wave_sync();                                     // q is rewritten by the next call
    first = f - a;
    last = l - a;
    return true;                                     // run > 0: some window in [a, b) holds a non-zero entry
}

// Phase B of k_sws_fit2: the serial part of sliding_window_search (lane_tracker.py:290-430), run by one
// wave on the LDS image of the band sums.  Kept out of line on purpose: inlined into the large unrolled
// kernel body, the build (hipcc, ROCm 7.2) returned 32 instead of 432 as the default centre of an undetected
// left line -- caught by the golden tests, and correct again with a printf next to the assignment, i.e. a
// code-generation problem, not a data race.  As a separate function it is correct and costs one call.
// The per-side state lives in scalar registers (every value is wave-uniform; readfirstlane says so).
#ifdef LT_CASE_SWS2_INLINE         // tools/toolchain_cases.sh: the inlined form, to check whether the case still exists
#define LT_SWS2_RECURRENCE_LINKAGE __forceinline__
#else
#define LT_SWS2_RECURRENCE_LINKAGE __noinline__
#endif
// Two waves since round 6: a level's two window searches (left, right: box_argmax_window, 0.75 us each -- a chain of LDS round trips
// and DPP scans on one wave, 39 of the one-frame kernel's 57 us: LT_SWS2_PROBE) do not depend on each other -- only the
// bookkeeping behind them does (the right side's no-hit branch reads what the left side has just become, :423-425).  Wave 0
// searches the left side and wave 1 the right side AT THE SAME TIME, each on a prefix scratch of its own; each hands its result
// {found, first, last} to the other through LDS behind a level tag, and BOTH then run the bookkeeping of both sides, left first,
// from the same two results: every state variable is identical in the two waves at every level, which is also why a wave always
// knows whether the other one will search (and publish) at all.  Wave 0 alone stores centroids and windows.
struct Sws2Result { int found, first, last; };
__device__ __forceinline__ void sws2_publish(lds_i32* box, int tag, Sws2Result r) {
    if (lane_id() == 0) {
        box[0] = r.found; box[1] = r.first; box[2] = r.last;
        __hip_atomic_store(box + 3, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (workgroup scope: LDS only -- a wider scope would write back and invalidate the vector caches at every level)
    }
}
__device__ __forceinline__ Sws2Result sws2_collect(lds_i32* box, int tag) {
    // (bounded: half a second of polling -- a result that never came would be a bug in the conditions above, and a wrong record
    // that the parity tests catch is better than a kernel nobody can stop)
    for (unsigned spins = 0; uniform(__hip_atomic_load(box + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != tag && spins < (1u << 23); ++spins) __builtin_amdgcn_s_sleep(1);
    Sws2Result r;
    r.found = uniform(box[0]); r.first = uniform(box[1]); r.last = uniform(box[2]);
    return r;
}
__device__ LT_SWS2_RECURRENCE_LINKAGE void sws2_recurrence(SearchGeom g, int nlev, lds_cu32* sum0, lds_u32* prefix, lds_cu16* lev,
                                             lds_i32* roi_ab, lds_i32* state, glb_i32* cent, int wv, lds_i32* boxes) {
    const int lane = lane_id();
    const int W = g.w, ww = g.ww, wh = g.wh, hw = g.hw, H1 = g.img_height;
    const bool writer = wv == 0;
    struct Side { int c, ns, lo, hi, ndiff, last_diff, ncent, nroi; };
    Side sd[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    auto set_roi = [&](int s, Side& me, int level) {
        const int ra = me.c - hw, rb = min(me.c + hw, W);
        if (writer && lane == 0) {
            roi_ab[(s * nlev + level) * 2] = ra < 0 ? rb : ra;       // negative start: NumPy slice is empty
            roi_ab[(s * nlev + level) * 2 + 1] = rb;
        }
        me.nroi++;
    };
    auto put_centroid = [&](int s, Side& me, int value) {
        if (writer && lane == 0 && me.ncent < g.maxlev + 1) cent[s * (g.maxlev + 2) + 1 + me.ncent] = value;
        me.ncent++;
    };
    // a side's result of a level: searched by its own wave, fetched by the other.  One box per (side, level), tag 0 = not there yet:
    // a wave whose partner has stopped searching runs ahead of it by any number of levels, and must not overwrite what the partner
    // has yet to read
    auto box_of = [&](int side, int level) { return boxes + (side * nlev + level) * 4; };
    {                                                                // level 0 (:290-332)
        const int s = wv;
        const int c0 = s == 0 ? g.ignore_sides : g.img_center;
        const int c1 = s == 0 ? g.img_center : W - g.ignore_sides;
        Sws2Result mine = {0, 0, 0};
        if (c1 > c0 && H1 > g.y_start) {
            int first = 0, last = 0;
            mine.found = box_argmax_window(sum0 + c0, c1 - c0, ww, 0, (c1 - c0) + ww - 1, prefix, first, last) ? 1 : 0;
            mine.first = first; mine.last = last;
        }
        sws2_publish(box_of(wv, 0), 1, mine);
        const Sws2Result theirs = sws2_collect(box_of(1 - wv, 0), 1);
        const Sws2Result res0 = wv == 0 ? mine : theirs, res1 = wv == 0 ? theirs : mine;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Side& me = sd[t];
            const Sws2Result r = t == 0 ? res0 : res1;
            const int d0 = t == 0 ? g.ignore_sides : g.img_center;
            if (r.found) {
                me.c = uniform(((r.first + r.last) >> 1) - hw + d0);    // :296-297 / :316-317
                if (H1 - wh >= 0) set_roi(t, me, 0);
                else me.nroi++;
            } else {
                me.c = t == 0 ? g.def_left : g.def_right;             // :308 / :328
            }
            put_centroid(t, me, me.c);
            me.lo = -g.search_range;
            me.hi = g.search_range;
        }
    }
    const int conv_len = W + ww - 1;                                 // :351
    for (int level = 1; level < g.nlevels; ++level) {                // :346-430
        lds_cu16* sums = lev + (size_t)(level - 1) * W;
        const int tag = level + 1;
        // this wave's own side first: its window search needs nothing of the other side's at this level
        // (no array is indexed by the wave number: a private array with a run-time index lives in scratch memory, a global round
        // trip per access -- the first version of this function was SLOWER than one wave for that, 52 against 39 us)
        Sws2Result mine = {0, 0, 0};
        const bool searched0 = sd[0].ns < g.limit, searched1 = sd[1].ns < g.limit;
        const int lo_i0 = max(sd[0].c + sd[0].lo + hw, 0), lo_i1 = max(sd[1].c + sd[1].lo + hw, 0);      // :356
        const bool my_search = wv == 0 ? searched0 : searched1, their_search = wv == 0 ? searched1 : searched0;
        if (my_search) {
            const int my_c = wv == 0 ? sd[0].c : sd[1].c, my_hi = wv == 0 ? sd[0].hi : sd[1].hi;
            const int lo_i = wv == 0 ? lo_i0 : lo_i1;
            const int hi_i = min(my_c + my_hi + hw, W);              // :357
            const int a = min(lo_i, conv_len);                       // conv[lo_i:hi_i], Python slice rules
            const int b = hi_i < 0 ? max(conv_len + hi_i, 0) : min(hi_i, conv_len);
            int first = 0, last = 0;
            mine.found = (b > a && box_argmax_window(sums, W, ww, a, b, prefix, first, last)) ? 1 : 0;   // :360
            mine.first = first; mine.last = last;
            sws2_publish(box_of(wv, level), tag, mine);
        }
        Sws2Result theirs = {0, 0, 0};
        if (their_search) theirs = sws2_collect(box_of(1 - wv, level), tag);
        const Sws2Result res0 = wv == 0 ? mine : theirs, res1 = wv == 0 ? theirs : mine;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                                // left first, then right (:354, :395)
            Side& me = sd[s];
            const Side& other = sd[1 - s];
            if (!(s == 0 ? searched0 : searched1)) continue;
            const Sws2Result r = s == 0 ? res0 : res1;
            if (r.found) {
                const int newc = uniform(((r.first + r.last + 1) >> 1) + (s == 0 ? lo_i0 : lo_i1) - hw);   // ceil, :363-364
                put_centroid(s, me, newc);
                me.last_diff = newc - me.c;                          // :366
                me.ndiff++;
                me.c = newc;
                me.ns = 0;                                           // :368
                set_roi(s, me, level);
                const int t = uniform((int)(g.mu * (double)me.last_diff));   // :380-381, truncation toward zero
                me.lo += t;
                me.hi += t;
            } else {
                if (other.ndiff > 0 && other.ns == 0) me.c += other.last_diff;   // :385-387 / :423-425
                put_centroid(s, me, me.c);
                me.ns++;                                             // :390
                if (me.ns >= g.limit) me.ncent -= min(g.limit > 0 ? g.limit : me.ncent, me.ncent);   // :391-392
            }
        }
    }
    if (writer && lane == 0) {
        cent[0] = sd[0].ncent;
        cent[g.maxlev + 2] = sd[1].ncent;
        state[0] = sd[0].nroi;
        state[1] = sd[1].nroi;
    }
}

struct Sws2Layout {       // byte offsets into dynamic LDS
    int sum0, prefix, boxes, lev, roi, rowbits, state, mom, total;
    int nlev;
};
__host__ __device__ inline Sws2Layout sws2_layout(const SearchGeom& g) {
    Sws2Layout L;
    L.nlev = g.nlevels > 1 ? g.nlevels : 1;
    int o = 0;
    auto take = [&o](int bytes) { const int at = o; o = (o + bytes + 15) & ~15; return at; };   // 16-byte aligned sections
    L.sum0 = take(g.w * 4);
    L.prefix = take(2 * (g.w + g.ww + 64) * 4);       // window sums of one level: up to w + ww - 1 entries, one scratch per searching wave
    L.boxes = take(2 * L.nlev * 4 * 4);              // the two waves' results of every level: {found, first, last, tag} (sws2_recurrence)
    L.lev = take((L.nlev - 1) * g.w * 2);
    L.roi = take(L.nlev * 2 * 2 * 4);                // (a, b) per (side, level)
    L.rowbits = take(2 * L.nlev * g.wh * 8);         // one 64-bit column mask per window row
    L.state = take(8 * 4);                           // nroi[2], distinct[2], n_out[2]
    L.mom = take(16 * 8);
    L.total = o;
    return L;
}

// -DLT_SWS2_PROBE: wave 0 of frame 0 prints the 100 MHz ticks between the phases (measurement builds only)
#ifdef LT_SWS2_PROBE
#define SWS2_DECL long long sws2_t[7];
#define SWS2_T(i) sws2_t[i] = wall_clock64();
#define SWS2_REPORT                                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0)                                                                     \
        printf("sws2 ticks: A %lld  B %lld  C-load %lld  C-flags %lld  C-rows %lld  fit %lld\n", sws2_t[1] - sws2_t[0],     \
               sws2_t[2] - sws2_t[1], sws2_t[3] - sws2_t[2], sws2_t[4] - sws2_t[3], sws2_t[5] - sws2_t[4], sws2_t[6] - sws2_t[5]);
#else
#define SWS2_DECL
#define SWS2_T(i)
#define SWS2_REPORT
#endif

template <int ND, bool BITS>   // ND dwords cover one window row of a u8 mask: 9 for widths <= 32, 17 for <= 64
__global__ __launch_bounds__(NT) void k_sws_fit2(const uint8_t* __restrict__ masks, size_t mask_stride, MaskBits mb, SearchGeom g,
                                                const uint32_t* __restrict__ band_sums, uint32_t* __restrict__ pix_all,
                                                int32_t* __restrict__ cent_all, lt_lane_record* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SWS2_DECL
    const Sws2Layout L = sws2_layout(g);
    unsigned* sum0 = reinterpret_cast<unsigned*>(smem + L.sum0);
    unsigned* prefix = reinterpret_cast<unsigned*>(smem + L.prefix);
    uint16_t* lev = reinterpret_cast<uint16_t*>(smem + L.lev);
    int* roi_ab = reinterpret_cast<int*>(smem + L.roi);            // [(s * nlev + level) * 2 + {0,1}]
    int* state = reinterpret_cast<int*>(smem + L.state);
    long long* s_mom = reinterpret_cast<long long*>(smem + L.mom);

    const int frame = blockIdx.x, lane = lane_id(), wv = wave_id();
    const uint8_t* mask = masks + (size_t)frame * mask_stride;
    const uint32_t* fsums = band_sums + (size_t)frame * g.nbands * g.w;   // [band][w], band 0 = start slice
    int32_t* cent = cent_all + (size_t)frame * 2 * (g.maxlev + 2);
    const int W = g.w, wh = g.wh, H1 = g.img_height, nlev = L.nlev;
    const int y0c = g.h / 2, x0c = g.w / 2;

    SWS2_T(0)
    // ---- A: band sums -> LDS ---------------------------------------------------------------------
    {
        const int nq0 = W >> 2, nq = (nlev - 1) * nq0;                 // uint4 groups (W % 4 == 0)
        const uint4* src = reinterpret_cast<const uint4*>(fsums);
        for (int i = threadIdx.x; i < nq0; i += NT) reinterpret_cast<uint4*>(sum0)[i] = src[i];
        const uint4* lsrc = src + nq0;
        for (int base = threadIdx.x; base < nq; base += NT * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lsrc[min(base + u * NT, nq - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + u * NT < nq)
                    reinterpret_cast<uint2*>(lev)[base + u * NT] = make_uint2(v[u].x | (v[u].y << 16), v[u].z | (v[u].w << 16));
        }
        for (int i = threadIdx.x; i < nlev * 4; i += NT) roi_ab[i] = 0;  // a = b = 0: no window
        for (int i = threadIdx.x; i < 2 * nlev * wh; i += NT) reinterpret_cast<unsigned long long*>(smem + L.rowbits)[i] = 0ull;
        if (threadIdx.x < 8) state[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < 2 * nlev * 4; i += NT) reinterpret_cast<int*>(smem + L.boxes)[i] = 0;   // (tag 0: nothing published)
        if (threadIdx.x < 16) s_mom[threadIdx.x] = 0;
    }
    __syncthreads();

    SWS2_T(1)
    // ---- B: the recurrence over the levels, waves 0 (left side) and 1 (right side), LDS only (lane_tracker.py:290-430) -----------
    if (wv < 2)
        sws2_recurrence(g, nlev, (lds_cu32*)sum0, (lds_u32*)(prefix + (size_t)wv * (g.w + g.ww + 64)), (lds_cu16*)lev, (lds_i32*)roi_ab,
                        (lds_i32*)state, (glb_i32*)cent, wv, (lds_i32*)(smem + L.boxes));
    __syncthreads();

    SWS2_T(2)
    // ---- C: all windows in parallel -----------------------------------------------------------------
    // Window row r = (s * nlev + level) * wh + ry.
    //  1. every row is read as NQ aligned 16-byte pieces; the pieces of all rows are dealt to the 256 threads
    //     round-robin, so consecutive lanes read consecutive pieces (a row per lane would touch 64 cache
    //     lines per load instruction) and LOADS_IN_FLIGHT of them are outstanding per thread;
    //  2. the 16 non-zero flags of a piece are OR-ed into the row's 64-bit column mask in LDS;
    //  3. the rows are dealt to the threads: count, moments in closed form, and the mask itself goes to global
    //     memory -- the pixel lists of the reference (level-major, row-major, ascending x) are expanded from
    //     the masks on demand (lt_download_pixels), so nothing is scattered from here.
    constexpr int NQ = ND == 9 ? 3 : 5, LOADS_IN_FLIGHT = 16;
    struct __attribute__((packed, aligned(4))) Piece { uint32_t w[4]; };
    unsigned long long* rowbits = reinterpret_cast<unsigned long long*>(smem + L.rowbits);
    uint32_t* hdr = pix_all + (size_t)frame * 2 * g.maxpix;
    unsigned long long* gmask = reinterpret_cast<unsigned long long*>(hdr + sws2_mask_offset(nlev));
    if (threadIdx.x == 0) { hdr[0] = (uint32_t)nlev; hdr[1] = (uint32_t)wh; hdr[2] = (uint32_t)H1; hdr[3] = 0u; }
    for (int i = threadIdx.x; i < nlev * 4; i += NT) hdr[4 + i] = (uint32_t)roi_ab[i];
    const int rows_total = 2 * nlev * wh, pieces = rows_total * NQ;
    const unsigned inv_wh = div_magic(wh);
    SWS2_T(3)
    if constexpr (BITS) {
        // bit-plane input: a window row is at most 64 columns = two consecutive words of its mask row, funnel-shifted
        const unsigned long long* fb = mb.bits + (size_t)frame * mb.bits_stride;
        for (int r = threadIdx.x; r < rows_total; r += NT) {
            const int sl = div_by(r, wh, inv_wh), ry = r - sl * wh, level = sl >= nlev ? sl - nlev : sl;
            const int a = roi_ab[sl * 2], b = roi_ab[sl * 2 + 1];
            const int y = H1 - (1 + level) * wh + ry, j0 = min(a >> 6, mb.wpr - 1), sh = a & 63;
            const unsigned long long w0 = fb[(size_t)y * mb.wpr + j0], w1 = fb[(size_t)y * mb.wpr + min(j0 + 1, mb.wpr - 1)];
            unsigned long long m = sh ? (w0 >> sh) | ((j0 + 1 < mb.wpr ? w1 : 0ull) << (64 - sh)) : w0;
            const int bw = b - a;
            m &= bw <= 0 ? 0ull : bw < 64 ? (1ull << bw) - 1ull : ~0ull;
            rowbits[r] = m;
        }
    } else {
        for (int g0 = 0; g0 < pieces; g0 += NT * LOADS_IN_FLIGHT) {
            Piece v[LOADS_IN_FLIGHT];
            int prow[LOADS_IN_FLIGHT], psh[LOADS_IN_FLIGHT], pbw[LOADS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; ++u) {           // loads only: no branch, clamped addresses
                const int gi = g0 + u * NT + (int)threadIdx.x, gc = min(gi, pieces - 1);
                const int r = gc / NQ, q = gc - r * NQ;
                const int sl = div_by(r, wh, inv_wh), ry = r - sl * wh, level = sl >= nlev ? sl - nlev : sl;
                const int a = roi_ab[sl * 2], b = roi_ab[sl * 2 + 1];
                const int y = H1 - (1 + level) * wh + ry;
                const int x = min(max((a & ~3) + 16 * q, 0), W - 16);
                v[u] = *reinterpret_cast<const Piece*>(mask + (size_t)y * W + x);
                prow[u] = r;
                psh[u] = x - a;                                    // bit index of the piece's first column
                pbw[u] = (gi < pieces && b > a) ? b - a : 0;       // 0: nothing to keep
            }
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; ++u) {
                uint32_t flags = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // high bit of every non-zero byte, then the four high bits gathered into a nibble
                    const uint32_t w4 = v[u].w[k];
                    const uint32_t hb = ((w4 | ((w4 & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;
                    flags |= ((hb * 0x00204081u) >> 21 & 0xfu) << (4 * k);
                }
                const int sh = psh[u];
                unsigned long long piece = sh >= 64 ? 0ull : sh >= 0 ? (unsigned long long)flags << sh : (unsigned long long)flags >> min(-sh, 63);
                piece &= pbw[u] < 64 ? (1ull << (pbw[u] & 63)) - 1ull : ~0ull;
                if (piece) atomicOr(&rowbits[prow[u]], piece);
            }
        }
    }
    __syncthreads();
    SWS2_T(4)
    Moments mom[2];
    mom[0].clear();
    mom[1].clear();
    unsigned n_rows[2] = {0, 0}, n_pix[2] = {0, 0};
    for (int r = threadIdx.x; r < rows_total; r += NT) {      // lane-per-row: counts and moments in closed form
        const int sl = div_by(r, wh, inv_wh), ry = r - sl * wh, s = sl >= nlev ? 1 : 0, level = sl - s * nlev;
        const unsigned long long m = rowbits[r];
        gmask[r] = m;
        // y is fixed, so only the count and the sum of the columns are needed;
        // sum of the set bit positions = sum_b 2^b popcount(m & {positions with bit b})
        const int a = roi_ab[sl * 2];
        const int cnt = __popcll(m);
        const int sj = __popcll(m & 0xaaaaaaaaaaaaaaaaull) + 2 * __popcll(m & 0xccccccccccccccccull) +
                       4 * __popcll(m & 0xf0f0f0f0f0f0f0f0ull) + 8 * __popcll(m & 0xff00ff00ff00ff00ull) +
                       16 * __popcll(m & 0xffff0000ffff0000ull) + 32 * __popcll(m & 0xffffffff00000000ull);
        // |dy| <= 8192, cnt <= 64, |sdx| < 2^22: the 32-bit products below cannot overflow
        const int dy = H1 - (1 + level) * wh + ry - y0c, dy2 = dy * dy, sdx = sj + cnt * (a - x0c), cdy2 = cnt * dy2;
        Moments& mm = s == 0 ? mom[0] : mom[1];
        mm.m[0] += cnt; mm.m[1] += cnt * dy; mm.m[2] += cdy2; mm.m[3] += (long long)cdy2 * dy; mm.m[4] += (long long)cdy2 * dy2;
        mm.m[5] += sdx; mm.m[6] += (long long)sdx * dy; mm.m[7] += (long long)sdx * dy2;
        n_pix[s] += (unsigned)cnt;
        n_rows[s] += cnt != 0 ? 1u : 0u;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned rows = wave_inclusive_sum_dpp(n_rows[s]), pixels = wave_inclusive_sum_dpp(n_pix[s]);
        if (lane == 63) {
            if (rows) atomicAdd(&state[2 + s], (int)rows);     // distinct rows per side
            if (pixels) atomicAdd(&state[4 + s], (int)pixels); // n_out per side
        }
    }
    __syncthreads();
    SWS2_T(5)
    const int n_left = state[4], n_right = state[5];
    const int distinct[2] = {state[2], state[3]};
    const bool detected = state[0] > 0 && state[1] > 0 && n_left > 0 && n_right > 0;   // :432-447
    reduce_and_fit(mom, distinct, s_mom, g.h, g.w, n_left, n_right, detected, 0, recs + frame, 1);
    SWS2_T(6)
    SWS2_REPORT
}

// ---------------------------------------------------------------------------------------------------
// Integer column range [a, b) of the pixels with lo < x < hi, clipped to [0, W).  x is an integer, so
// x > lo <=> x >= floor(lo)+1 and x < hi <=> x < ceil(hi): the same set the reference's f64
// comparisons select (:474-489).  NaN / empty bands give a >= b.
__device__ __forceinline__ void band_columns(double lo, double hi, int W, int& a, int& b) {
    a = 0;
    b = 0;
    if (!(lo < hi) || !(hi > 0.0) || !(lo < (double)(W - 1))) return;
    a = lo < 0.0 ? 0 : (int)floor(lo) + 1;
    b = hi > (double)W ? W : (int)ceil(hi);
    b = min(b, W);
}

template <bool VEC4>
__global__ __launch_bounds__(NT) void k_band_fit(const uint8_t* __restrict__ masks, size_t mask_stride, SearchGeom g,
                                                const double* __restrict__ prev, BandPrev bp, uint32_t* __restrict__ pix_all,
                                                lt_lane_record* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* rowcnt = reinterpret_cast<unsigned*>(smem);        // 2 * h
    unsigned* rowoff = rowcnt + 2 * g.h;                         // 2 * (h + 1)
    long long* s_mom = reinterpret_cast<long long*>(smem + (((size_t)(4 * g.h + 2 + 4) * 4 + 15) & ~(size_t)15));
    const int frame = blockIdx.x, lane = lane_id(), wv = wave_id();
    const uint8_t* mask = masks + (size_t)frame * mask_stride;
    uint32_t* pix = pix_all + (size_t)frame * 2 * g.maxpix;
    const double* pc = bp.by_value ? bp.c : prev + (size_t)frame * 6;
    const int W = g.w, top = g.band_top, bottom = g.band_bottom, nrows = max(bottom - top, 0);
    const int y0c = g.h / 2, x0c = g.w / 2;
    const double bw = g.bandwidth;
    Moments mom[2];
    mom[0].clear();
    mom[1].clear();

    // one thread per (side, row): the band of a row is the reference's f64 expression
    // ((a*y^2 + b*y) + c) -/+ bw, evaluated without FMA contraction
    auto columns = [&](int s, int y, int& a, int& b) {
        const double y2 = (double)((long long)y * y), yd = (double)y;
        const double t = pc[s * 3] * y2 + pc[s * 3 + 1] * yd + pc[s * 3 + 2];
        band_columns(t - bw, t + bw, W, a, b);
    };
    const int pairs = 2 * nrows;
    for (int pi = threadIdx.x; pi < pairs; pi += NT) {
        const int s = pi >= nrows ? 1 : 0, ry = pi - s * nrows, y = top + ry;
        int a, b;
        columns(s, y, a, b);
        rowcnt[s * g.h + ry] = b > a ? roi_row<false, VEC4>(mask + (size_t)y * W, a, b, y, nullptr, 0, 0, mom[s], y0c, x0c) : 0;
    }
    __syncthreads();
    if (wv < 2) {
        // exclusive scan over rows: wave 0 -> left, wave 1 -> right
        const int s = wv, chunk = (nrows + 63) / 64;
        const int a = min(lane * chunk, nrows), b = min(a + chunk, nrows);
        unsigned sum = 0;
        for (int i = a; i < b; ++i) sum += rowcnt[s * g.h + i];
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += t;
        }
        unsigned run = incl - sum;
        for (int i = a; i < b; ++i) {
            const unsigned v = rowcnt[s * g.h + i];
            rowoff[s * (g.h + 1) + i] = run;
            run += v;
        }
        if (lane == 63) rowoff[s * (g.h + 1) + nrows] = incl;
    }
    __syncthreads();
    // ordered write (row-major, like nonzero()): each (side, row) thread emits its own pixels
    for (int pi = threadIdx.x; pi < pairs; pi += NT) {
        const int s = pi >= nrows ? 1 : 0, ry = pi - s * nrows, y = top + ry;
        if (rowcnt[s * g.h + ry] == 0) continue;
        int a, b;
        columns(s, y, a, b);
        roi_row<true, VEC4>(mask + (size_t)y * W, a, b, y, pix + (size_t)s * g.maxpix, (int)rowoff[s * (g.h + 1) + ry],
                            g.maxpix, mom[s], y0c, x0c);
    }
    int distinct[2] = {0, 0};
    for (int s = 0; s < 2; ++s) {
        int d = 0;
        for (int i = threadIdx.x; i < nrows; i += NT) d += rowcnt[s * g.h + i] != 0;
        int tot = (int)wave_sum_i64(d);
        __syncthreads();
        if (lane == 0) rowoff[2 * (g.h + 1) + wv] = (unsigned)tot;   // 4 spare words reserved by the launcher
        __syncthreads();
        distinct[s] = (int)(rowoff[2 * (g.h + 1)] + rowoff[2 * (g.h + 1) + 1] + rowoff[2 * (g.h + 1) + 2] + rowoff[2 * (g.h + 1) + 3]);
    }
    const int nl = (int)rowoff[nrows], nr = (int)rowoff[(g.h + 1) + nrows];
    const bool detected = nl != 0 && nr != 0;                        // :491
    __syncthreads();
    reduce_and_fit(mom, distinct, s_mom, g.h, g.w, nl, nr, detected, 1, recs + frame);
}

// k_band_fit2: the band search with the same row-mask scheme as k_sws_fit2.  Every (side, row) has a column
// interval [a, b) of at most 64 pixels; its non-zero pixels are one 64-bit mask, built from coalesced 16-byte
// pieces dealt to the 256 threads; counts and moments follow from the masks in closed form; the masks (and the
// a of every row) are what is stored -- lt_download_pixels expands them (row-major, ascending x).
// Per-frame block in the pixel buffer (u32 units): [0] rows per side, [1] first row, [2] 0, [3] 0; then the
// a of every (side, row) as int32; then, 8-byte aligned, one u64 mask per (side, row).
template <bool BITS>
__device__ __forceinline__ void band_fit2_frame(unsigned char* smem, const int frame, const uint8_t* __restrict__ masks, size_t mask_stride,
                                                const MaskBits& mb, const SearchGeom& g, const double* pc,
                                                uint32_t* __restrict__ pix_all, lt_lane_record* __restrict__ recs, int nq,
                                                double* carry) {
    const int lane = lane_id();
    const int W = g.w, top = g.band_top, nrows = max(g.band_bottom - top, 0), rows_total = 2 * nrows;
    unsigned long long* rowbits = reinterpret_cast<unsigned long long*>(smem);          // rows_total
    int* row_a = reinterpret_cast<int*>(rowbits + rows_total);                          // rows_total
    int* row_w = row_a + rows_total;                                                    // rows_total
    int* state = row_w + rows_total;                                                    // distinct[2], n[2]
    long long* s_mom = reinterpret_cast<long long*>(smem + band2_mom_offset(nrows));
    const uint8_t* mask = masks + (size_t)frame * mask_stride;
    uint32_t* hdr = pix_all + (size_t)frame * 2 * g.maxpix;
    int32_t* g_a = reinterpret_cast<int32_t*>(hdr + 4);
    unsigned long long* gmask = reinterpret_cast<unsigned long long*>(hdr + band2_mask_offset(nrows));
    const int y0c = g.h / 2, x0c = g.w / 2;
    const double bw = g.bandwidth;
    if (threadIdx.x == 0) { hdr[0] = (uint32_t)nrows; hdr[1] = (uint32_t)top; hdr[2] = 0u; hdr[3] = 0u; }
    if (threadIdx.x < 4) state[threadIdx.x] = 0;
    if (threadIdx.x < 16) s_mom[threadIdx.x] = 0;
    // the band of a row is the reference's f64 expression ((a*y^2 + b*y) + c) -/+ bw, evaluated without FMA
    // contraction (lane_tracker.py:474-489)
    for (int r = threadIdx.x; r < rows_total; r += NT) {
        const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
        const double y2 = (double)((long long)y * y), yd = (double)y;
        const double t = pc[s * 3] * y2 + pc[s * 3 + 1] * yd + pc[s * 3 + 2];
        int a, b;
        band_columns(t - bw, t + bw, W, a, b);
        row_a[r] = a;
        row_w[r] = max(b - a, 0);
        g_a[r] = a;
        rowbits[r] = 0ull;
    }
    __syncthreads();
    if constexpr (BITS) {
        // bit-plane input: the band of a row is at most 64 columns = two consecutive words, funnel-shifted
        const unsigned long long* fb = mb.bits + (size_t)frame * mb.bits_stride;
        for (int r = threadIdx.x; r < rows_total; r += NT) {
            const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
            const int a = row_a[r], bw = row_w[r], j0 = min(a >> 6, mb.wpr - 1), sh = a & 63;
            const unsigned long long w0 = fb[(size_t)y * mb.wpr + j0], w1 = fb[(size_t)y * mb.wpr + min(j0 + 1, mb.wpr - 1)];
            unsigned long long m = sh ? (w0 >> sh) | ((j0 + 1 < mb.wpr ? w1 : 0ull) << (64 - sh)) : w0;
            m &= bw <= 0 ? 0ull : bw < 64 ? (1ull << bw) - 1ull : ~0ull;
            rowbits[r] = m;
        }
    } else {
        constexpr int LOADS_IN_FLIGHT = 16;
        struct __attribute__((packed, aligned(4))) Piece { uint32_t w[4]; };
        const int pieces = rows_total * nq;
        const unsigned inv_nq = div_magic(nq);
        for (int g0 = 0; g0 < pieces; g0 += NT * LOADS_IN_FLIGHT) {
            Piece v[LOADS_IN_FLIGHT];
            int prow[LOADS_IN_FLIGHT], psh[LOADS_IN_FLIGHT], pbw[LOADS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; ++u) {           // loads only: no branch, clamped addresses
                const int gi = g0 + u * NT + (int)threadIdx.x, gc = min(gi, pieces - 1);
                const int r = div_by(gc, nq, inv_nq), q = gc - r * nq;
                const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
                const int a = row_a[r];
                const int x = min(max((a & ~3) + 16 * q, 0), W - 16);
                v[u] = *reinterpret_cast<const Piece*>(mask + (size_t)y * W + x);
                prow[u] = r;
                psh[u] = x - a;
                pbw[u] = gi < pieces ? row_w[r] : 0;
            }
#pragma unroll
            for (int u = 0; u < LOADS_IN_FLIGHT; ++u) {
                uint32_t flags = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w4 = v[u].w[k];
                    const uint32_t hb = ((w4 | ((w4 & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;
                    flags |= ((hb * 0x00204081u) >> 21 & 0xfu) << (4 * k);
                }
                const int sh = psh[u];
                unsigned long long piece = sh >= 64 ? 0ull : sh >= 0 ? (unsigned long long)flags << sh : (unsigned long long)flags >> min(-sh, 63);
                piece &= pbw[u] < 64 ? (1ull << (pbw[u] & 63)) - 1ull : ~0ull;
                if (piece) atomicOr(&rowbits[prow[u]], piece);
            }
        }
    }
    __syncthreads();
    Moments mom[2];
    mom[0].clear();
    mom[1].clear();
    unsigned n_rows[2] = {0, 0}, n_pix[2] = {0, 0};
    for (int r = threadIdx.x; r < rows_total; r += NT) {
        const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
        const unsigned long long m = rowbits[r];
        gmask[r] = m;
        const int a = row_a[r];
        const int cnt = __popcll(m);
        const int sj = __popcll(m & 0xaaaaaaaaaaaaaaaaull) + 2 * __popcll(m & 0xccccccccccccccccull) +
                       4 * __popcll(m & 0xf0f0f0f0f0f0f0f0ull) + 8 * __popcll(m & 0xff00ff00ff00ff00ull) +
                       16 * __popcll(m & 0xffff0000ffff0000ull) + 32 * __popcll(m & 0xffffffff00000000ull);
        const int dy = y - y0c, dy2 = dy * dy, sdx = sj + cnt * (a - x0c), cdy2 = cnt * dy2;   // 32-bit safe: h <= 8192
        Moments& mm = s == 0 ? mom[0] : mom[1];
        mm.m[0] += cnt; mm.m[1] += cnt * dy; mm.m[2] += cdy2; mm.m[3] += (long long)cdy2 * dy; mm.m[4] += (long long)cdy2 * dy2;
        mm.m[5] += sdx; mm.m[6] += (long long)sdx * dy; mm.m[7] += (long long)sdx * dy2;
        n_pix[s] += (unsigned)cnt;
        n_rows[s] += cnt != 0 ? 1u : 0u;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned rows = wave_inclusive_sum_dpp(n_rows[s]), pixels = wave_inclusive_sum_dpp(n_pix[s]);
        if (lane == 63) {
            if (rows) atomicAdd(&state[s], (int)rows);
            if (pixels) atomicAdd(&state[2 + s], (int)pixels);
        }
    }
    __syncthreads();
    const int distinct[2] = {state[0], state[1]};
    const int nl = state[2], nr = state[3];
    const bool detected = nl != 0 && nr != 0;                        // :491
    reduce_and_fit(mom, distinct, s_mom, g.h, g.w, nl, nr, detected, 1, recs + frame, 2, carry);
}

template <bool BITS>
__global__ __launch_bounds__(NT) void k_band_fit2(const uint8_t* __restrict__ masks, size_t mask_stride, MaskBits mb, SearchGeom g,
                                                 const double* __restrict__ prev, BandPrev bp, uint32_t* __restrict__ pix_all,
                                                 lt_lane_record* __restrict__ recs, int nq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int frame = blockIdx.x;
    band_fit2_frame<BITS>(smem, frame, masks, mask_stride, mb, g, bp.by_value ? bp.c : prev + (size_t)frame * 6, pix_all, recs, nq, nullptr);
}

// k_band_chain2: the warm path of ONE stateful stream (lane_tracker.py:851-872 with last_detection <= n_reset on every
// frame): frame k+1's band is drawn around frame k's fit (:474-489 read last_left_coeffs / last_right_coeffs, which a valid
// frame k sets to its own raw fit, :1182-1183).  The frames are sequentially dependent through those six numbers only, so one
// workgroup walks the n resident masks in order and hands the coefficients on through LDS -- no host round trip, no launch
// per frame.  Whether frame k was VALID (check_validity, :561-627) is decided on the host afterwards; the host keeps the
// records up to the first frame it rejects and discards the rest, so this is speculation, never a change of results.  The
// walk stops by itself at a frame without both lanes or with a rank-deficient fit (nothing can build on it): the records of
// the remaining slots get detected = 0, mode = 255 ("not searched").
template <bool BITS>
__global__ __launch_bounds__(NT) void k_band_chain2(const uint8_t* __restrict__ masks, size_t mask_stride, MaskBits mb, SearchGeom g,
                                                   const lt_lane_record* __restrict__ seed_rec, BandPrev seed,
                                                   uint32_t* __restrict__ pix_all, lt_lane_record* __restrict__ recs, int nq, int n,
                                                   const int* cancel_epoch, int my_epoch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double carry[8];
    __shared__ int s_stop;
    // One workgroup, serial in the frames, usually sharing its CU with waves of the mask chain of later frames: whatever it
    // loses in issue arbitration lengthens every frame of the stream, so its four waves ask for the highest wave priority.
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) s_stop = 0;
    if (threadIdx.x < 6)
        carry[threadIdx.x] = seed.by_value ? seed.c[threadIdx.x]
                                           : (threadIdx.x < 3 ? seed_rec->left_coeffs[threadIdx.x] : seed_rec->right_coeffs[threadIdx.x - 3]);
    if (threadIdx.x == 6) carry[6] = seed.by_value || (seed_rec->detected && seed_rec->fit_flags == 0) ? 1.0 : 0.0;
    __syncthreads();
    int f = 0;
    for (; f < n && carry[6] != 0.0; ++f) {
        // the coefficients the frame reads must not change under it: copy them out of the carry slot first
        __shared__ double pc[6];
        if (threadIdx.x < 6) pc[threadIdx.x] = carry[threadIdx.x];
        __syncthreads();
        if (s_stop) break;
        // "the host gave up on this speculation" (lt_band_fit_chain_cancel) is a page-locked host word: the read crosses the
        // bus (~2 us), so it is issued here and looked at after the frame -- a cancelled chain runs one frame further
        int epoch_now = 0;
        if (threadIdx.x == 6) epoch_now = __atomic_load_n(cancel_epoch, __ATOMIC_RELAXED);
        band_fit2_frame<BITS>(smem, f, masks, mask_stride, mb, g, pc, pix_all, recs, nq, carry);
        if (threadIdx.x == 6) s_stop = epoch_now > my_epoch ? 1 : 0;
        __syncthreads();
    }
    for (int i = f + (int)threadIdx.x; i < n; i += NT) {
        lt_lane_record r;
        for (int k = 0; k < 3; ++k) r.left_coeffs[k] = r.right_coeffs[k] = 0.0;
        r.n_left = r.n_right = 0;
        r.detected = 0; r.fit_flags = 0; r.mode = 255; r._pad = 0;
        r.frame = recs[i].frame;
        recs[i] = r;
    }
}

// k_band_chain3: the same walk as k_band_chain2 (bit-plane masks only), reorganised around what a frame costs when ONE
// workgroup does it alone -- latency, barriers and the serial solve -- because in a stream this kernel, not the mask chain,
// sets the pace (17.6 us per frame beside the mask kernels of later frames against 16.3 us for the bus):
//   * one pass per row with no LDS staging: band interval, the two bit-plane words (every load of a thread issued before
//     the first is used), funnel shift, closed-form moments, and the row's `a` / mask straight to the pixel block;
//   * CT = 512 threads (five rows each at 1100 x 1080) instead of 256;
//   * the 18 partial sums of a thread (8 moments per side as int64, pixel and row counts packed) go to LDS value-major and
//     each wave folds a few of the 18 values (8 reads per lane + ONE wave sum per value) -- 16 wave sums of 64-bit values per
//     wave were the largest single piece of k_band_fit2's reduction;
//   * the two Cholesky solves (a dozen dependent f64 divisions and square roots each) run side by side on two waves.
// Records, pixel blocks and the stop / cancel behaviour are identical to k_band_chain2 (tests/test_gpu_chain.py holds both
// against the frame-by-frame band search).
constexpr int CT = 512;           // threads of k_band_chain3
constexpr int C3_VALUES = 18;     // per side: 8 moments + (rows << 32 | pixels)
__global__ __launch_bounds__(CT) void k_band_chain3(MaskBits mb, SearchGeom g, const lt_lane_record* __restrict__ seed_rec, BandPrev seed,
                                                   uint32_t* __restrict__ pix_all, lt_lane_record* __restrict__ recs, int n,
                                                   const int* cancel_epoch, int my_epoch, int prio, int ablate_arg,
                                                   lt_lane_record* __restrict__ mirror, unsigned mirror_ticket) {
    // mirror (a chain of one, launch_band_fit_one): page-locked host memory that gets a copy of the record and, behind it (a
    // system-scope fence in between), the launch's ticket -- the host polls that word and has the record the moment it is
    // written, without waiting for the end-of-kernel interrupt (tools/microbench/sync_latency.hip)
    // Timing probes of tools/stream_interference.py, in a build with -DLT_CHAIN_PROBES only (WRONG results): ablate bit 1 no
    // pixel-block stores, 2 no bit-plane loads, 4 no f64 band evaluation, 8 no solve, 16 no reduction; prio = wave priority.
    // The product build compiles them out (ablate is the constant 0, the priority the constant 3).
#ifdef LT_CHAIN_PROBES
    const int ablate = ablate_arg;
#else
    constexpr int ablate = 0;
    prio = 3;
    (void)ablate_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long long* part = reinterpret_cast<long long*>(smem);                  // [C3_VALUES][CT]
    __shared__ long long s_tot[C3_VALUES];
    __shared__ double carry[8], pc[6], s_fit[6];
    __shared__ int s_stop, s_ok[2];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const int W = g.w, top = g.band_top, nrows = max(g.band_bottom - top, 0), rows_total = 2 * nrows;
    const int y0c = g.h / 2, x0c = g.w / 2;
    const double bw = g.bandwidth;
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (threadIdx.x < 6)
        carry[threadIdx.x] = seed.by_value ? seed.c[threadIdx.x]
                                           : (threadIdx.x < 3 ? seed_rec->left_coeffs[threadIdx.x] : seed_rec->right_coeffs[threadIdx.x - 3]);
    if (threadIdx.x == 6) carry[6] = seed.by_value || (seed_rec->detected && seed_rec->fit_flags == 0) ? 1.0 : 0.0;
    if (threadIdx.x == 0) s_stop = 0;
    __syncthreads();
    int f = 0;
    for (; f < n && carry[6] != 0.0; ++f) {
        if (threadIdx.x < 6) pc[threadIdx.x] = carry[threadIdx.x];
        __syncthreads();
        if (s_stop) break;
        int epoch_now = 0;
        if (threadIdx.x == 6) epoch_now = __atomic_load_n(cancel_epoch, __ATOMIC_RELAXED);   // crosses the bus: looked at after the frame
        uint32_t* hdr = pix_all + (size_t)f * 2 * g.maxpix;
        int32_t* g_a = reinterpret_cast<int32_t*>(hdr + 4);
        unsigned long long* gmask = reinterpret_cast<unsigned long long*>(hdr + band2_mask_offset(nrows));
        const unsigned long long* fb = mb.bits + (size_t)f * mb.bits_stride;
        if (threadIdx.x == 0) { hdr[0] = (uint32_t)nrows; hdr[1] = (uint32_t)top; hdr[2] = 0u; hdr[3] = 0u; }
        Moments mom[2];
        mom[0].clear();
        mom[1].clear();
        unsigned n_rows[2] = {0, 0}, n_pix[2] = {0, 0};
        constexpr int U = 5;          // rows per thread and batch: 2140 rows / 512 threads
        for (int r0 = (int)threadIdx.x; r0 < rows_total; r0 += CT * U) {
            int ra[U], rw[U];
            unsigned long long w0[U], w1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                  // intervals and loads of the whole batch first
                const int r = min(r0 + u * CT, rows_total - 1);
                const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
                int a, b;
                if (ablate & 4) { a = 400 + 200 * s + (y & 7); b = a + 50; }
                else {
                    const double y2 = (double)((long long)y * y), yd = (double)y;
                    const double t = pc[s * 3] * y2 + pc[s * 3 + 1] * yd + pc[s * 3 + 2];   // ((a y^2 + b y) + c), no FMA (:474-489)
                    band_columns(t - bw, t + bw, W, a, b);
                }
                ra[u] = a;
                rw[u] = max(b - a, 0);
                const int j0 = min(a >> 6, mb.wpr - 1);
                if (ablate & 2) { w0[u] = 0x0000ffff00000000ull >> (y & 15); w1[u] = 0; }
                else {
                    w0[u] = fb[(size_t)y * mb.wpr + j0];
                    w1[u] = j0 + 1 < mb.wpr ? fb[(size_t)y * mb.wpr + j0 + 1] : 0ull;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * CT;
                if (r >= rows_total) break;
                const int s = r >= nrows ? 1 : 0, y = top + r - s * nrows;
                const int a = ra[u], bwid = rw[u], sh = a & 63;
                unsigned long long m = sh ? (w0[u] >> sh) | (w1[u] << (64 - sh)) : w0[u];
                m &= bwid <= 0 ? 0ull : bwid < 64 ? (1ull << bwid) - 1ull : ~0ull;
                if (!(ablate & 1)) { g_a[r] = a; gmask[r] = m; }
                const int cnt = __popcll(m);
                const int sj = __popcll(m & 0xaaaaaaaaaaaaaaaaull) + 2 * __popcll(m & 0xccccccccccccccccull) +
                               4 * __popcll(m & 0xf0f0f0f0f0f0f0f0ull) + 8 * __popcll(m & 0xff00ff00ff00ff00ull) +
                               16 * __popcll(m & 0xffff0000ffff0000ull) + 32 * __popcll(m & 0xffffffff00000000ull);
                const int dy = y - y0c, dy2 = dy * dy, sdx = sj + cnt * (a - x0c), cdy2 = cnt * dy2;   // 32-bit safe: h <= 8192
                Moments& mm = s == 0 ? mom[0] : mom[1];
                mm.m[0] += cnt; mm.m[1] += cnt * dy; mm.m[2] += cdy2; mm.m[3] += (long long)cdy2 * dy; mm.m[4] += (long long)cdy2 * dy2;
                mm.m[5] += sdx; mm.m[6] += (long long)sdx * dy; mm.m[7] += (long long)sdx * dy2;
                n_pix[s] += (unsigned)cnt;
                n_rows[s] += cnt != 0 ? 1u : 0u;
            }
        }
        // stage 1: every thread's 18 partial sums, value-major
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int k = 0; k < 8; ++k) part[(s * 9 + k) * CT + threadIdx.x] = mom[s].m[k];
            part[(s * 9 + 8) * CT + threadIdx.x] = (long long)(((unsigned long long)n_rows[s] << 32) | n_pix[s]);
        }
        __syncthreads();
        // stage 2: wave w folds values w, w + 8, w + 16
        for (int v = wv; v < C3_VALUES && !(ablate & 16); v += CT / 64) {
            long long acc = 0;
#pragma unroll
            for (int i = 0; i < CT / 64; ++i) acc += part[v * CT + i * 64 + lane];
            const long long tot = wave_sum_i64(acc);
            if (lane == 0) s_tot[v] = tot;
        }
        __syncthreads();
        // the two fits side by side: wave 0 lane 0 left, wave 1 lane 0 right
        const int nl = (int)(unsigned)(s_tot[8] & 0xffffffffll), nr = (int)(unsigned)(s_tot[17] & 0xffffffffll);
        const bool detected = nl != 0 && nr != 0;                            // :491
        if (lane == 0 && wv < 2) {
            const double y0 = (double)(g.h / 2), x0 = (double)(g.w / 2), sy = (double)(g.h > 1 ? g.h : 2) * 0.5;
            double c[3] = {0.0, 0.0, 0.0};
            bool ok = true;
            if (detected && !(ablate & 8)) ok = solve_poly2(s_tot + wv * 9, (int)(s_tot[wv * 9 + 8] >> 32), y0, x0, sy, c);
            if (ablate & 8) { c[0] = pc[wv * 3]; c[1] = pc[wv * 3 + 1]; c[2] = pc[wv * 3 + 2]; }
            s_fit[wv * 3] = c[0]; s_fit[wv * 3 + 1] = c[1]; s_fit[wv * 3 + 2] = c[2];
            s_ok[wv] = ok ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            lt_lane_record r;
            const unsigned flags = detected ? ((s_ok[0] ? 0u : 1u) | (s_ok[1] ? 0u : 2u)) : 0u;
            for (int k = 0; k < 3; ++k) { r.left_coeffs[k] = detected ? s_fit[k] : 0.0; r.right_coeffs[k] = detected ? s_fit[3 + k] : 0.0; }
            if (detected && (flags & 1u)) for (int k = 0; k < 3; ++k) r.left_coeffs[k] = 0.0;
            if (detected && (flags & 2u)) for (int k = 0; k < 3; ++k) r.right_coeffs[k] = 0.0;
            r.n_left = nl;
            r.n_right = nr;
            r.detected = detected ? 1 : 0;
            r.fit_flags = (uint8_t)flags;
            r.mode = 1;
            r._pad = 2;               // per-row column masks (k_band_fit2's format)
            r.frame = recs[f].frame;  // keep the caller's tag
            recs[f] = r;
            if (mirror) {
                mirror[f] = r;
                __threadfence_system();
                *reinterpret_cast<volatile unsigned*>(mirror + 1) = mirror_ticket;
            }
            for (int k = 0; k < 3; ++k) { carry[k] = r.left_coeffs[k]; carry[3 + k] = r.right_coeffs[k]; }
            carry[6] = detected && flags == 0 ? 1.0 : 0.0;
        }
        if (threadIdx.x == 6) s_stop = epoch_now > my_epoch ? 1 : 0;
        __syncthreads();
    }
    for (int i = f + (int)threadIdx.x; i < n; i += CT) {
        lt_lane_record r;
        for (int k = 0; k < 3; ++k) r.left_coeffs[k] = r.right_coeffs[k] = 0.0;
        r.n_left = r.n_right = 0;
        r.detected = 0; r.fit_flags = 0; r.mode = 255; r._pad = 0;
        r.frame = recs[i].frame;
        recs[i] = r;
    }
}

// fit_poly() on an explicit pixel list: moments by all threads, one Cholesky solve
__global__ __launch_bounds__(NT) void k_fit_list(const uint32_t* __restrict__ pix, int n, int h, int w,
                                                double* __restrict__ out4) {
    __shared__ long long s_mom[8];
    __shared__ int s_ymin, s_ymax, s_mid;
    if (threadIdx.x < 8) s_mom[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_ymin = 0x7fffffff; s_ymax = -1; s_mid = 0; }
    __syncthreads();
    Moments m;
    m.clear();
    int ymin = 0x7fffffff, ymax = -1;
    for (int i = threadIdx.x; i < n; i += NT) {
        const uint32_t p = pix[i];
        const int y = (int)(p >> 16), x = (int)(p & 0xffffu);
        m.add(y, x, h / 2, w / 2);
        ymin = min(ymin, y);
        ymax = max(ymax, y);
    }
    ymin = wave_min_i32(ymin);
    ymax = wave_max_i32(ymax);
    if (lane_id() == 0) { atomicMin(&s_ymin, ymin); atomicMax(&s_ymax, ymax); }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long v = wave_sum_i64(m.m[k]);
        if (lane_id() == 0 && v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&s_mom[k]), (unsigned long long)v);
    }
    __syncthreads();
    // a third distinct y exists iff some y is strictly between the extremes
    int mid = 0;
    for (int i = threadIdx.x; i < n; i += NT) {
        const int y = (int)(pix[i] >> 16);
        mid |= (y != s_ymin && y != s_ymax);
    }
    if (mid) atomicOr(&s_mid, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int distinct = n <= 0 ? 0 : (s_ymin == s_ymax ? 1 : (s_mid ? 3 : 2));
        double c[3];
        const bool ok = solve_poly2(s_mom, distinct, (double)(h / 2), (double)(w / 2), (double)(h > 1 ? h : 2) * 0.5, c);
        out4[0] = c[0]; out4[1] = c[1]; out4[2] = c[2];
        out4[3] = ok ? 0.0 : 1.0;
    }
}

}  // namespace

void launch_fit_list(hipStream_t s, const uint32_t* pix, int n, int h, int w, double* out4) {
    hipLaunchKernelGGL(k_fit_list, dim3(1), dim3(NT), 0, s, pix, n, h, w, out4);
}

namespace {
#define env_flag(name) ([] { const char* e = LT_EXP_ENV(name); return e && e[0] == '1'; }())      // (a macro: in the release build the name never reaches the binary)

template <class K>
bool allow_big_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
}
bool sws2_big_lds() {
    static const bool ok = allow_big_lds(k_sws_fit2<9, false>) && allow_big_lds(k_sws_fit2<17, false>) &&
                           allow_big_lds(k_sws_fit2<9, true>) && allow_big_lds(k_sws_fit2<17, true>);
    return ok;
}
bool band2_big_lds() {
    static const bool ok = allow_big_lds(k_band_fit2<false>) && allow_big_lds(k_band_fit2<true>) &&
                           allow_big_lds(k_band_chain2<false>) && allow_big_lds(k_band_chain2<true>);
    return ok;
}

// k_sws_fit2: dword rows, window width <= 64 bits, level sums that fit u16, 32-bit row moments and an LDS image of
// all band sums; anything else takes the level-by-level kernel
bool sws2_eligible(const SearchGeom& g, size_t mask_stride) {
    static const bool v1 = env_flag("LT_SWS_V1");
    const bool vec4 = (g.w & 3) == 0 && (mask_stride & 3) == 0;
    const Sws2Layout L = sws2_layout(g);
    return !v1 && vec4 && 2 * g.hw <= 64 && g.wh * 255 <= 65535 && g.h <= 8192 && g.w >= 16 && L.total <= 150 * 1024 &&
           g.img_height - g.wh >= 0 && g.nlevels * g.wh <= g.img_height && sws2_block_words(L.nlev, g.wh) <= (long long)g.maxpix &&
           (L.total <= 48 * 1024 || sws2_big_lds());
}

// k_band_fit2: a band of at most 64 columns (2 * bandwidth + 2), dword rows, 32-bit row moments
bool band2_eligible(const SearchGeom& g, size_t mask_stride) {
    static const bool v1 = env_flag("LT_BAND_V1");
    const bool vec4 = (g.w & 3) == 0 && (mask_stride & 3) == 0;
    const int nrows = std::max(g.band_bottom - g.band_top, 0);
    const long long width = 2LL * (long long)g.bandwidth + 2;
    const size_t lds = band2_mom_offset(nrows) + 16 * sizeof(long long);
    return !v1 && vec4 && width <= 64 && g.h <= 8192 && g.w >= 16 && lds <= 150 * 1024 && band2_block_words(nrows) <= (long long)g.maxpix &&
           (lds <= 48 * 1024 || band2_big_lds());
}

}  // namespace

bool sws_fit_takes_bits(const SearchGeom& g, size_t mask_stride) { return sws2_eligible(g, mask_stride) && !env_flag("LT_SEARCH_U8"); }
bool band_fit_takes_bits(const SearchGeom& g, size_t mask_stride) { return band2_eligible(g, mask_stride) && !env_flag("LT_SEARCH_U8"); }

void launch_sws_fit(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, uint32_t* band_sums,
                    uint32_t* pix, int32_t* cent, lt_lane_record* rec, int n) {
    if (n <= 0) return;
    const bool vec4 = (g.w & 3) == 0 && (mask_stride & 3) == 0, v2 = sws2_eligible(g, mask_stride);
    if (mb.bits && v2) {
        const size_t lds = (size_t)BSB_ROWS * mb.wpr * sizeof(unsigned long long);   // <= 128 KB at w = 4096
        static const bool big = allow_big_lds(k_band_sums_bits);
        (void)big;
        hipLaunchKernelGGL(k_band_sums_bits, dim3(1, g.nbands, n), dim3(BSB_NT), lds, s, mb, g, band_sums);
    } else {
        dim3 sgrid(((vec4 ? g.w / 4 : g.w) + 63) / 64, g.nbands, n);
        if (vec4) hipLaunchKernelGGL(k_band_sums<true>, sgrid, dim3(64, BS_RG), 0, s, masks, mask_stride, g, band_sums);
        else hipLaunchKernelGGL(k_band_sums<false>, sgrid, dim3(64, BS_RG), 0, s, masks, mask_stride, g, band_sums);
    }
    if (v2) {
        const Sws2Layout L = sws2_layout(g);
        {
            const bool narrow = 2 * g.hw <= 32;
            const size_t lds = (size_t)L.total;
#define LT_LAUNCH_SWS2(ND_, BITS_) \
    hipLaunchKernelGGL((k_sws_fit2<ND_, BITS_>), dim3(n), dim3(NT), lds, s, masks, mask_stride, mb, g, band_sums, pix, cent, rec)
            if (mb.bits) { if (narrow) LT_LAUNCH_SWS2(9, true); else LT_LAUNCH_SWS2(17, true); }
            else { if (narrow) LT_LAUNCH_SWS2(9, false); else LT_LAUNCH_SWS2(17, false); }
#undef LT_LAUNCH_SWS2
            return;
        }
    }
    const size_t words = (size_t)(2 * g.w + 1 + 4 * g.wh + 2);
    const size_t lds = ((words * 4 + 15) & ~(size_t)15) + 16 * sizeof(long long);
    if (vec4) hipLaunchKernelGGL(k_sws_fit<true>, dim3(n), dim3(NT), lds, s, masks, mask_stride, g, band_sums, pix, cent, rec);
    else hipLaunchKernelGGL(k_sws_fit<false>, dim3(n), dim3(NT), lds, s, masks, mask_stride, g, band_sums, pix, cent, rec);
}

void launch_band_fit(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, const double* prev,
                     const BandPrev& bp, uint32_t* pix, lt_lane_record* rec, int n) {
    if (n <= 0) return;
    const bool vec4 = (g.w & 3) == 0 && (mask_stride & 3) == 0;
    if (band2_eligible(g, mask_stride)) {
        const int nrows = std::max(g.band_bottom - g.band_top, 0);
        const size_t lds2 = band2_mom_offset(nrows) + 16 * sizeof(long long);
        {
            const int nq = (int)((2LL * (long long)g.bandwidth + 2 + 3 + 15) / 16);   // 16-byte pieces covering a row's band from (a & ~3)
            if (mb.bits) hipLaunchKernelGGL(k_band_fit2<true>, dim3(n), dim3(NT), lds2, s, masks, mask_stride, mb, g, prev, bp, pix, rec, nq);
            else hipLaunchKernelGGL(k_band_fit2<false>, dim3(n), dim3(NT), lds2, s, masks, mask_stride, mb, g, prev, bp, pix, rec, nq);
            return;
        }
    }
    const size_t words = (size_t)(4 * g.h + 2) + 4;  // + 4 words for the distinct-row reduction
    const size_t lds = ((words * 4 + 15) & ~(size_t)15) + 16 * sizeof(long long);
    if (vec4)
        hipLaunchKernelGGL(k_band_fit<true>, dim3(n), dim3(NT), lds, s, masks, mask_stride, g, prev, bp, pix, rec);
    else
        hipLaunchKernelGGL(k_band_fit<false>, dim3(n), dim3(NT), lds, s, masks, mask_stride, g, prev, bp, pix, rec);
}

bool band_chain_supported(const SearchGeom& g, size_t mask_stride) { return band2_eligible(g, mask_stride); }

// The band search of ONE frame around coefficients given by value, by the chain kernel with a chain of one: k_band_chain3 is
// built around the latency of a single workgroup (8 us per frame against k_band_fit2's 22 for one frame; record and pixel
// block are the same, tests/test_gpu_chain.py), which is what process() -- one frame per call, the host waiting -- pays for.
// `zero` = any readable device word (the chain's cancel flag; a chain of one never looks at it again); `mirror` = nullptr or
// device-visible page-locked memory for a copy of the record.  false: not launched.
bool launch_band_fit_one(hipStream_t s, MaskBits mb, SearchGeom g, const BandPrev& bp, uint32_t* pix, lt_lane_record* rec,
                         size_t mask_stride, const int* zero, lt_lane_record* mirror, unsigned mirror_ticket) {
    static const bool big3 = allow_big_lds(k_band_chain3);
    if (!mb.bits || !bp.by_value || !big3 || g.h > 8192 || !band2_eligible(g, mask_stride)) return false;
    const size_t lds3 = (size_t)C3_VALUES * CT * sizeof(long long);
    hipLaunchKernelGGL(k_band_chain3, dim3(1), dim3(CT), lds3, s, mb, g, (const lt_lane_record*)nullptr, bp, pix, rec, 1, zero, 0x7fffffff,
                       3, 0, mirror, mirror_ticket);
    return true;
}

void launch_band_chain(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, const lt_lane_record* seed_rec,
                       const BandPrev& seed, uint32_t* pix, lt_lane_record* rec, int n, const int* cancel_epoch, int my_epoch) {
    if (n <= 0) return;
    const int nrows = std::max(g.band_bottom - g.band_top, 0);
    const size_t lds2 = band2_mom_offset(nrows) + 16 * sizeof(long long);
    const int nq = (int)((2LL * (long long)g.bandwidth + 2 + 3 + 15) / 16);
    static const bool v2 = env_flag("LT_CHAIN_V2");      // A/B: the first chained kernel (the body of k_band_fit2 in a loop)
    const size_t lds3 = (size_t)C3_VALUES * CT * sizeof(long long);
    static const bool big3 = allow_big_lds(k_band_chain3);
    if (mb.bits && !v2 && big3 && g.h <= 8192) {
#ifdef LT_CHAIN_PROBES
        static const int prio = [] { const char* e = LT_EXP_ENV("LT_CHAIN_PRIO"); return e ? std::atoi(e) : 3; }();
        static const int ablate = [] { const char* e = LT_EXP_ENV("LT_CHAIN_ABLATE"); return e ? std::atoi(e) : 0; }();
#else
        constexpr int prio = 3, ablate = 0;
#endif
        hipLaunchKernelGGL(k_band_chain3, dim3(1), dim3(CT), lds3, s, mb, g, seed_rec, seed, pix, rec, n, cancel_epoch, my_epoch, prio, ablate,
                           (lt_lane_record*)nullptr, 0u);
        return;
    }
    if (mb.bits) hipLaunchKernelGGL(k_band_chain2<true>, dim3(1), dim3(NT), lds2, s, masks, mask_stride, mb, g, seed_rec, seed, pix, rec, nq, n, cancel_epoch, my_epoch);
    else hipLaunchKernelGGL(k_band_chain2<false>, dim3(1), dim3(NT), lds2, s, masks, mask_stride, mb, g, seed_rec, seed, pix, rec, nq, n, cancel_epoch, my_epoch);
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_search() {} }
void preload_k_search(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_search, dim3(1), dim3(1), 0, s); }

}  // namespace lt

