// Thresholds, merge and the 5x5 open of filter_lane_points(), on bit-packed planes, gfx950.
//
//   k_bilateral_tile  bilateral_adaptive_threshold x2 (+ greenery mask) + OR-merge
//                     lane_tracker.py:14-83, 214-215, 221-235   -> 1 bit per pixel
//   k_pack_merge      same merge for already-thresholded u8 planes ('neighborhood' branch, :217-218)
//   k_erode5_bits / k_dilate5_mask   morphologyEx(MORPH_OPEN, 5x5 ellipse)   lane_tracker.py:205, 238
//
// Bit plane layout: row-major, `wpr` = ceil(w/64) 64-bit words per row, bit i of word j = pixel
// x = 64*j + i.  A wave produces one word per row with a single ballot.
//
// Bilateral threshold: for a pixel p with k neighbours per side,
//     pass  <=>  (k*p > S_left + C*k  and  k*p > S_right + C*k)  or  (same for up / down)
// S_* are sums of the k pixels on that side, zero outside the image.
//
// One 256-thread workgroup owns a 64x64 tile.  Its cross-shaped neighbourhood is staged in LDS per
// plane as a row band (64 rows x (64+2k) columns) and a column band ((64+2k) rows x 64 columns).
// Each of the four waves then runs one sliding-window phase:
//   H phase: lane <-> row.  The lane walks along x keeping S_left / S_right as running sums in
//            registers (3 LDS byte reads per pixel) and shifts its 64 verdicts into one u64 -- which
//            is exactly the bit-plane word of that row.  Row pitch is 4 x odd, so the 32 lanes of a
//            DS lane group hit 32 different banks.
//   V phase: lane <-> column, walking down y with S_up / S_down; one ballot per row gives the word.
// R-top-hat H/V and b-top-hat H/V run concurrently on the 4 waves; the words are OR-ed at the end
// (plus the greenery term, lane_tracker.py:223-231, in a second round when mask_noise is on).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <type_traits>
#include <utility>

#include "lt_internal.h"

namespace lt {
namespace {

constexpr int TW = 64, TH = 64;

struct BilateralPlane {
    const uint8_t* src;  // nullptr: plane unused
    int k, C;
};

struct BilateralArgs {
    BilateralPlane pl[3];   // [0] R top-hat, [1] Lab-b top-hat, [2] raw Lab-b for the greenery mask (optional)
    int noise_thresh;
    int h, w, wpr;
    int tiles_x, tiles_y, ntiles;      // per-frame tile grid and the total over all frames
    size_t plane_stride, bits_stride;  // bytes per frame of the u8 planes / u64 words per frame of the bit plane
};

// Workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).  Remap
// the linear id so that each XCD gets one contiguous range of tiles: neighbouring tiles share
// their halo through that XCD's L2 instead of fetching it once per XCD.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__host__ __device__ __forceinline__ int hband_pitch(int k) {   // >= 64 + 2k, multiple of 4, (pitch/4) odd
    int p = (TW + 2 * k + 3 + 3) & ~3;   // up to 3 extra columns on the left from aligning the band start
    if (((p >> 2) & 1) == 0) p += 4;
    return p;
}
__host__ __device__ __forceinline__ int plane_lds_bytes(int k) { return TH * hband_pitch(k) + (TH + 2 * k + 1) * TW; }

__global__ __launch_bounds__(256) void k_bilateral_tile(BilateralArgs a, unsigned long long* __restrict__ bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_words[7][TH];   // [2q] H words, [2q+1] V words of plane q; [6] inRange words
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const int tile = xcd_contiguous(blockIdx.x, a.ntiles);
    const int frame = tile / (a.tiles_x * a.tiles_y), tin = tile - frame * (a.tiles_x * a.tiles_y);
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const size_t fo = (size_t)frame * a.plane_stride;
    const int x0 = txi * TW, y0 = tyi * TH;

    // ---- stage the bands of every active plane (zero outside the image = BORDER_CONSTANT 0) ----
    // The row band starts at the 4-aligned column xa <= x0-k, so that (for w % 4 == 0) every band
    // dword is one aligned global dword, entirely inside or entirely outside the image.
    uint8_t* hband[3];
    uint8_t* vband[3];
    int koff[3];   // index of tile column 0 inside the row band (k .. k+3)
    int off = 0;
    const bool aligned = (a.w & 3) == 0 && (a.plane_stride & 3) == 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        hband[q] = vband[q] = nullptr;
        koff[q] = 0;
        if (!a.pl[q].src) continue;
        const int k = a.pl[q].k, pitch = hband_pitch(k);
        const int xa = (x0 - k) & ~3;
        koff[q] = x0 - xa;
        hband[q] = smem + off;
        vband[q] = hband[q] + TH * pitch;
        off += plane_lds_bytes(k);
        const uint8_t* s = a.pl[q].src + fo;
        if (aligned) {
            // Loads are unconditional on clamped addresses and masked afterwards (a guarded load
            // compiles to branch + load + s_waitcnt vmcnt(0) and serialises the staging loop), and
            // the thread -> (row, dword) mapping needs no division: lane <-> dword column, wave <-> row.
            const int dpr = pitch >> 2;                                   // dwords per band row (<= 81)
            uint32_t* hb = reinterpret_cast<uint32_t*>(hband[q]);
            for (int cd0 = 0; cd0 < dpr; cd0 += 64) {
                const int cd = min(cd0 + lane, dpr - 1), gx = xa + cd * 4;
                const bool xin = gx >= 0 && gx < a.w && cd0 + lane < dpr;
                const uint8_t* colp = s + min(max(gx, 0), a.w - 4);
                for (int r0 = wv; r0 < TH; r0 += 32) {                    // 8 rows of this wave per batch
                    uint32_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int gy = y0 + r0 + 4 * u;
                        v[u] = *reinterpret_cast<const uint32_t*>(colp + (uint32_t)__mul24(min(gy, a.h - 1), a.w));   // < 2^24: a full-rate multiply
                        v[u] = (xin && gy < a.h) ? v[u] : 0u;
                    }
                    if (cd0 + lane < dpr) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) hb[(r0 + 4 * u) * dpr + cd] = v[u];
                    }
                }
            }
            uint32_t* vb = reinterpret_cast<uint32_t*>(vband[q]);
            const int nrows = TH + 2 * k, cdv = lane & 15, rsub = wv * 4 + (lane >> 4);   // 16 rows per block pass
            const int gxv = x0 + cdv * 4;
            const bool xinv = gxv < a.w;
            const uint8_t* colv = s + min(gxv, a.w - 4);
            for (int r0 = rsub; r0 < nrows; r0 += 16 * 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int gy = y0 - k + r0 + 16 * u;
                    v[u] = *reinterpret_cast<const uint32_t*>(colv + (uint32_t)__mul24(min(max(gy, 0), a.h - 1), a.w));
                    v[u] = (xinv && gy >= 0 && gy < a.h) ? v[u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + 16 * u < nrows) vb[(r0 + 16 * u) * 16 + cdv] = v[u];
            }
        } else {
            for (int r = wv; r < TH; r += 4) {                            // row band: rows y0+r, columns xa+c
                const int gy = y0 + r;
                for (int c = lane; c < pitch; c += 64) {
                    const int gx = xa + c;
                    hband[q][r * pitch + c] = (gy < a.h && gx >= 0 && gx < a.w) ? s[(size_t)gy * a.w + gx] : 0;
                }
            }
            const int gx = x0 + lane;
            for (int r = wv; r < TH + 2 * k; r += 4) {                    // column band: rows y0-k+r, columns x0+lane
                const int gy = y0 - k + r;
                vband[q][r * TW + lane] = (gx < a.w && gy >= 0 && gy < a.h) ? s[(size_t)gy * a.w + gx] : 0;
            }
        }
    }
    __syncthreads();

    // ---- sliding-window phases: phase 2q = H of plane q, 2q+1 = V of plane q ----
    for (int ph = wv; ph < 6; ph += 4) {
        const int q = ph >> 1;
        if (!a.pl[q].src) continue;
        const int k = a.pl[q].k, C = a.pl[q].C, Ck = C * k;
        if ((ph & 1) == 0) {
            // lane <-> row; b[c] <-> column x0-k+c (the band itself starts up to 3 columns earlier)
            const uint8_t* b = hband[q] + lane * hband_pitch(k) + (koff[q] - k);
            int sl = 0, sr = 0;
#pragma unroll 8
            for (int c = 0; c < k; ++c) { sl += b[c]; sr += b[k + 1 + c]; }
            unsigned long long word = 0;
#pragma unroll 8
            for (int t = 0; t < TW - 1; ++t) {
                const int p = b[k + t];
                const int thr = k * p - Ck;
                word |= (unsigned long long)(sl < thr && sr < thr) << t;
                sl += p - b[t];                      // slide both windows one column to the right
                sr += b[2 * k + 1 + t] - b[k + 1 + t];
            }
            {
                const int p = b[k + TW - 1];
                const int thr = k * p - Ck;
                word |= (unsigned long long)(sl < thr && sr < thr) << (TW - 1);
            }
            s_words[ph][lane] = word;
        } else {
            const uint8_t* b = vband[q] + lane;                    // lane <-> column; b[r*64] <-> row y0-k+r
            int su = 0, sd = 0;
#pragma unroll 8
            for (int r = 0; r < k; ++r) { su += b[r * TW]; sd += b[(k + 1 + r) * TW]; }
            unsigned long long mine = 0, mine_range = 0;
#pragma unroll 8
            for (int t = 0; t < TH; ++t) {
                const int p = b[(k + t) * TW];
                const int thr = k * p - Ck;
                const unsigned long long bal = __ballot(su < thr && sd < thr);
                if (lane == t) mine = bal;
                if (q == 2) {                                       // inRange(lab_b, noise_thresh, 255)
                    const unsigned long long br = __ballot(p >= a.noise_thresh);
                    if (lane == t) mine_range = br;
                }
                // slide both windows one row down; the band has one spare row so that the last,
                // unused update stays inside it
                su += p - b[t * TW];
                sd += b[(2 * k + 1 + t) * TW] - b[(k + 1 + t) * TW];
            }
            s_words[ph][lane] = mine;
            if (q == 2) s_words[6][lane] = mine_range;
        }
    }
    __syncthreads();
    if (threadIdx.x < TH) {
        const int r = threadIdx.x, gy = y0 + r;
        if (gy < a.h) {
            unsigned long long m = 0;
            if (a.pl[0].src) m |= s_words[0][r] | s_words[1][r];
            if (a.pl[1].src) m |= s_words[2][r] | s_words[3][r];
            if (a.pl[2].src) m &= ~s_words[6][r] | s_words[4][r] | s_words[5][r];   // (r|b) & (!part1 | part2)
            const int nvalid = a.w - x0;
            if (nvalid < 64) m &= (1ull << nvalid) - 1ull;
            bits[(size_t)frame * a.bits_stride + (size_t)gy * a.wpr + txi] = m;
        }
    }
}

// ================================================================================================
// Packed variant (default): 128x128 tile, two rows (H) / two columns (V) per lane in one VGPR.
//
// All window sums are < 2^15 for k <= 128 (k*255 <= 32640), so they live in 16-bit halves and the
// whole test is packed 16-bit arithmetic:   thr = k*p - C*k  (v_pk_mad_i16),
//   d = (S_a - thr) & (S_b - thr)  -> the sign bit of each half says "both sides below thr".
// H phase (lane <-> rows l and l+64, walking 64 columns): the two sign bits are shifted into a
// packed pair of 16-bit shift registers (2 ops per step for both rows); every 16 steps the
// registers are flushed into the lane's two 64-bit row words.  V phase (lane <-> columns l and
// l+64, walking 64 rows): one ballot per half gives the two row words of that row.
// The four waves run H(cols 0-63), H(cols 64-127), V(rows 0-63), V(rows 64-127) of one plane
// concurrently; planes are processed one after the other and OR-ed into LDS accumulators.
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2t __attribute__((ext_vector_type(2)));
constexpr int T2 = 128;

__host__ __device__ __forceinline__ int hband2_pitch(int k) {   // >= 128 + 2k + 3, multiple of 4, (pitch/4) odd
    int p = (T2 + 2 * k + 3 + 3) & ~3;
    if (((p >> 2) & 1) == 0) p += 4;
    return p;
}
// the row band and the column band are staged one after the other into the same buffer
__host__ __device__ __forceinline__ int plane2_lds_bytes(int k) {
    const int hb = T2 * hband2_pitch(k), vb = (T2 + 2 * k + 1) * T2;
    return hb > vb ? hb : vb;
}

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2t)(__builtin_bit_cast(u16x2t, a) + __builtin_bit_cast(u16x2t, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2t)(__builtin_bit_cast(u16x2t, a) - __builtin_bit_cast(u16x2t, b)));
}
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2t)(__builtin_bit_cast(u16x2t, a) * __builtin_bit_cast(u16x2t, b)));
}
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_bit_cast(uint32_t, (u16x2t)(__builtin_bit_cast(u16x2t, a) * __builtin_bit_cast(u16x2t, b) + __builtin_bit_cast(u16x2t, c)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2t, a), __builtin_bit_cast(u16x2t, b)));
}
// four consecutive band bytes from any byte address (LDS serves unaligned dwords)
__device__ __forceinline__ uint32_t lds_dword(const uint8_t* p) {
    struct __attribute__((packed, aligned(1))) U { uint32_t v; };
    return reinterpret_cast<const U*>(p)->v;
}
__device__ __forceinline__ uint32_t pk_shr1(uint32_t a) {
    return __builtin_bit_cast(uint32_t, (u16x2t)(__builtin_bit_cast(u16x2t, a) >> (u16x2t){1, 1}));
}

// Two ballots (four wave-uniform dwords) into lane LANE of four VGPRs with v_writelane_b32.  The operands
// come straight from v_cmp; on gfx950 a VALU write of an SGPR needs wait states before another VALU
// instruction may read it, and the compiler's hazard recogniser does not look inside inline assembly (without
// the s_nop the parity test fails on a handful of pixels), so the wait states are part of the statement.
template <int LANE>
__device__ __forceinline__ void write_lane_words(uint32_t& al, uint32_t& ah, uint32_t& bl, uint32_t& bh,
                                                 unsigned long long a, unsigned long long b) {
    asm("s_nop 3\n\t"
        "v_writelane_b32 %0, %4, %8\n\t"
        "v_writelane_b32 %1, %5, %8\n\t"
        "v_writelane_b32 %2, %6, %8\n\t"
        "v_writelane_b32 %3, %7, %8"
        : "+v"(al), "+v"(ah), "+v"(bl), "+v"(bh)
        : "s"((uint32_t)a), "s"((uint32_t)(a >> 32)), "s"((uint32_t)b), "s"((uint32_t)(b >> 32)), "n"(LANE));
}
template <class F, int... I>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

constexpr int SB = 16;   // staging loads in flight per lane: the bands are latency-bound, not bandwidth-bound

struct Bilateral2Args {
    BilateralPlane pl[3];
    int noise_thresh;
    int h, w, wpr;
    int tiles_x, tiles_y, ntiles;
    size_t plane_stride, bits_stride;
    // split: two workgroups per tile -- one runs the H phases into `bits`, the other the V phases into `bits_v` (two partial planes
    // for the merge to OR).  One frame is 81 tiles on 256 CUs; the halves of a tile's work do not depend on each other.
    int split;
    unsigned long long* bits_v;
};

// Per plane: stage the row band -> H phase on all four waves (32 columns each) -> stage the column
// band into the same LDS -> V phase on all four waves (32 rows each).  One band in LDS at a time
// keeps the workgroup at ~32 KB, i.e. four workgroups per CU instead of two.
__global__ __launch_bounds__(256) void k_bilateral_tile2(Bilateral2Args a, unsigned long long* __restrict__ bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_or[T2][2];      // (R | b) verdict words: [row][column half]
    __shared__ unsigned long long s_noise[T2][2];   // bilateral verdict of the raw Lab-b plane (greenery part 2)
    __shared__ unsigned long long s_range[T2][2];   // inRange(lab_b, noise_thresh, 255)             (part 1)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = xcd_contiguous(blockIdx.x, a.split ? 2 * a.ntiles : a.ntiles);
    const int tile = a.split ? unit >> 1 : unit;
    const int only = a.split ? 1 + (unit & 1) : 3;           // bit 0: the H phases, bit 1: the V phases (workgroup-uniform)
    const int frame = tile / (a.tiles_x * a.tiles_y), tin = tile - frame * (a.tiles_x * a.tiles_y);
    const int tyi = tin / a.tiles_x, txi = tin - tyi * a.tiles_x;
    const size_t fo = (size_t)frame * a.plane_stride;
    const int x0 = txi * T2, y0 = tyi * T2;
    (&s_or[0][0])[threadIdx.x] = 0;
    (&s_noise[0][0])[threadIdx.x] = 0;
    (&s_range[0][0])[threadIdx.x] = 0;
    const bool aligned = (a.w & 3) == 0 && (a.plane_stride & 3) == 0;

    for (int q = 0; q < 3; ++q) {
        if (!a.pl[q].src) continue;
        const int k = a.pl[q].k, C = a.pl[q].C, pitch = hband2_pitch(k);
        const int xa = (x0 - k) & ~3, koff = x0 - xa;   // band column of tile column 0
        uint8_t* band = smem;
        const uint8_t* s = a.pl[q].src + fo;
        const uint32_t kk = (uint32_t)k | ((uint32_t)k << 16);
        const uint32_t nck = (uint32_t)((-C * k) & 0xffff) * 0x10001u;
        unsigned long long (*dst)[2] = q == 2 ? s_noise : s_or;

        // ---------------- row band: rows y0..y0+127, columns xa..xa+pitch-1 ----------------
        __syncthreads();   // the previous phase is done with the buffer
        if (only & 1) {
        if (aligned) {
            const int dpr = pitch >> 2;
            uint32_t* hb = reinterpret_cast<uint32_t*>(band);
            for (int cd0 = 0; cd0 < dpr; cd0 += 64) {
                const int cd = min(cd0 + lane, dpr - 1), gx = xa + cd * 4;
                const bool xin = gx >= 0 && gx < a.w && cd0 + lane < dpr;
                const uint8_t* colp = s + min(max(gx, 0), a.w - 4);
                for (int r0 = wv; r0 < T2; r0 += 4 * SB) {
                    uint32_t v[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int gy = y0 + r0 + 4 * u;
                        v[u] = *reinterpret_cast<const uint32_t*>(colp + (uint32_t)__mul24(min(gy, a.h - 1), a.w));   // < 2^24: a full-rate multiply
                        v[u] = (xin && gy < a.h) ? v[u] : 0u;
                    }
                    if (cd0 + lane < dpr) {
#pragma unroll
                        for (int u = 0; u < SB; ++u) hb[(r0 + 4 * u) * dpr + cd] = v[u];
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < T2 * pitch; i += 256) {
                const int r = i / pitch, c = i - r * pitch, gy = y0 + r, gx = xa + c;
                band[i] = (gy < a.h && gx >= 0 && gx < a.w) ? s[(size_t)gy * a.w + gx] : 0;
            }
        }
        __syncthreads();
        if (x0 + 32 * wv < a.w) {   // a wave whose 32 columns lie beyond the image has no verdicts to add (wave-uniform)
            // ---- H phase: rows (lane, lane+64), tile columns 32*wv .. 32*wv+31
            // The band is read a dword (four columns of one row) at a time -- LDS takes unaligned dwords -- and
            // v_perm_b32 builds the packed pair (row l | row l+64 << 16) of a column from the two rows' dwords in
            // ONE instruction; byte reads cost an extract and an insert per value.  The test itself is
            //     thr = k p - C k            (v_pk_mad_u16)
            //     pass <=> max(S_l, S_r) < thr   (v_pk_max_u16, v_pk_sub_i16: the sign bit is the verdict)
            // All sums are < 2^15, thr > -2^15: signed 16-bit differences cannot wrap.
            const uint8_t* b0 = band + lane * pitch + koff - k + 32 * wv;   // b[c] <-> tile column 32*wv - k + c
            const uint8_t* b1 = b0 + 64 * pitch;
            uint32_t sla = 0, slb = 0, sra = 0, srb = 0;   // prologue: v_sad_u8 against 0 adds the four bytes of a dword
            int c = 0;
            for (; c + 4 <= k; c += 4) {
                sla = __builtin_amdgcn_sad_u8(lds_dword(b0 + c), 0u, sla);
                slb = __builtin_amdgcn_sad_u8(lds_dword(b1 + c), 0u, slb);
                sra = __builtin_amdgcn_sad_u8(lds_dword(b0 + k + 1 + c), 0u, sra);
                srb = __builtin_amdgcn_sad_u8(lds_dword(b1 + k + 1 + c), 0u, srb);
            }
            if (k & 3) {
                const uint32_t keep = (1u << (8 * (k & 3))) - 1u;   // the band row extends past 2k+32 columns: the read stays inside it
                sla = __builtin_amdgcn_sad_u8(lds_dword(b0 + c) & keep, 0u, sla);
                slb = __builtin_amdgcn_sad_u8(lds_dword(b1 + c) & keep, 0u, slb);
                sra = __builtin_amdgcn_sad_u8(lds_dword(b0 + k + 1 + c) & keep, 0u, sra);
                srb = __builtin_amdgcn_sad_u8(lds_dword(b1 + k + 1 + c) & keep, 0u, srb);
            }
            uint32_t sl = sla | (slb << 16), sr = sra | (srb << 16);
            uint32_t w0 = 0, w1 = 0;
            uint32_t P = (uint32_t)b0[k] | ((uint32_t)b1[k] << 16);
            const uint8_t *o0 = b0, *o1 = b1, *p0 = b0 + k + 1, *p1 = b1 + k + 1, *i0 = b0 + 2 * k + 1, *i1 = b1 + 2 * k + 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint32_t acc = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t4 = g * 16 + j * 4;
                    const uint32_t ao = lds_dword(o0 + t4), bo = lds_dword(o1 + t4);
                    const uint32_t ap = lds_dword(p0 + t4), bp = lds_dword(p1 + t4);
                    const uint32_t ai = lds_dword(i0 + t4), bi = lds_dword(i1 + t4);
                    for_each_const([&](auto uc) {
                        constexpr uint32_t u = decltype(uc)::value, SEL = 0x0c000c00u | u | ((4u + u) << 16);
                        const uint32_t thr = pk_mad(P, kk, nck);
                        const uint32_t m = pk_sub(pk_max(sl, sr), thr);               // sign bits: both sides < thr
                        acc = pk_shr1(acc) | (m & 0x80008000u);
                        const uint32_t outl = __builtin_amdgcn_perm(bo, ao, SEL);
                        const uint32_t in = __builtin_amdgcn_perm(bi, ai, SEL);
                        const uint32_t Pn = __builtin_amdgcn_perm(bp, ap, SEL);
                        sl = pk_sub(pk_add(sl, P), outl);
                        sr = pk_sub(pk_add(sr, in), Pn);
                        P = Pn;
                    }, std::make_integer_sequence<int, 4>{});
                }
                w0 |= (acc & 0xffffu) << (16 * g);
                w1 |= (acc >> 16) << (16 * g);
            }
            const int sh = 32 * (wv & 1), half = wv >> 1;
            atomicOr(&dst[lane][half], (unsigned long long)w0 << sh);
            atomicOr(&dst[lane + 64][half], (unsigned long long)w1 << sh);
        }
        }
        if (!(only & 2)) continue;

        // ---------------- column band: rows y0-k..y0+127+k+1, columns x0..x0+127 ----------------
        __syncthreads();   // every H wave is done with the row band
        // Layout of a band row (128 bytes): byte 2c = tile column c, byte 2c+1 = tile column c + 64, so that the V
        // phase fetches both of a lane's columns with ONE 16-bit read.
        if (aligned) {
            uint2* vb = reinterpret_cast<uint2*>(band);
            const int nrows = T2 + 2 * k + 1, cdv = lane & 15, rsub = wv * 4 + (lane >> 4);   // 16 rows per block pass
            const int gxa = x0 + cdv * 4, gxb = gxa + 64;
            const bool xina = gxa < a.w, xinb = gxb < a.w;
            const uint8_t *cola = s + min(gxa, a.w - 4), *colb = s + min(gxb, a.w - 4);
            constexpr int SV = SB / 2;
            for (int r0 = rsub; r0 < nrows; r0 += 16 * SV) {
                uint32_t va[SV], vc[SV];
#pragma unroll
                for (int u = 0; u < SV; ++u) {
                    const int gy = y0 - k + r0 + 16 * u;
                    const uint32_t ro = (uint32_t)__mul24(min(max(gy, 0), a.h - 1), a.w);
                    va[u] = *reinterpret_cast<const uint32_t*>(cola + ro);
                    vc[u] = *reinterpret_cast<const uint32_t*>(colb + ro);
                    const bool yin = gy >= 0 && gy < a.h;
                    va[u] = (xina && yin) ? va[u] : 0u;
                    vc[u] = (xinb && yin) ? vc[u] : 0u;
                }
#pragma unroll
                for (int u = 0; u < SV; ++u)
                    if (r0 + 16 * u < nrows)
                        vb[(r0 + 16 * u) * 16 + cdv] = make_uint2(__builtin_amdgcn_perm(vc[u], va[u], 0x05010400u),
                                                                  __builtin_amdgcn_perm(vc[u], va[u], 0x07030602u));
            }
        } else {
            for (int i = threadIdx.x; i < (T2 + 2 * k + 1) * T2; i += 256) {
                const int r = i >> 7, c = i & 127, gy = y0 - k + r, gx = x0 + c;
                band[r * T2 + ((c & 63) << 1) + (c >> 6)] = (gx < a.w && gy >= 0 && gy < a.h) ? s[(size_t)gy * a.w + gx] : 0;
            }
        }
        __syncthreads();
        if (y0 + 32 * wv < a.h) {   // likewise for 32 rows below the image
            // ---- V phase: columns (lane, lane+64), tile rows 32*wv .. 32*wv+31
            const int rbase = 32 * wv;
            const uint8_t* b = band + rbase * T2 + 2 * lane;   // b[r*128] <-> tile row rbase - k + r, columns (lane, lane+64)
            auto pair = [](const uint8_t* q) {                  // (col lane | col lane+64 << 16) from the two adjacent bytes
                return __builtin_amdgcn_perm(0u, (uint32_t)*reinterpret_cast<const uint16_t*>(q), 0x0c010c00u);
            };
            uint32_t su = 0, sd = 0;
#pragma unroll 4
            for (int r = 0; r < k; ++r) {
                su += pair(b + r * T2);
                sd += pair(b + (k + 1 + r) * T2);
            }
            uint32_t P = pair(b + k * T2);
            // Row t's verdict word is wave-uniform (a ballot) and belongs in lane t: v_writelane puts an SGPR
            // into one lane of a VGPR in a single instruction.  The greenery range test only exists for the
            // third plane; keeping it out of the other two loops keeps them free of branches, so the 32
            // steps unroll into one block and the LDS reads are issued ahead of their use.
            uint32_t m0l = 0, m0h = 0, m1l = 0, m1h = 0, g0l = 0, g0h = 0, g1l = 0, g1h = 0;
            auto vsteps = [&](auto with_range) {
                for_each_const([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    const uint32_t thr = pk_mad(P, kk, nck);
                    const uint32_t m = pk_sub(pk_max(su, sd), thr);
                    // sign of the low half in ONE compare (v_cmp_gt_i16 looks at bits 15:0); written in C the
                    // compiler extracts bit 15 first (v_bfe_u32 + v_cmp_ne_u32)
                    unsigned long long bal0;
                    asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(bal0) : "v"(m));
                    const unsigned long long bal1 = __ballot((int32_t)m < 0);
                    write_lane_words<t>(m0l, m0h, m1l, m1h, bal0, bal1);
                    if (decltype(with_range)::value) {   // inRange(lab_b, noise_thresh, 255) on the raw plane
                        const unsigned long long r0 = __ballot((int)(P & 0xffffu) >= a.noise_thresh),
                                                 r1 = __ballot((int)(P >> 16) >= a.noise_thresh);
                        write_lane_words<t>(g0l, g0h, g1l, g1h, r0, r1);
                    }
                    const uint32_t outu = pair(b + t * T2), in = pair(b + (2 * k + 1 + t) * T2), Pn = pair(b + (k + 1 + t) * T2);
                    su = pk_sub(pk_add(su, P), outu);
                    sd = pk_sub(pk_add(sd, in), Pn);
                    P = Pn;
                }, std::make_integer_sequence<int, 32>{});
            };
            if (q == 2) vsteps(std::true_type{});
            else vsteps(std::false_type{});
            if (lane < 32) {
                atomicOr(&dst[rbase + lane][0], (unsigned long long)m0l | ((unsigned long long)m0h << 32));
                atomicOr(&dst[rbase + lane][1], (unsigned long long)m1l | ((unsigned long long)m1h << 32));
                if (q == 2) {
                    s_range[rbase + lane][0] = (unsigned long long)g0l | ((unsigned long long)g0h << 32);
                    s_range[rbase + lane][1] = (unsigned long long)g1l | ((unsigned long long)g1h << 32);
                }
            }
        }
    }
    __syncthreads();
    {
        const int r = threadIdx.x >> 1, half = threadIdx.x & 1, gy = y0 + r, xw = x0 + 64 * half;
        if (gy < a.h && xw < a.w) {
            unsigned long long m = s_or[r][half];
            if (a.pl[2].src) m &= ~s_range[r][half] | s_noise[r][half];   // (r|b) & (!part1 | part2)
            const int nvalid = a.w - xw;
            if (nvalid < 64) m &= (1ull << nvalid) - 1ull;
            (only == 2 ? a.bits_v : bits)[(size_t)frame * a.bits_stride + (size_t)gy * a.wpr + (xw >> 6)] = m;
        }
    }
}

// merge of already-thresholded u8 planes into the bit plane
__global__ __launch_bounds__(256) void k_pack_merge(const uint8_t* __restrict__ tr, const uint8_t* __restrict__ tb,
                                                   const uint8_t* __restrict__ labb, const uint8_t* __restrict__ nb,
                                                   int noise_thresh, int use_noise, int h, int w, int wpr,
                                                   size_t plane_stride, size_t bits_stride,
                                                   unsigned long long* __restrict__ bits) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int word = blockIdx.x * 4 + wv, y = blockIdx.y;
    if (word >= wpr) return;
    const int x = word * 64 + lane;
    const size_t o = (size_t)blockIdx.z * plane_stride + (size_t)y * w + x;
    bool v = false;
    if (x < w) {
        v = tr[o] || tb[o];
        if (use_noise) v = v && (!(labb[o] >= noise_thresh) || nb[o]);
    }
    const unsigned long long b = __ballot(v);
    if (lane == 0) bits[(size_t)blockIdx.z * bits_stride + (size_t)y * wpr + word] = b;
}

// ---- 5x5 ellipse on bit planes: rows -2,+2 contribute the centre column, rows -1,0,+1 five columns ----
__device__ __forceinline__ unsigned long long load_word(const unsigned long long* p, int y, int j, int h, int wpr,
                                                       unsigned long long outside) {
    return (y >= 0 && y < h && j >= 0 && j < wpr) ? p[(size_t)y * wpr + j] : outside;
}

template <bool DIL>
__device__ __forceinline__ unsigned long long hspan5(unsigned long long l, unsigned long long m,
                                                    unsigned long long r) {
    const unsigned long long a1 = (m << 1) | (l >> 63), a2 = (m << 2) | (l >> 62);
    const unsigned long long b1 = (m >> 1) | (r << 63), b2 = (m >> 2) | (r << 62);
    return DIL ? (m | a1 | a2 | b1 | b2) : (m & a1 & a2 & b1 & b2);
}

// valid-pixel mask of word j (pixels >= w do not exist)
__device__ __forceinline__ unsigned long long valid_bits(int j, int w) {
    const int n = w - j * 64;
    return n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
}

// erode: out-of-image taps are ignored = treated as set.  Output bits beyond w are cleared.
__global__ __launch_bounds__(256) void k_erode5_bits(const unsigned long long* __restrict__ in,
                                                    unsigned long long* __restrict__ out, int h, int w, int wpr,
                                                    size_t bits_stride) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= h * wpr) return;
    const int y = idx / wpr, j = idx - y * wpr;
    const unsigned long long* p = in + (size_t)blockIdx.z * bits_stride;
    auto ld = [&](int yy, int jj) {
        unsigned long long v = load_word(p, yy, jj, h, wpr, ~0ull);
        if (yy >= 0 && yy < h && jj >= 0 && jj < wpr) v |= ~valid_bits(jj, w);   // pixels past the right edge count as set
        return v;
    };
    unsigned long long e = ld(y - 2, j) & ld(y + 2, j);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) e &= hspan5<false>(ld(y + dy, j - 1), ld(y + dy, j), ld(y + dy, j + 1));
    out[(size_t)blockIdx.z * bits_stride + idx] = e & valid_bits(j, w);
}

// OR-merge of the four partial planes of the walking thresholds + the 5x5 open, one pass (NP = 4), or the open alone of
// a merged plane (NP = 1).  k_or4_bits + k_erode5_bits + k_dilate5_bits issue 26 loads and 3 stores per 64-pixel word,
// and all three are bound by exactly that count (profiles/r02_vmem_issue.json: a wave's load costs the CU ~20 cycles
// whatever its width); here a lane owns one word column of one frame and walks down a band of rows, so every word is
// loaded once (+ 8 halo rows per band) and stored once.  A wave holds G = 64 / wpr frames side by side (lane = g * wpr
// + j): the left / right neighbour words of a row are the adjacent lanes (whole-wave DPP shifts; the first and last word
// of a row take the border value instead).  Rows in flight: m (merged), A = 5-wide horizontal AND of m, e (eroded), O =
// 5-wide horizontal OR of e;  e[y] = m[y-2] & A[y-1] & A[y] & A[y+1] & m[y+2],  d[y] = e[y-2] | O[y-1] | O[y] | O[y+1] | e[y+2]
// (the 17-tap ellipse: one pixel in the outer rows, five in the three inner ones).  Erode ignores taps outside the image
// (= set), dilate too (= clear).  NP = 4 also writes the merged plane over p0 (lt_download_plane(LT_PLANE_MERGED)); a
// neighbouring band that still reads that row gets the same OR either way.
// NP = 2: the two thresholded planes of the 'neighborhood' filter (k_adaptive_walk.hip).
// NP = 6: the four partial planes AND-ed with the OR of the two greenery-mask planes n0 / n1 (mask_noise, lane_tracker.py:229).
template <int NP>
__global__ __launch_bounds__(64) void k_merge_open5(unsigned long long* p0, const unsigned long long* __restrict__ p1,
                                                   const unsigned long long* __restrict__ p2, const unsigned long long* __restrict__ p3,
                                                   const unsigned long long* __restrict__ n0, const unsigned long long* __restrict__ n1,
                                                   unsigned long long* __restrict__ opened, int h, int w, int wpr, size_t bits_stride,
                                                   int band_rows, int G, int n) {
    typedef unsigned long long u64;
    const int lane = threadIdx.x, g = lane / wpr, j = lane - g * wpr;
    const int frame = blockIdx.x * G + g;
    const bool active = g < G && frame < n;
    const size_t base = (size_t)(active ? frame : 0) * bits_stride + (size_t)j;
    const int yb0 = blockIdx.y * band_rows, yb1 = min(yb0 + band_rows, h);
    const bool has_left = j > 0, has_right = j < wpr - 1;
    const u64 vb = valid_bits(j, w);
    auto shift_in = [&](u64 v, u64 border, u64& l, u64& r) {   // words of lanes -1 / +1, or the border value at the ends of a row
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        const uint32_t llo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x138, 0xf, 0xf, true);   // wave_shr:1: lane i <- lane i-1
        const uint32_t lhi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xf, 0xf, true);
        const uint32_t rlo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xf, 0xf, true);   // wave_shl:1: lane i <- lane i+1
        const uint32_t rhi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x130, 0xf, 0xf, true);
        l = has_left ? ((u64)lhi << 32 | llo) : border;
        r = has_right ? ((u64)rhi << 32 | rlo) : border;
    };
    auto load_row = [&](int y, u64 (&q)[6]) {
        const size_t o = base + (size_t)min(max(y, 0), h - 1) * wpr;
        q[0] = p0[o];
        if (NP >= 2) q[1] = p1[o];
        if (NP >= 4) { q[2] = p2[o]; q[3] = p3[o]; }
        if (NP == 6) { q[4] = n0[o]; q[5] = n1[o]; }
    };
    u64 m0 = ~0ull, m1 = ~0ull, m2 = ~0ull, m3 = ~0ull;     // m[t-4 .. t-1]
    u64 a1 = ~0ull, a2 = ~0ull, a3 = ~0ull;                  // A[t-3 .. t-1]
    u64 e0 = 0, e1 = 0, e2 = 0, e3 = 0;                      // e[t-6 .. t-3]
    u64 o1 = 0, o2 = 0, o3 = 0;                              // O[t-5 .. t-3]
    u64 q[6], qn[6];
    load_row(yb0 - 4, q);
    for (int t = yb0 - 4; t <= yb1 + 3; ++t) {
        load_row(t + 1, qn);                                  // next row in flight while this one goes through the pipeline
        const u64 raw = NP == 6 ? ((q[0] | q[1] | q[2] | q[3]) & (q[4] | q[5])) : NP == 4 ? (q[0] | q[1] | q[2] | q[3]) : NP == 2 ? (q[0] | q[1]) : q[0];
        const bool in_img = t >= 0 && t < h;
        if (NP >= 2 && active && t >= yb0 && t < yb1) p0[base + (size_t)t * wpr] = raw;
        const u64 m = in_img ? (raw | ~vb) : ~0ull;           // pixels right of / rows outside the image count as set
        u64 l, r;
        shift_in(m, ~0ull, l, r);
        const u64 a = hspan5<false>(l, m, r);
        const int ye = t - 2;
        u64 e = m0 & a1 & a2 & a3 & m;
        e = (ye >= 0 && ye < h) ? (e & vb) : 0ull;
        shift_in(e, 0ull, l, r);
        const u64 o = hspan5<true>(l, e, r);
        const int yd = t - 4;
        const u64 d = (e0 | o1 | o2 | o3 | e) & vb;
        if (active && yd >= yb0 && yd < yb1) opened[base + (size_t)yd * wpr] = d;
        m0 = m1; m1 = m2; m2 = m3; m3 = m;
        a1 = a2; a2 = a3; a3 = a;
        e0 = e1; e1 = e2; e2 = e3; e3 = e;
        o1 = o2; o2 = o3; o3 = o;
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = qn[k];
    }
}

// A few frames (process(): one): OR of the partial planes + the 5x5 open in ONE launch of many small workgroups.  k_merge_open5
// walks bands of rows, one wave per band -- a few dozen waves for one frame, each a chain of dependent rows; the three shallow
// kernels (k_or4_bits, k_erode5_bits, k_dilate5_bits) are three launches of 5 us each on a one-frame chain that is made of
// launch gaps.  Here a workgroup owns RB output rows: the merged words of RB + 8 rows go to LDS, the eroded words of RB + 4 rows
// are formed from them into LDS, the dilated words from those.  Same arithmetic as k_merge_open5 (borders: erode ignores taps
// outside the image, dilate too); NP as there; NP >= 2 also writes the merged plane over p0.
constexpr int OS_RB = 8;
template <int NP>
__global__ __launch_bounds__(256) void k_or_open5_small(unsigned long long* p0, const unsigned long long* __restrict__ p1,
                                                       const unsigned long long* __restrict__ p2, const unsigned long long* __restrict__ p3,
                                                       const unsigned long long* __restrict__ n0, const unsigned long long* __restrict__ n1,
                                                       unsigned long long* __restrict__ opened, int h, int w, int wpr, size_t bits_stride) {
    typedef unsigned long long u64;
    __shared__ u64 s_m[(OS_RB + 8) * 64], s_e[(OS_RB + 4) * 64];
    const int yb = blockIdx.x * OS_RB, rows = min(OS_RB, h - yb);
    const size_t fo = (size_t)blockIdx.y * bits_stride;
    for (int i = threadIdx.x; i < (OS_RB + 8) * wpr; i += 256) {
        const int r = i / wpr, j = i - r * wpr, y = yb - 4 + r;
        u64 m = ~0ull;                                        // rows outside the image count as set (erode ignores them)
        if (y >= 0 && y < h) {
            const size_t o = fo + (size_t)y * wpr + j;
            u64 raw = p0[o];
            if (NP >= 2) raw |= p1[o];
            if (NP >= 4) raw |= p2[o] | p3[o];
            if (NP == 6) raw &= n0[o] | n1[o];
            if (NP >= 2 && r >= 4 && r < 4 + rows) p0[o] = raw;
            m = raw | ~valid_bits(j, w);                      // ... and so do pixels right of it
        }
        s_m[r * 64 + j] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (OS_RB + 4) * wpr; i += 256) {
        const int r = i / wpr, j = i - r * wpr, y = yb - 2 + r;   // eroded row y <-> merged rows r .. r + 4 of s_m
        u64 e = 0ull;
        if (y >= 0 && y < h) {
            e = s_m[r * 64 + j] & s_m[(r + 4) * 64 + j];
#pragma unroll
            for (int d = 1; d <= 3; ++d) {
                const u64* q = s_m + (r + d) * 64;
                e &= hspan5<false>(j > 0 ? q[j - 1] : ~0ull, q[j], j < wpr - 1 ? q[j + 1] : ~0ull);
            }
            e &= valid_bits(j, w);
        }
        s_e[r * 64 + j] = e;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * wpr; i += 256) {
        const int r = i / wpr, j = i - r * wpr;                   // output row yb + r <-> eroded rows r .. r + 4 of s_e
        u64 d = s_e[r * 64 + j] | s_e[(r + 4) * 64 + j];
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const u64* q = s_e + (r + k) * 64;
            d |= hspan5<true>(j > 0 ? q[j - 1] : 0ull, q[j], j < wpr - 1 ? q[j + 1] : 0ull);
        }
        opened[fo + (size_t)(yb + r) * wpr + j] = d & valid_bits(j, w);
    }
}

// dilate of the eroded bits (out-of-image = clear) and expansion to the u8 {0,255} mask.
// A block owns DR rows: its threads first build the dilated words of those rows in LDS, then
// every thread expands 4 pixels at a time into one coalesced dword store.
constexpr int DR = 16, DW_MAX = 64;   // rows per block, max words per row handled by the LDS buffer (w <= 4096)
__global__ __launch_bounds__(256) void k_dilate5_mask(const unsigned long long* __restrict__ in,
                                                     uint8_t* __restrict__ mask, int h, int w, int wpr,
                                                     size_t bits_stride, size_t plane_stride) {
    __shared__ unsigned long long s_d[DR * DW_MAX];
    const int yb = blockIdx.x * DR, rows = min(DR, h - yb);
    const unsigned long long* p = in + (size_t)blockIdx.z * bits_stride;
    auto ld = [&](int yy, int jj) { return load_word(p, yy, jj, h, wpr, 0ull); };
    for (int i = threadIdx.x; i < rows * wpr; i += 256) {
        const int r = i / wpr, j = i - r * wpr, y = yb + r;
        unsigned long long d = ld(y - 2, j) | ld(y + 2, j);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) d |= hspan5<true>(ld(y + dy, j - 1), ld(y + dy, j), ld(y + dy, j + 1));
        s_d[r * wpr + j] = d;
    }
    __syncthreads();
    uint8_t* out = mask + (size_t)blockIdx.z * plane_stride;
    if ((w & 3) == 0 && (plane_stride & 3) == 0) {
        const int qpr = w >> 2;   // dwords per row
        for (int i = threadIdx.x; i < rows * qpr; i += 256) {
            const int r = i / qpr, qd = i - r * qpr, x = qd * 4;
            const unsigned nib = (unsigned)(s_d[r * wpr + (x >> 6)] >> (x & 63)) & 0xfu;
            // 4 bits -> 4 bytes of 0x00 / 0xff
            const unsigned spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
            reinterpret_cast<uint32_t*>(out + (size_t)(yb + r) * w)[qd] = spread * 0xffu;
        }
    } else {
        for (int i = threadIdx.x; i < rows * w; i += 256) {
            const int r = i / w, x = i - r * w;
            out[(size_t)(yb + r) * w + x] = ((s_d[r * wpr + (x >> 6)] >> (x & 63)) & 1ull) ? 255 : 0;
        }
    }
}

// The same dilation, bit plane out: what the searches read (one word per thread, like k_erode5_bits)
__global__ __launch_bounds__(256) void k_dilate5_bits(const unsigned long long* __restrict__ in,
                                                     unsigned long long* __restrict__ out, int h, int w, int wpr,
                                                     size_t bits_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * wpr) return;
    const int y = i / wpr, j = i - y * wpr;
    const unsigned long long* p = in + (size_t)blockIdx.z * bits_stride;
    auto ld = [&](int yy, int jj) { return load_word(p, yy, jj, h, wpr, 0ull); };
    unsigned long long d = ld(y - 2, j) | ld(y + 2, j);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) d |= hspan5<true>(ld(y + dy, j - 1), ld(y + dy, j), ld(y + dy, j + 1));
    const int nvalid = w - j * 64;
    if (nvalid < 64) d &= (1ull << nvalid) - 1ull;           // no pixels beyond the image width
    out[(size_t)blockIdx.z * bits_stride + i] = d;
}

// bit plane -> u8 {0, 255}: four pixels per thread, one dword store (w % 4 == 0), else one pixel per thread
__global__ __launch_bounds__(256) void k_bits_to_u8x4(const unsigned long long* __restrict__ in, uint8_t* __restrict__ out,
                                                     int h, int w, int wpr, size_t bits_stride, size_t plane_stride) {
    const int qpr = w >> 2, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * qpr) return;
    const int y = i / qpr, qd = i - y * qpr, x = qd * 4;
    const unsigned nib = (unsigned)(in[(size_t)blockIdx.z * bits_stride + (size_t)y * wpr + (x >> 6)] >> (x & 63)) & 0xfu;
    const unsigned spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    reinterpret_cast<uint32_t*>(out + (size_t)blockIdx.z * plane_stride + (size_t)y * w)[qd] = spread * 0xffu;
}

__global__ __launch_bounds__(256) void k_bits_to_u8(const unsigned long long* __restrict__ in, uint8_t* __restrict__ out,
                                                   int h, int w, int wpr, size_t bits_stride, size_t plane_stride) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wv, y = blockIdx.y;
    if (j >= wpr) return;
    const unsigned long long d = in[(size_t)blockIdx.z * bits_stride + (size_t)y * wpr + j];
    const int x = j * 64 + lane;
    if (x < w) out[(size_t)blockIdx.z * plane_stride + (size_t)y * w + x] = ((d >> lane) & 1ull) ? 255 : 0;
}

}  // namespace

int launch_bilateral_bits(hipStream_t s, const uint8_t* thr, int k_r, int C_r, const uint8_t* thb, int k_b, int C_b,
                          const uint8_t* labb, int k_n, int C_n, int noise_thresh, int use_noise,
                          unsigned long long* bits, int h, int w, size_t plane_stride, size_t bits_stride, int n,
                          unsigned long long* bits_v) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    if (bits_v && use_noise) return -1;      // the split form has no greenery mask
    static const bool unpacked = [] { const char* e = LT_EXP_ENV("LT_BILATERAL_UNPACKED"); return e && e[0] == '1'; }();
    const int kmax = std::max(k_r, std::max(k_b, use_noise ? k_n : 0));
    const int cmax = std::max(std::abs(C_r), std::max(std::abs(C_b), use_noise ? std::abs(C_n) : 0));
    // packed 16-bit arithmetic is exact while every |S - (k*p - C*k)| < 2^15
    const bool fits16 = (long long)kmax * (255 + cmax) < 32768 && C_r >= 0 && C_b >= 0 && (!use_noise || C_n >= 0);
    if (!unpacked && fits16) {
        Bilateral2Args a;
        a.pl[0] = {thr, k_r, C_r};
        a.pl[1] = {thb, k_b, C_b};
        a.pl[2] = {use_noise ? labb : nullptr, k_n, C_n};
        a.noise_thresh = noise_thresh;
        a.h = h; a.w = w; a.wpr = (w + 63) / 64;
        a.plane_stride = plane_stride;
        a.bits_stride = bits_stride;
        a.tiles_x = (w + T2 - 1) / T2;
        a.tiles_y = (h + T2 - 1) / T2;
        a.ntiles = a.tiles_x * a.tiles_y * n;
        a.split = bits_v ? 1 : 0;
        a.bits_v = bits_v;
        const size_t lds = (size_t)plane2_lds_bytes(kmax);
        if (lds + 3 * T2 * 2 * sizeof(unsigned long long) <= 160 * 1024) {
            if (lds > 48 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral_tile2),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return -1;
            static const bool report = [] { const char* e = LT_EXP_ENV("LT_REPORT_OCCUPANCY"); return e && e[0] == '1'; }();
            if (report) {
                int nb = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_bilateral_tile2), 256, lds);
                std::fprintf(stderr, "k_bilateral_tile2: %zu B dynamic LDS, %d workgroups per CU\n", lds, nb);
            }
            hipLaunchKernelGGL(k_bilateral_tile2, dim3(a.split ? 2 * a.ntiles : a.ntiles), dim3(256), lds, s, a, bits);
            return 0;
        }
    }
    if (bits_v) return -1;
    BilateralArgs a;
    a.pl[0] = {thr, k_r, C_r};
    a.pl[1] = {thb, k_b, C_b};
    a.pl[2] = {use_noise ? labb : nullptr, k_n, C_n};
    a.noise_thresh = noise_thresh;
    a.h = h;
    a.w = w;
    a.wpr = (w + 63) / 64;
    a.plane_stride = plane_stride;
    a.bits_stride = bits_stride;
    size_t lds = 0;
    for (int q = 0; q < 3; ++q)
        if (a.pl[q].src) lds += (size_t)plane_lds_bytes(a.pl[q].k);
    if (lds + sizeof(unsigned long long) * 7 * TH > 160 * 1024) return -1;  // caller runs one plane at a time
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_bilateral_tile), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return -1;
    a.tiles_x = (w + TW - 1) / TW;
    a.tiles_y = (h + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * n;
    hipLaunchKernelGGL(k_bilateral_tile, dim3(a.ntiles), dim3(256), lds, s, a, bits);
    return 0;
}

void launch_pack_merge(hipStream_t s, const uint8_t* tr, const uint8_t* tb, const uint8_t* labb, const uint8_t* nb,
                       int noise_thresh, int use_noise, unsigned long long* bits, int h, int w, size_t plane_stride,
                       size_t bits_stride, int n) {
    if (n <= 0 || h <= 0 || w <= 0) return;
    const int wpr = (w + 63) / 64;
    dim3 grid((wpr + 3) / 4, h, n);
    hipLaunchKernelGGL(k_pack_merge, grid, dim3(256), 0, s, tr, tb, labb, nb, noise_thresh, use_noise, h, w, wpr,
                       plane_stride, bits_stride, bits);
}

void launch_open5_bits(hipStream_t s, const unsigned long long* merged, unsigned long long* eroded, uint8_t* mask, int h,
                       int w, size_t plane_stride, size_t bits_stride, int n) {
    if (n <= 0 || h <= 0 || w <= 0) return;
    const int wpr = (w + 63) / 64;
    hipLaunchKernelGGL(k_erode5_bits, dim3((h * wpr + 255) / 256, 1, n), dim3(256), 0, s, merged, eroded, h, w, wpr,
                       bits_stride);
    hipLaunchKernelGGL(k_dilate5_mask, dim3((h + DR - 1) / DR, 1, n), dim3(256), 0, s, eroded, mask, h, w, wpr, bits_stride,
                       plane_stride);
}

void launch_open5_to_bits(hipStream_t s, const unsigned long long* merged, unsigned long long* eroded,
                          unsigned long long* opened, int h, int w, size_t bits_stride, int n) {
    if (n <= 0 || h <= 0 || w <= 0) return;
    const int wpr = (w + 63) / 64;
    hipLaunchKernelGGL(k_erode5_bits, dim3((h * wpr + 255) / 256, 1, n), dim3(256), 0, s, merged, eroded, h, w, wpr,
                       bits_stride);
    hipLaunchKernelGGL(k_dilate5_bits, dim3((h * wpr + 255) / 256, 1, n), dim3(256), 0, s, eroded, opened, h, w, wpr,
                       bits_stride);
}

// merged (or the first of four partial planes, OR-ed in place) -> opened, both bit planes; false when the row does not
// fit a wave (w > 4096): the caller runs the separate kernels
bool launch_merge_open5(hipStream_t s, unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                        const unsigned long long* p3, unsigned long long* opened, int h, int w, size_t bits_stride, int n,
                        const unsigned long long* n0, const unsigned long long* n1) {
    const int wpr = (w + 63) / 64;
    if ((n0 || n1) && !(n0 && n1 && p1 && p2 && p3)) return false;
    if ((p2 != nullptr) != (p3 != nullptr) || (p2 && !p1)) return false;
    static const bool off = [] { const char* e = LT_EXP_ENV("LT_OPEN5_SEPARATE"); return e && e[0] == '1'; }();   // A/B
    if (off || n <= 0 || h <= 0 || wpr > 64 || opened == p0) return false;
    const int G = 64 / wpr;
    static const int rows_env = [] { const char* e = LT_EXP_ENV("LT_OPEN5_BAND_ROWS"); return e ? std::atoi(e) : 0; }();
    // enough bands to give every SIMD a few waves, none shorter than 24 rows (8 halo rows per band are recomputed)
    const int groups = (n + G - 1) / G;
    int nbands = std::max(1, std::min(h / 24, (4096 + groups - 1) / groups));
    int band_rows = rows_env > 0 ? rows_env : (h + nbands - 1) / nbands;
    nbands = (h + band_rows - 1) / band_rows;
    dim3 grid(groups, nbands);
    if (n0) hipLaunchKernelGGL(k_merge_open5<6>, grid, dim3(64), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride, band_rows, G, n);
    else if (p1 && !p2) hipLaunchKernelGGL(k_merge_open5<2>, grid, dim3(64), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride, band_rows, G, n);
    else if (p1) hipLaunchKernelGGL(k_merge_open5<4>, grid, dim3(64), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride, band_rows, G, n);
    else hipLaunchKernelGGL(k_merge_open5<1>, grid, dim3(64), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride, band_rows, G, n);
    return true;
}

// the same for a few frames (k_or_open5_small); false: not launched (a row wider than 64 words, LT_OPEN_SMALL=0)
bool launch_or_open5_small(hipStream_t s, unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                           const unsigned long long* p3, unsigned long long* opened, int h, int w, size_t bits_stride, int n,
                           const unsigned long long* n0, const unsigned long long* n1) {
    const int wpr = (w + 63) / 64;
    if ((n0 || n1) && !(n0 && n1 && p1 && p2 && p3)) return false;
    if ((p2 != nullptr) != (p3 != nullptr) || (p2 && !p1)) return false;
    static const bool off = [] { const char* e = LT_EXP_ENV("LT_OPEN_SMALL"); return e && e[0] == '0'; }();   // A/B
    if (off || n <= 0 || h <= 0 || wpr > 64 || opened == p0) return false;
    const dim3 grid((h + OS_RB - 1) / OS_RB, n);
    if (n0) hipLaunchKernelGGL(k_or_open5_small<6>, grid, dim3(256), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride);
    else if (p1 && !p2) hipLaunchKernelGGL(k_or_open5_small<2>, grid, dim3(256), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride);
    else if (p1) hipLaunchKernelGGL(k_or_open5_small<4>, grid, dim3(256), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride);
    else hipLaunchKernelGGL(k_or_open5_small<1>, grid, dim3(256), 0, s, p0, p1, p2, p3, n0, n1, opened, h, w, wpr, bits_stride);
    return true;
}

void launch_bits_to_u8(hipStream_t s, const unsigned long long* bits, uint8_t* out, int h, int w, size_t plane_stride,
                       size_t bits_stride, int n) {
    if (n <= 0 || h <= 0 || w <= 0) return;
    const int wpr = (w + 63) / 64;
    if ((w & 3) == 0 && (plane_stride & 3) == 0) {
        hipLaunchKernelGGL(k_bits_to_u8x4, dim3((h * (w >> 2) + 255) / 256, 1, n), dim3(256), 0, s, bits, out, h, w, wpr,
                           bits_stride, plane_stride);
        return;
    }
    hipLaunchKernelGGL(k_bits_to_u8, dim3((wpr + 3) / 4, h, n), dim3(256), 0, s, bits, out, h, w, wpr, bits_stride,
                       plane_stride);
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_threshold() {} }
void preload_k_threshold(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_threshold, dim3(1), dim3(1), 0, s); }

}  // namespace lt

