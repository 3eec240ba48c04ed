// cv2.adaptiveThreshold(plane, 255, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY, bs, -C) as one walk per plane, gfx950
// (lane_tracker.py:217-218: the 'neighborhood' filter of process()'s second try, on the RAW R and Lab-b planes):
//
//     pass(y, x)  <=>  p(y, x) - round(box_bs(y, x) / bs^2) > C,       box = sum over the bs x bs window, replicated border.
//
// k_adaptive_mean (k_filter.hip) evaluates the window per pixel (2 bs LDS reads per pixel: 12.6 ms per 256 frames for the
// second-try set).  Here the box sum is two running sums and costs the same for every window size:
//
//   * a wave walks DOWN a strip of 256 columns, four adjacent columns per lane (one dword load per lane and row, no
//     staging pass).  V[x] = sum of the bs rows around the current row, per column: V += row(y + r) - row(y - r - 1).
//   * the horizontal sum over x - r .. x + r is a difference of two entries of the inclusive prefix sum P of V along the
//     strip: three adds inside the lane, a six-step DPP scan of the lane totals across the wave, one 16-byte LDS store; then
//     every lane reads P[j + r] and P[j - r - 1] for the columns j = l, l + 64, l + 128 of the strip's 192 output columns
//     -- a lane-consecutive layout, so that the three v_cmp results ARE the three 64-bit words of the output row.
//   * no division:  p - round(S / A) > C  <=>  A p - S >= (A (2 C + 1) + 1) / 2   (A = bs^2 is odd, so S / A never ties
//     and the right-hand side is an integer).
//   * the rows of the window live in a ring of 2 r + 2 rows in LDS: the row that leaves the window and the centre pixels
//     come from there, not from memory again.
//
// Replicated border: a lane's dword lies wholly inside the image, wholly left or wholly right of it (width and strip origin
// are multiples of four), so clamping the address and one v_perm_b32 with a per-lane selector replicate the edge pixel;
// rows are clamped by address.  The strip's first 32 and last 32 columns are halo (window sizes up to 63).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "lt_internal.h"

namespace lt {
namespace {

typedef uint32_t __attribute__((may_alias)) u32a;
typedef uint8_t __attribute__((may_alias)) u8a;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((may_alias)) u128a;

constexpr int STRIP_IN = 256, STRIP_OUT = 192, HALO = 32;   // columns a wave loads / decides; halo on each side
constexpr int P_BYTES = STRIP_IN * 4;                       // the prefix sums of one row

struct BoxPlane {
    const uint8_t* src;          // u8 plane of frame 0 (row pitch = w)
    unsigned long long* out;     // bit plane of frame 0
    int r;                       // bs / 2
    int area;                    // bs * bs
    int c0;                      // (area * (2 C + 1) + 1) / 2
};

struct BoxArgs {
    BoxPlane pl[2];
    int nplanes;
    int h, w, wpr, nstrips, nbands, band_rows, nframes, ntasks;
    size_t plane_stride, bits_stride;
};

__device__ __forceinline__ int xcd_contiguous(int id, int n) {   // see k_threshold.hip
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ void lds_fence() {   // a wave's DS operations execute in order; this orders the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <class F>
__device__ __forceinline__ void static_for16(F&& f) { static_for_impl(f, std::make_integer_sequence<int, 16>{}); }

// Three wave-uniform 64-bit words into lane T of m[0..5]: v_mov_b32 under a one-lane EXEC mask (half the issue cost of
// v_writelane_b32, which also wants its lane select in M0).  Every lane is active around this statement (the control flow
// there is wave-uniform), so EXEC goes back to all ones; the words have passed through the scalar unit (an s_and), so no
// VALU-written SGPR is read by these VALU instructions (the compiler does not look inside inline assembly).
template <int T>
__device__ __forceinline__ void put_row(uint32_t (&m)[6], const unsigned long long (&word)[3]) {
    static_assert(T >= 0 && T < 32, "the EXEC mask is written as a 32-bit literal");
    asm volatile("s_mov_b64 exec, %12\n\t"
                 "v_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %8\n\t"
                 "v_mov_b32 %3, %9\n\tv_mov_b32 %4, %10\n\tv_mov_b32 %5, %11\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5])
                 : "s"((uint32_t)word[0]), "s"((uint32_t)(word[0] >> 32)), "s"((uint32_t)word[1]), "s"((uint32_t)(word[1] >> 32)),
                   "s"((uint32_t)word[2]), "s"((uint32_t)(word[2] >> 32)), "n"(1u << T));
}

// inclusive prefix sum over the 64 lanes: four shifts inside the rows of 16, two row broadcasts
__device__ __forceinline__ uint32_t wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

__global__ __launch_bounds__(64) void k_adaptive_box_walk(BoxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [P: 256 x u32][ring: (2 r + 2) x 256 bytes]
    const int lane = threadIdx.x;
    int task = __builtin_amdgcn_readfirstlane(xcd_contiguous(blockIdx.x, a.ntasks));
    const int strip = task % a.nstrips;  task /= a.nstrips;
    const int band = task % a.nbands;    task /= a.nbands;
    const int plane = task % a.nplanes;
    const int frame = task / a.nplanes;
    const BoxPlane pl = a.pl[plane];
    const int h = a.h, w = a.w, r = pl.r, R = 2 * r + 2;
    const int y0 = band * a.band_rows, y1 = min(y0 + a.band_rows, h);
    if (y0 >= y1) return;
    const int x_out0 = strip * STRIP_OUT, start = x_out0 - HALO;
    const int c = start + 4 * lane;
    // replicated columns: the dword of a lane is inside, left or right of the image as a whole
    const uint32_t sel = c < 0 ? 0x00000000u : (c >= w ? 0x03030303u : 0x03020100u);
    const bool edge = start < 0 || start + STRIP_IN > w;                          // wave-uniform
    const uint8_t* src = pl.src + (size_t)frame * a.plane_stride + min(max(c, 0), w - 4);
    auto load_row = [&](int q) -> uint32_t {
        const int row = min(max(q, 0), h - 1);
        uint32_t v = *reinterpret_cast<const uint32_t*>(src + (size_t)row * w);
        if (edge) v = __builtin_amdgcn_perm(0u, v, sel);
        return v;
    };
    unsigned char* ring = lds + P_BYTES;
    unsigned char* ring_lane = ring + 4 * lane;                                   // this lane's dword of a ring row
    const unsigned char* ring_ctr = ring + HALO + lane;                           // centre pixel of output column (64 i + lane)
    const unsigned char* pa = lds + 4 * (HALO + lane + r);                        // P[j + r]
    const unsigned char* pb = lds + 4 * (HALO + lane - r - 1);                    // P[j - r - 1]
    const int word0 = strip * (STRIP_OUT / 64);
    unsigned long long valid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int nv = w - (word0 + i) * 64;
        valid[i] = nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
    }

    // ---- prologue: the bs rows around row y0 - 1 into the ring (slot j = row y0 - 1 - r + j), their column sums into V:
    // the window of a virtual row y0 - 1, so that every real row is the same step ----
    uint32_t V0 = 0, V1 = 0, V2 = 0, V3 = 0;
    for (int j = 0; j <= 2 * r; ++j) {
        const uint32_t x = load_row(y0 - 1 - r + j);
        *reinterpret_cast<u32a*>(ring_lane + j * STRIP_IN) = x;
        V0 += x & 0xffu; V1 += (x >> 8) & 0xffu; V2 += (x >> 16) & 0xffu; V3 += x >> 24;
    }

    // Output: lane t of m[] collects the three words of row (group start + t); after 16 rows lanes 0..15 store them through
    // a buffer descriptor of the frame's bit plane.  Rows past the band, words past the row and the other 48 lanes get an
    // out-of-range offset, which the hardware drops: the stores are unconditional, so the compiler can count them and the
    // wait for a prefetched row stays a vmcnt(n) -- under a branch it becomes vmcnt(0) and the prefetch distance is gone.
    constexpr int RSRC_RAW = 0x00027000;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(pl.out + (size_t)frame * a.bits_stride, 0, h * a.wpr * 8, RSRC_RAW);
    uint32_t m[6] = {0, 0, 0, 0, 0, 0};
    // ring slot of row q = (q - (y0 - 1 - r)) mod R: the new row y + r takes the free slot, the old row y - r - 1 sits one further
    int s_new = 2 * r + 1;
    uint32_t nq[4];                                           // rows y + r of the next four steps, in flight
#pragma unroll
    for (int k = 0; k < 4; ++k) nq[k] = load_row(y0 + k + r);
    for (int yb = y0; yb < y1; yb += 16) {                    // groups of 16 rows; the last one may run past the band (nothing of it is stored)
        static_for16([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int y = yb + t;
            const uint32_t x = nq[t & 3];
            nq[t & 3] = load_row(y + 4 + r);
            const int s_old = s_new + 1 == R ? 0 : s_new + 1;
            const uint32_t o = *reinterpret_cast<const u32a*>(ring_lane + s_old * STRIP_IN);
            lds_fence();
            *reinterpret_cast<u32a*>(ring_lane + s_new * STRIP_IN) = x;
            V0 += (x & 0xffu) - (o & 0xffu);
            V1 += ((x >> 8) & 0xffu) - ((o >> 8) & 0xffu);
            V2 += ((x >> 16) & 0xffu) - ((o >> 16) & 0xffu);
            V3 += (x >> 24) - (o >> 24);
            int s_ctr = s_new - r;                            // slot of row y
            if (s_ctr < 0) s_ctr += R;
            s_new = s_old;
            // prefix sums of V along the strip
            const uint32_t L1 = V0 + V1, L2 = L1 + V2, L3 = L2 + V3;
            const uint32_t incl = wave_scan(L3), off = incl - L3;
            u32x4 P;
            P.x = off + V0; P.y = off + L1; P.z = off + L2; P.w = incl;
            *reinterpret_cast<u128a*>(lds + 16 * lane) = P;
            lds_fence();
            unsigned long long word[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t hi = *reinterpret_cast<const u32a*>(pa + 256 * i);
                const uint32_t lo = *reinterpret_cast<const u32a*>(pb + 256 * i);
                const uint32_t v = *reinterpret_cast<const u8a*>(ring_ctr + s_ctr * STRIP_IN + 64 * i);
                const int u = (int)(__umul24(v, (uint32_t)pl.area) + lo - hi);   // A p - S
                word[i] = __ballot(u >= pl.c0) & valid[i];
            }
            lds_fence();
            put_row<t>(m, word);
        });
        const int row = yb + lane;
        const uint32_t base = (lane < 16 && row < y1) ? (uint32_t)(row * a.wpr + word0) * 8u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 v;
            v.x = m[2 * i]; v.y = m[2 * i + 1];
            __builtin_amdgcn_raw_buffer_store_b64(v, out_rs, (int)(word0 + i < a.wpr ? base + 8u * i : 0x80000000u), 0, 0);
        }
    }
}

int g_bands_override = [] { const char* e = LT_EXP_ENV("LT_BOX_BANDS"); return e ? std::atoi(e) : 0; }();

}  // namespace

bool adaptive_walk_supported(int bs_r, int bs_b, int h, int w, size_t plane_stride) {
    static const bool off = [] { const char* e = LT_EXP_ENV("LT_ADAPTIVE_TILES"); return e && e[0] == '1'; }();
    if (off) return false;
    auto ok = [](int bs) { return (bs & 1) && bs >= 1 && bs <= 2 * HALO - 1; };
    return ok(bs_r) && ok(bs_b) && (w & 3) == 0 && w >= 4 && h >= 1 && (plane_stride & 3) == 0;
}

// one launch over `np` planes (1 or 2)
static bool launch_box(hipStream_t s, const BoxPlane* planes, int np, int h, int w, size_t plane_stride, size_t bits_stride, int n) {
    BoxArgs a;
    int rmax = 0;
    for (int i = 0; i < np; ++i) { a.pl[i] = planes[i]; rmax = std::max(rmax, planes[i].r); }
    if (np == 1) a.pl[1] = planes[0];
    a.nplanes = np;
    a.h = h; a.w = w; a.wpr = (w + 63) / 64;
    a.nstrips = (w + STRIP_OUT - 1) / STRIP_OUT;
    a.nframes = n;
    a.plane_stride = plane_stride;
    a.bits_stride = bits_stride;
    const size_t lds = (size_t)P_BYTES + (size_t)(2 * rmax + 2) * STRIP_IN;
    // bands: every task pays a prologue of bs rows (about a third of a walked row each); enough tasks to fill the chip
    // a few times over, none shorter than 4 r rows
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long slots = (long long)cus * std::max<long long>(1, std::min<long long>(32, (160 * 1024) / (long long)lds));
    int best_nb = 1;
    double best_cost = 1e300;
    for (int nb = 1; nb <= 64; ++nb) {
        const int rows = (h + nb - 1) / nb;
        const int real_nb = (h + rows - 1) / rows;
        if (real_nb != nb) continue;
        const long long tasks = (long long)np * n * a.nstrips * nb;
        if (nb > 1 && rows < std::max(8, (tasks <= slots ? 1 : 4) * rmax)) break;
        const double per_task = rows + (2 * rmax + 1) * 0.35;
        const double cost = tasks < slots ? per_task : (double)((tasks + slots - 1) / slots) * per_task;
        if (cost < best_cost - 1e-9) { best_cost = cost; best_nb = nb; }
    }
    // measured on launches that fill the chip several times over (256 frames, a sweep of 3 .. 32 bands, each twice): 10 bands of
    // 110 rows are fastest, 0.336 ms for both planes against 0.36 for the model's choice and 0.37-0.46 for 16 .. 32 bands
    if ((long long)np * n * a.nstrips * 10 >= 2 * slots && h / 10 >= 2 * std::max(rmax, 4)) best_nb = 10;
    if (g_bands_override > 0) best_nb = std::min(g_bands_override, h);
    a.band_rows = (h + best_nb - 1) / best_nb;
    a.nbands = (h + a.band_rows - 1) / a.band_rows;
    a.ntasks = n * a.nplanes * a.nbands * a.nstrips;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_adaptive_box_walk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(k_adaptive_box_walk, dim3(a.ntasks), dim3(64), lds, s, a);
    return true;
}

// both planes of the 'neighborhood' filter: out_r / out_b are the two thresholded bit planes
bool launch_adaptive_walk(hipStream_t s, const uint8_t* R, int bs_r, int C_r, unsigned long long* out_r, const uint8_t* B, int bs_b,
                          int C_b, unsigned long long* out_b, int h, int w, size_t plane_stride, size_t bits_stride, int n) {
    if (n <= 0) return true;
    if (!adaptive_walk_supported(bs_r, bs_b, h, w, plane_stride)) return false;
    auto fill = [](BoxPlane& p, const uint8_t* src, unsigned long long* out, int bs, int C) {
        p.src = src; p.out = out; p.r = bs / 2; p.area = bs * bs;
        const long long t = (long long)p.area * (2LL * C + 1) + 1;
        p.c0 = (int)(t / 2);       // exact: area and 2 C + 1 are odd
    };
    if (std::llabs((long long)bs_r * bs_r * (2LL * C_r + 1)) > (1ll << 30) || std::llabs((long long)bs_b * bs_b * (2LL * C_b + 1)) > (1ll << 30)) return false;
    BoxPlane pl[2];
    fill(pl[0], R, out_r, bs_r, C_r);
    fill(pl[1], B, out_b, bs_b, C_b);
    // One launch per plane, each with the ring its own window needs (a 15-pixel window: 5 KB per wave instead of the 10 KB a
    // 35-pixel window beside it imposes): 0.336 against 0.352 ms per 256 frames for one launch over both (LT_BOX_SPLIT=0).  A
    // few dozen frames cannot fill the chip either way and take the single launch (one tail instead of two: 59 against 66 us at 32).
    static const bool split = [] { const char* e = LT_EXP_ENV("LT_BOX_SPLIT"); return !(e && e[0] == '0'); }();
    if (split && n >= 64) return launch_box(s, pl, 1, h, w, plane_stride, bits_stride, n) && launch_box(s, pl + 1, 1, h, w, plane_stride, bits_stride, n);
    return launch_box(s, pl, 2, h, w, plane_stride, bits_stride, n);
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_adaptive_walk() {} }
void preload_k_adaptive_walk(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_adaptive_walk, dim3(1), dim3(1), 0, s); }

}  // namespace lt

