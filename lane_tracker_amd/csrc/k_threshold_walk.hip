// bilateral_adaptive_threshold (lane_tracker.py:14-83) as long running-sum walks, gfx950.
//
//     pass(y, x)  <=>  (k p > S_left + C k  and  k p > S_right + C k)  or  (k p > S_up + C k  and  k p > S_down + C k)
//
// k_bilateral_tile2 (k_threshold.hip) evaluates this on 128 x 128 tiles: every wave walks 32 pixels, pays a
// k-long prologue for them, re-stages a 2k-wide halo per tile and synchronises four times per plane.  Here a wave owns
// 128 rows (horizontal pass) or 128 columns (vertical pass) and walks a long segment of them -- half an image row or
// column -- with its own little ring of pixels in LDS and no barrier at all:
//
//   * lanes hold two pixels per VGPR (rows l | l+64, or columns l | l+64) as 16-bit halves.  The running sums are
//     updated with plain 32-bit v_add_u32 / v_sub_u32 -- measured 2 cycles per wave64 instruction on gfx950 against 4
//     for every packed (VOP3P) form, profiles/r02_valu_issue_table.txt -- which is exact because no carry or borrow
//     ever crosses bit 16:  sum + bias < 2^16 and every intermediate stays >= 0.
//   * the verdict needs no max and no compare:  with  bias = 0x8000 + C k  folded into both sums,
//         D = (S + bias) - k p       has bit 15 set   <=>   S + C k >= k p   <=>  this side fails,
//     so  (D_a | D_b)  carries "fails" in bits 15 / 31.
//   * horizontal pass: the fail bits are shifted into a packed pair of 16-bit registers (v_lshrrev + v_bfi / v_bitop3)
//     and every 16 pixels into the lane's two 64-bit row words; vertical pass: two v_cmp give the row words of a row.
//   * staging: coalesced loads one chunk / block ahead, written into a ring of 2k + 18 positions; all ring offsets are
//     compile-time constants because the walk is unrolled over one ring length.
//
// The planes these kernels read (the two top-hat planes) have a row pitch that is a multiple of 64 bytes, so that every
// staging load covers whole, aligned lines.  The four passes (R / Lab-b top-hat x horizontal / vertical) write four
// partial bit planes; k_or4_bits merges them.  Window sizes are template parameters (15, 20, 35: process() defaults,
// second try and the documented settings of tracker_settings.md); any other size, the greenery mask (mask_noise) or a
// width that is not a multiple of 4 takes k_bilateral_tile2.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "lt_internal.h"

namespace lt {
namespace {

template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

__device__ __forceinline__ int xcd_contiguous(int id, int n) {   // see k_threshold.hip
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct WalkArgs {
    const uint8_t* src;          // u8 plane of frame 0
    unsigned long long* out;     // partial bit plane of frame 0
    int h, w, wpr;
    int pitch;                   // bytes per row of the plane (a multiple of 64 >= w)
    int C;
    uint32_t rbias;              // RANGE walks: (0x8000 - noise_thresh) in both halves (see walk_h_task)
    int groups;                  // 128-row groups (horizontal) / 128-column groups (vertical) per frame
    int segs, seg_len;           // segments per walk, nominal length (horizontal: a multiple of 128)
    int ntasks;
    size_t plane_stride, bits_stride;
};

template <int K>
struct WalkCfg {                                                    // vertical pass
    static constexpr int NPRE = (2 * K + 2 + 15) / 16;              // chunks in the ring before the first step
    static constexpr int WIN = ((2 * K + 18 + 15) / 16) * 16;       // ring length in pixels
    static constexpr int NCH = WIN / 16;
    static constexpr int WSTEP = 16 * NPRE - 2 * K - 2;             // step (mod 16) in front of which the next chunk lands
    static constexpr int V_LDS = WIN * 128;
};

// Horizontal pass: the ring origin is the 128-aligned column at or before the first pixel the walk needs, so that a
// staging load covers whole 128-byte lines of 8 rows (L2 fetches whole lines: with 64-byte pieces every line crossed the
// fabric twice).  Segments start at multiples of 128.  OFF = ring position of that first pixel (ys - K).
template <int K>
struct WalkCfgH {
    static constexpr int E = (4 - (K & 3)) & 3;                     // pixels walked before the segment start
    static constexpr int OFF = ((-(E + K)) % 128 + 128) % 128;
    static constexpr int NPRE = (OFF + 2 * K + 2 + 15) / 16;
    static constexpr int WIN = ((2 * K + 18 + 15) / 16) * 16;
    static constexpr int NCH = WIN / 16;
    static constexpr int WSTEP = 16 * NPRE - (OFF + 2 * K + 2);
#ifndef LT_WALK65_PITCH_PAD
#define LT_WALK65_PITCH_PAD 0
#endif
    // bytes per ring row, 16 x odd: conflict-free 128-bit reads.  Window 65 goes without the pad: 20 KB instead of 22.5 KB per
    // wave = 8 waves per CU instead of 7, which is worth more than the two-way conflicts on those (rare) reads cost -- the
    // greenery-mask walk 0.28 -> 0.26 ms per 256 frames (two A/B pairs; -DLT_WALK65_PITCH_PAD=1 for the padded form).
    static constexpr int PITCH = ((NCH % 2) || (K > 35 && !LT_WALK65_PITCH_PAD)) ? WIN : WIN + 16;
    static constexpr int LDS = 128 * PITCH;
    static_assert(NPRE >= 9 && NPRE <= 15, "the prologue loads two 128-column blocks and leaves the second one in registers");
};

// The ring is written as dwords / dword quads and read as dwords, quads or 16-bit pairs: every LDS access goes through
// may_alias types, or type-based alias analysis lets the compiler move the reads across the writes.
typedef uint32_t __attribute__((may_alias)) u32a;
typedef uint16_t __attribute__((may_alias)) u16a;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((may_alias)) u128a;
__device__ __forceinline__ uint32_t lds_u32(const uint8_t* p) { return *reinterpret_cast<const u32a*>(p); }

// Two wave-uniform 64-bit masks into lane LANE (< 32) of four VGPRs: v_mov_b32 under a one-lane EXEC mask does what
// v_writelane_b32 does at half the issue cost (2 cycles against 4, profiles/r02_valu_issue_table.txt).  Every lane of
// the wave is active around this statement, so EXEC is restored to all ones.  The masks must not come from a VALU
// instruction issued in the last few cycles (a VALU-written SGPR needs wait states before another VALU instruction
// reads it, and the compiler does not look inside inline assembly): the caller passes masks that are a whole step old.
template <int LANE>
__device__ __forceinline__ void put_lane(uint32_t& m0, uint32_t& m1, uint32_t& m2, uint32_t& m3, unsigned long long a,
                                         unsigned long long b) {
    static_assert(LANE >= 0 && LANE < 32, "the EXEC mask is written as a 32-bit literal");
    asm volatile("s_mov_b64 exec, %8\n\t"
                 "v_mov_b32 %0, %4\n\t"
                 "v_mov_b32 %1, %5\n\t"
                 "v_mov_b32 %2, %6\n\t"
                 "v_mov_b32 %3, %7\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                 : "s"((uint32_t)a), "s"((uint32_t)(a >> 32)), "s"((uint32_t)b), "s"((uint32_t)(b >> 32)), "n"(1u << LANE));
}

struct Task {
    int seg, grp, frame, y0, y1;
};
// blk: the workgroup's id in launch order (renumbered here so that every XCD owns a contiguous range of tasks), or -- with
// PLACED -- a task number the caller has already placed (k_bilateral_walk_hv)
template <bool PLACED = false>
__device__ __forceinline__ bool decode_task(const WalkArgs& a, int len, Task& t, int blk) {
    const int task = PLACED ? blk : xcd_contiguous(blk, a.ntasks);
    t.seg = task % a.segs;
    t.grp = (task / a.segs) % a.groups;
    t.frame = task / (a.segs * a.groups);
    t.y0 = t.seg * a.seg_len;
    if (t.y0 >= len) return false;
    t.y1 = t.y0 + a.seg_len;
    if (t.seg == a.segs - 1 || t.y1 > len) t.y1 = len;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Vertical pass: lanes = columns (l | l + 64) of a 128-column group, the walk goes down the rows.
template <int K, bool PLACED = false>
__device__ __forceinline__ void walk_v_task(const WalkArgs& a, int blk) {
    using Cfg = WalkCfg<K>;
    constexpr int WIN = Cfg::WIN, NPRE = Cfg::NPRE, NCH = Cfg::NCH, WSTEP = Cfg::WSTEP;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int lane = threadIdx.x;
    const int h = a.h, w = a.w;
    Task t;
    if (!decode_task<PLACED>(a, h, t, blk)) return;
    const uint8_t* src = a.src + (size_t)t.frame * a.plane_stride;
    unsigned long long* out = a.out + (size_t)t.frame * a.bits_stride;
    const int y0 = t.y0, y1 = t.y1;
    const int xs = y0 - K;                            // row at ring position 0
    const int total = y1 - y0;                        // steps
    const int g0 = t.grp * 128;                       // first column of the group

    // staging: lane -> (row lane>>4 + 4 i, dwords lane&15 of both 64-column halves): 4 rows x one 128-byte line per pair of loads
    const int ca = g0 + 4 * (lane & 15), cb = ca + 64;
    const bool ain = ca < w, bin = cb < w;
    const uint8_t *pa = src + min(ca, w - 4), *pb = src + min(cb, w - 4);
    auto issue = [&](int chunk, uint32_t (&st)[8]) {  // chunk c covers rows xs + 16 c .. + 15
        const int p0 = xs + 16 * chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = p0 + (lane >> 4) + 4 * i;
            const uint32_t ro = (uint32_t)__mul24(min(max(row, 0), h - 1), a.pitch);
            st[2 * i] = *reinterpret_cast<const uint32_t*>(pa + ro);
            st[2 * i + 1] = *reinterpret_cast<const uint32_t*>(pb + ro);
        }
    };
    auto land = [&](int chunk, int slot, const uint32_t (&st)[8]) {   // ring slot = chunk mod NCH (a constant at every call site)
        const int p0 = xs + 16 * chunk;
        u32a* dst = reinterpret_cast<u32a*>(ring + (slot * 16 + (lane >> 4)) * 128 + 8 * (lane & 15));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = p0 + (lane >> 4) + 4 * i;
            const bool yin = row >= 0 && row < h;
            const uint32_t va = (ain && yin) ? st[2 * i] : 0u, vb = (bin && yin) ? st[2 * i + 1] : 0u;
            // byte 2c = column c, byte 2c+1 = column c + 64 of the group: one 16-bit read per lane in the walk
            dst[i * 4 * 128 / 4] = __builtin_amdgcn_perm(vb, va, 0x05010400u);
            dst[i * 4 * 128 / 4 + 1] = __builtin_amdgcn_perm(vb, va, 0x07030602u);
        }
    };
    auto vraw = [&](auto pc) -> uint32_t {            // bytes (column l, column l+64) of ring row P
        constexpr int P = decltype(pc)::value;
        return *reinterpret_cast<const u16a*>(ring + P * 128 + 2 * lane);
    };
    auto spread = [](uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x0c010c00u); };   // -> (column l | column l+64 << 16)
    auto vread = [&](auto pc) -> uint32_t { return spread(vraw(pc)); };

    // ---- prologue: fill the ring, initial sums ----
    uint32_t st[8];
    {
        uint32_t pre[NPRE][8];                        // every load of the prologue in flight before the first ring write
        static_for([&](auto cc) { issue(decltype(cc)::value, pre[decltype(cc)::value]); }, std::make_integer_sequence<int, NPRE>{});
        issue(NPRE, st);                              // lands in front of step WSTEP
        static_for([&](auto cc) { land(decltype(cc)::value, decltype(cc)::value % NCH, pre[decltype(cc)::value]); },
                   std::make_integer_sequence<int, NPRE>{});
    }
    const uint32_t kbias = (uint32_t)(0x8000 + a.C * K) * 0x10001u;
    uint32_t sl = kbias, sr = kbias, qc;
    static_for([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        sl += vread(std::integral_constant<int, p>{});                    // rows y0-K .. y0-1
        sr += vread(std::integral_constant<int, K + 1 + p>{});            // rows y0+1 .. y0+K
    }, std::make_integer_sequence<int, K>{});
    qc = vread(std::integral_constant<int, K>{});
    // The ring reads of a step are issued one step ahead: the inline-assembly statements of a step keep the compiler
    // from hoisting them, and a read issued right in front of its use costs the LDS latency on every step.
    uint32_t ro = vraw(std::integral_constant<int, 0>{}), rn = vraw(std::integral_constant<int, (K + 1) % WIN>{}),
             ri = vraw(std::integral_constant<int, (2 * K + 1) % WIN>{});

    // ---- the walk ----
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;          // lane t = the two column words of row (16-row batch start + t)
    const int wordv = t.grp * 2;                      // first of the group's two words in a row
    unsigned long long pva = 0, pvb = 0;              // pass masks of the previous row
    auto store_rows = [&](int row0) {                 // lanes 0..15 hold rows row0 .. row0 + 15
        const int row = row0 + lane;
        if (lane < 16 && row >= y0 && row < y1) {
            unsigned long long* o = out + (size_t)row * a.wpr + wordv;
            o[0] = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
            if (wordv + 1 < a.wpr) o[1] = (unsigned long long)m2 | ((unsigned long long)m3 << 32);
        }
    };
    for (int sb = 0; sb < total; sb += WIN) {
        static_for([&](auto gc) {                     // 16-step groups of one ring length
            constexpr int g = decltype(gc)::value;
            if (sb + 16 * g < total) {                // wave-uniform
                static_for([&](auto jc) {
                    constexpr int u = 16 * g + decltype(jc)::value;        // step inside the ring length, s = sb + u
                    if constexpr (u % 16 == WSTEP) {
                        // the chunk that step u + 1 starts to read replaces the 16 positions the walk has left behind
                        land((sb >> 4) + g + NPRE, (g + NPRE) % NCH, st);
                        issue((sb >> 4) + g + NPRE + 1, st);
                    }
                    // this step's pixels were read during the previous step; read the next step's now (after the landing)
                    constexpr int PO = (u + 1) % WIN, PN = (u + K + 2) % WIN, PI = (u + 2 * K + 2) % WIN;
                    const uint32_t ro2 = vraw(std::integral_constant<int, PO>{}), rn2 = vraw(std::integral_constant<int, PN>{}),
                                   ri2 = vraw(std::integral_constant<int, PI>{});
                    const uint32_t qo = spread(ro), qn = spread(rn), qi = spread(ri);
                    ro = ro2; rn = rn2; ri = ri2;
                    // verdict of row y0 + s: both columns at once, "fails" in bits 15 and 31
                    const uint32_t kp = __umul24(qc, (uint32_t)K);
                    const uint32_t v = (sl - kp) | (sr - kp);
                    // The masks of row s-1 go into lane (s-1) % 16 now: a whole step lies between the compares that
                    // wrote the SGPRs and this point (see put_lane).
                    constexpr int tp = (u + 15) % 16;
                    put_lane<tp>(m0, m1, m2, m3, pva, pvb);
                    if constexpr (tp == 15) store_rows(y0 + sb + u - 16);               // the 16 rows before this one
                    asm volatile("v_cmp_le_i16_e64 %0, 0, %1" : "=s"(pva) : "v"(v));   // bit 15 clear: column l passes
                    asm volatile("v_cmp_le_i32_e64 %0, 0, %1" : "=s"(pvb) : "v"(v));   // bit 31 clear: column l + 64 passes
                    // slide both windows by one row
                    sl = sl + qc - qo;
                    sr = sr + qi - qn;
                    qc = qn;
                }, std::make_integer_sequence<int, 16>{});
            }
        }, std::make_integer_sequence<int, NCH>{});
    }
    // steps run in groups of 16, so the walk ended on a group boundary G >= total: row G-1's masks are still in
    // SGPRs and the batch [G-16, G) has not been stored (rows >= y1 are dropped by store_rows)
    const int G = (total + 15) & ~15;
    asm volatile("s_nop 4");                          // the last compares may be only a few instructions back
    put_lane<15>(m0, m1, m2, m3, pva, pvb);
    store_rows(y0 + G - 16);
}

// ---------------------------------------------------------------------------------------------------------------
// Horizontal pass: lanes = rows (l | l + 64) of a 128-row group, the walk goes along the columns.
// Staging: a block = 128 columns of the 128 rows, sixteen 16-byte loads per lane (lane -> row lane>>3 + 8 i, 16-byte
// piece lane&7: 8 rows x one whole 128-byte line per instruction); the eighth of the lanes that holds the next 16 columns
// writes them into the ring every 16 steps with 128-bit stores, and the streams read 128 bits per row and 16 steps.
// (64 VGPRs for the block: the ring limits a CU to 11 waves at k = 35 anyway, so up to 168 VGPRs cost no occupancy.)
// RANGE: the greenery mask of filter_lane_points (lane_tracker.py:223-225),  noise = !inRange(b, T, 255) | bilateral(b),
// folded into this pass: a pixel below T passes whatever its sums say.  With rbias = 0x8000 - T in both halves (T clamped
// to [0, 256]) bit 15 of  q + rbias  is set exactly when q >= T, so  fails &= q + rbias  (one add, one and per step; the
// horizontal walks have the issue slots to spare, the vertical ones do not).
template <int K, bool PLACED = false, bool RANGE = false>
__device__ __forceinline__ void walk_h_task(const WalkArgs& a, int task_blk) {
    using Cfg = WalkCfgH<K>;
    constexpr int WIN = Cfg::WIN, NPRE = Cfg::NPRE, NCH = Cfg::NCH, WSTEP = Cfg::WSTEP, PITCH = Cfg::PITCH, E = Cfg::E,
                  OFF = Cfg::OFF;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int lane = threadIdx.x;
    const int h = a.h, w = a.w;
    Task t;
    if (!decode_task<PLACED>(a, w, t, task_blk)) return;
    const uint8_t* src = a.src + (size_t)t.frame * a.plane_stride;
    unsigned long long* out = a.out + (size_t)t.frame * a.bits_stride;
    const int y0 = t.y0;
    const int yend = (t.y1 + 63) & ~63;               // whole 64-pixel words
    const int ys = y0 - E;                            // first pixel walked (its verdict is dropped when < y0)
    const int xs = ys - K - OFF;                      // pixel at ring position 0: a multiple of 128
    const int total = yend - ys;                      // steps
    const int g0 = t.grp * 128;                       // first row of the group

    const int piece = lane & 7, rsub = lane >> 3;     // lane -> (row rsub + 8 i, 16-byte piece of a 128-byte line)
    auto rowoff = [&](int i) { return (uint32_t)__mul24(min(g0 + rsub + 8 * i, h - 1), a.pitch); };
    auto issue_block = [&](int blk, u32x4 (&r)[16]) { // block b covers pixels xs + 128 b .. + 127
        // (block 0: the pieces in front of ring position OFF are never read; they re-read the first live piece's line)
        const int col = max(xs + 128 * blk + 16 * piece, xs + (OFF & ~15));
        const uint8_t* colp = src + min(max(col, 0), a.pitch - 16);   // 128-byte aligned lines of 64-byte aligned rows
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = *reinterpret_cast<const u32x4*>(colp + rowoff(i));
    };
    // chunk c = pixels xs + 16 c .. + 15; slot = c mod NCH (a constant at every call site).  The eighth of the lanes
    // whose piece it is writes it; chunks that lie inside the image and row groups that lie inside it skip the masks.
    const bool rows_in = g0 + 128 <= h;
    auto land = [&](int chunk, int slot, const u32x4 (&r)[16]) {
        if (piece == (chunk & 7)) {
            const int col = xs + 16 * chunk;
            unsigned char* dst = ring + rsub * PITCH + slot * 16;
            if (rows_in && col >= 0 && col + 16 <= w) {        // wave-uniform
#pragma unroll
                for (int i = 0; i < 16; ++i) *reinterpret_cast<u128a*>(dst + 8 * i * PITCH) = r[i];
            } else {
                const bool in0 = col >= 0 && col < w, in1 = col + 4 >= 0 && col + 4 < w, in2 = col + 8 >= 0 && col + 8 < w,
                           in3 = col + 12 >= 0 && col + 12 < w;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool rin = g0 + rsub + 8 * i < h;
                    u32x4 v = r[i];
                    v.x = (rin && in0) ? v.x : 0u;
                    v.y = (rin && in1) ? v.y : 0u;
                    v.z = (rin && in2) ? v.z : 0u;
                    v.w = (rin && in3) ? v.w : 0u;
                    *reinterpret_cast<u128a*>(dst + 8 * i * PITCH) = v;
                }
            }
        }
    };
    struct Stream { u32x4 a, b; };                    // the 16 ring positions around the stream's current one, rows l and l + 64
    auto fetch = [&](Stream& s, auto pc) {
        constexpr int P = decltype(pc)::value;
        const uint8_t* base = ring + lane * PITCH + (P & ~15);
        s.a = *reinterpret_cast<const u128a*>(base);
        s.b = *reinterpret_cast<const u128a*>(base + 64 * PITCH);
    };
    auto pick = [&](const Stream& s, auto pc) -> uint32_t {   // (row l | row l+64 << 16) at ring position P
        constexpr int P = decltype(pc)::value;
        constexpr uint32_t B = P & 3;
        constexpr int comp = (P & 15) >> 2;
        return __builtin_amdgcn_perm(s.b[comp], s.a[comp], 0x0c000c00u | B | ((4u + B) << 16));
    };

    // ---- prologue: block 0 (its chunks below OFF / 16 are never read), then block 1, which stays in registers ----
    u32x4 blk[16];
    issue_block(0, blk);
    static_for([&](auto cc) {
        constexpr int c = OFF / 16 + decltype(cc)::value;
        land(c, c % NCH, blk);
    }, std::make_integer_sequence<int, 8 - OFF / 16>{});
    issue_block(1, blk);
    static_for([&](auto cc) {
        constexpr int c = 8 + decltype(cc)::value;
        land(c, c % NCH, blk);
    }, std::make_integer_sequence<int, NPRE - 8>{});
    const uint32_t kbias = (uint32_t)(0x8000 + a.C * K) * 0x10001u;
    uint32_t sl = kbias, sr = kbias, qc;
    Stream so, sn, si;                                // out (position OFF + s), next (+ K + 1), in (+ 2K + 1)
    Stream so2, sn2, si2;                             // their next 16 positions, read one step early
    {
        // sl = positions OFF .. OFF+K-1, sr = positions OFF+K+1 .. OFF+2K, dword by dword with v_sad_u8
        uint32_t la = 0, lb = 0, ra = 0, rb = 0;
        static_for([&](auto dc) {
            constexpr int d = OFF / 4 + decltype(dc)::value;     // dword index along the walk
            constexpr int lo = d * 4, hi = lo + 4;
            auto part = [&](int from, int to, uint32_t& accA, uint32_t& accB) {   // bytes of this dword inside [from, to)
                const int f = from > lo ? from : lo, tt = to < hi ? to : hi;
                if (f >= tt) return;
                const uint32_t mask = (tt - lo == 4 ? 0xffffffffu : ((1u << (8 * (tt - lo))) - 1u)) & ~((1u << (8 * (f - lo))) - 1u);
                const uint8_t* base = ring + lane * PITCH + lo % WIN;   // positions beyond the ring length wrap (WIN % 4 == 0)
                accA = __builtin_amdgcn_sad_u8(lds_u32(base) & mask, 0u, accA);
                accB = __builtin_amdgcn_sad_u8(lds_u32(base + 64 * PITCH) & mask, 0u, accB);
            };
            part(OFF, OFF + K, la, lb);
            part(OFF + K + 1, OFF + 2 * K + 1, ra, rb);
        }, std::make_integer_sequence<int, (OFF % 4 + 2 * K + 4) / 4>{});
        sl += la | (lb << 16);
        sr += ra | (rb << 16);
        fetch(sn, std::integral_constant<int, (OFF + K) % WIN>{});
        qc = pick(sn, std::integral_constant<int, (OFF + K) % WIN>{});
        fetch(so, std::integral_constant<int, OFF % WIN>{});
        fetch(sn, std::integral_constant<int, (OFF + K + 1) % WIN>{});
        fetch(si, std::integral_constant<int, (OFF + 2 * K + 1) % WIN>{});
        so2 = so; sn2 = sn; si2 = si;                 // step 0 may start a stream on a block boundary
    }

    // ---- the walk ----
    uint32_t acc = 0;                                 // 16 fail bits per half
    unsigned long long wa = 0, wb = 0;                // the row words being filled (rows lane, lane + 64)
    unsigned long long wa0 = 0, wb0 = 0;              // the even word of a pair (two words = one 16-byte store)
    int nflush = 0, word = y0 >> 6;
    const int rowa = g0 + lane, rowb = rowa + 64;
    struct __attribute__((aligned(8))) Pair { unsigned long long lo, hi; };
    for (int sb = 0; sb < total; sb += WIN) {
        static_for([&](auto gc) {                     // 16-step groups of one ring length
            constexpr int g = decltype(gc)::value;
            if (sb + 16 * g < total) {                // wave-uniform
                static_for([&](auto jc) {
                    constexpr int u = 16 * g + decltype(jc)::value;        // step inside the ring length, s = sb + u
                    if constexpr (u % 16 == WSTEP) {
                        // the chunk that step u + 1 starts to read replaces the 16 positions the walk has left behind
                        const int c = (sb >> 4) + g + NPRE;
                        land(c, (g + NPRE) % NCH, blk);
                        if ((c & 7) == 7) issue_block((c >> 3) + 1, blk);   // wave-uniform: the block is used up
                    }
                    constexpr int PO = (u + OFF) % WIN, PN = (u + OFF + K + 1) % WIN, PI = (u + OFF + 2 * K + 1) % WIN;
                    // a stream's next 16 positions are read during the step before it enters them (for the in stream
                    // that is the step whose landing, just above, brought them); the switch is a register rename
                    if constexpr (PO % 16 == 0) so = so2;
                    if constexpr (PN % 16 == 0) sn = sn2;
                    if constexpr (PI % 16 == 0) si = si2;
                    if constexpr ((PO + 1) % 16 == 0) fetch(so2, std::integral_constant<int, (PO + 1) % WIN>{});
                    if constexpr ((PN + 1) % 16 == 0) fetch(sn2, std::integral_constant<int, (PN + 1) % WIN>{});
                    if constexpr ((PI + 1) % 16 == 0) fetch(si2, std::integral_constant<int, (PI + 1) % WIN>{});
                    const uint32_t qo = pick(so, std::integral_constant<int, PO>{});
                    const uint32_t qn = pick(sn, std::integral_constant<int, PN>{});
                    const uint32_t qi = pick(si, std::integral_constant<int, PI>{});
                    // verdict of pixel ys + s: both rows at once, "fails" in bits 15 and 31
                    const uint32_t kp = __umul24(qc, (uint32_t)K);
                    uint32_t v = (sl - kp) | (sr - kp);
                    if (RANGE) v &= qc + a.rbias;
                    acc = (v & 0x80008000u) | ((acc >> 1) & 0x7fff7fffu);
                    if constexpr ((u - E + 16 * 64) % 16 == 15) if (sb + u >= E + 15) {   // 16 pixels of the segment are complete
                        wa = (wa >> 16) | ((unsigned long long)(acc & 0xffffu) << 48);
                        wb = (wb >> 16) | ((unsigned long long)(acc >> 16) << 48);
                        if ((++nflush & 3) == 0) {                            // a word is complete
                            if ((nflush & 4) != 0) {                          // the first of a pair: keep it
                                wa0 = wa;
                                wb0 = wb;
                            } else {                                          // words word-1, word: one 16-byte store per row
                                if (word < a.wpr) {
                                    if (rowa < h) *reinterpret_cast<Pair*>(out + (size_t)rowa * a.wpr + word - 1) = Pair{~wa0, ~wa};
                                    if (rowb < h) *reinterpret_cast<Pair*>(out + (size_t)rowb * a.wpr + word - 1) = Pair{~wb0, ~wb};
                                } else if (word - 1 < a.wpr) {
                                    if (rowa < h) out[(size_t)rowa * a.wpr + word - 1] = ~wa0;
                                    if (rowb < h) out[(size_t)rowb * a.wpr + word - 1] = ~wb0;
                                }
                            }
                            ++word;
                        }
                    }
                    // slide both windows by one pixel
                    sl = sl + qc - qo;
                    sr = sr + qi - qn;
                    qc = qn;
                }, std::make_integer_sequence<int, 16>{});
            }
        }, std::make_integer_sequence<int, NCH>{});
    }
    if ((nflush & 7) == 4 && word - 1 < a.wpr) {      // an odd number of words: the last one is still held back
        if (rowa < h) out[(size_t)rowa * a.wpr + word - 1] = ~wa0;
        if (rowb < h) out[(size_t)rowb * a.wpr + word - 1] = ~wb0;
    }
}

template <int K>
__global__ __launch_bounds__(64) void k_bilateral_walk_v(WalkArgs a) { walk_v_task<K>(a, blockIdx.x); }
template <int K>
__global__ __launch_bounds__(64, 3) void k_bilateral_walk_h(WalkArgs a) { walk_h_task<K>(a, blockIdx.x); }   // 3 waves per SIMD: <= 168 VGPRs

// Both passes of a plane in ONE launch, their tasks alternating: the horizontal walks leave more than half of the VALU
// issue slots free while they wait for their staging, the vertical walks are bound by exactly those slots, so waves of
// the two kinds on one SIMD fill each other's gaps (separate launches on one stream run back to back).
// Placement: workgroup b runs on XCD b % 8.  The launch order is renumbered so that every XCD owns one contiguous range of
// it, and inside that range horizontal and vertical tasks alternate with the SAME task number -- the same frame -- so both
// passes of a frame's plane run on one XCD close together and its L2 fetches the plane from HBM once.  (Alternating by the
// parity of b itself put every horizontal task on the even XCDs and every vertical task on the odd ones: each plane
// crossed the fabric once per pass.)  xcd = 0: the parity order, for A/B.
// (window 65: the rings leave room for 7 waves per CU, so the registers of two waves per SIMD are free to use)
template <int K, bool RANGE = false>
__global__ __launch_bounds__(64, K > 35 ? 2 : 3) void k_bilateral_walk_hv(WalkArgs ah, WalkArgs av, int xcd) {
    const int pairs = min(ah.ntasks, av.ntasks);
    if (xcd) {
        const int b = xcd_contiguous(blockIdx.x, gridDim.x);
        if (b < 2 * pairs) {
            if (b & 1) walk_v_task<K, true>(av, b >> 1);
            else walk_h_task<K, true, RANGE>(ah, b >> 1);
        } else if (ah.ntasks > pairs) walk_h_task<K, true, RANGE>(ah, b - pairs);
        else walk_v_task<K, true>(av, b - pairs);
        return;
    }
    const int b = blockIdx.x;
    if (b < 2 * pairs) {
        if (b & 1) walk_v_task<K>(av, b >> 1);
        else walk_h_task<K, false, RANGE>(ah, b >> 1);
    } else if (ah.ntasks > pairs) walk_h_task<K, false, RANGE>(ah, b - pairs);
    else walk_v_task<K>(av, b - pairs);
}

// partial planes -> merged plane (out may alias p0)
__global__ __launch_bounds__(256) void k_or4_bits(const unsigned long long* __restrict__ p0, const unsigned long long* __restrict__ p1,
                                                 const unsigned long long* __restrict__ p2, const unsigned long long* __restrict__ p3,
                                                 const unsigned long long* __restrict__ n0, const unsigned long long* __restrict__ n1,
                                                 unsigned long long* out, size_t n) {
    const size_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    unsigned long long v = p0[i] | p1[i] | p2[i] | p3[i];
    if (n0) v &= n0[i] | n1[i];      // the greenery mask (lane_tracker.py:229)
    out[i] = v;
}

template <bool VERT>
WalkArgs walk_args(const uint8_t* src, int C, unsigned long long* out, int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n) {
    WalkArgs a;
    a.src = src;
    a.out = out;
    a.h = h; a.w = w; a.wpr = (w + 63) / 64;
    a.pitch = pitch;
    a.C = C;
    a.rbias = 0;
    a.groups = ((VERT ? w : h) + 127) / 128;
    const int len = VERT ? h : w;
    a.segs = len > 640 ? 2 : 1;
    // horizontal segments start at multiples of 128 (the staging blocks are whole 128-byte lines)
    a.seg_len = a.segs == 1 ? ((len + 127) & ~127) : (VERT ? (len + 1) / 2 : ((len / 2) & ~127));
    a.ntasks = a.groups * a.segs * n;
    a.plane_stride = plane_stride;
    a.bits_stride = bits_stride;
    return a;
}

template <int K, bool VERT>
void launch_walk(hipStream_t s, const uint8_t* src, int C, unsigned long long* out, int h, int w, int pitch, size_t plane_stride,
                 size_t bits_stride, int n) {
    const WalkArgs a = walk_args<VERT>(src, C, out, h, w, pitch, plane_stride, bits_stride, n);
    if (VERT) hipLaunchKernelGGL((k_bilateral_walk_v<K>), dim3(a.ntasks), dim3(64), WalkCfg<K>::V_LDS, s, a);
    else hipLaunchKernelGGL((k_bilateral_walk_h<K>), dim3(a.ntasks), dim3(64), WalkCfgH<K>::LDS, s, a);
}

template <int K>
void launch_walk_both(hipStream_t s, const uint8_t* src, int C, unsigned long long* out_h, unsigned long long* out_v, int h, int w,
                      int pitch, size_t plane_stride, size_t bits_stride, int n, int passes) {
    // `passes` (bit 0 horizontal, bit 1 vertical) is a measurement / debugging switch, LT_WALK_PASSES; default both
    const size_t bytes = ((size_t)(n - 1) * bits_stride + (size_t)h * ((w + 63) / 64)) * 8;
    static const bool split = [] { const char* e = LT_EXP_ENV("LT_WALK_SPLIT"); return e && e[0] == '1'; }();   // A/B: one launch per pass
    if ((passes & 3) == 3 && !split) {
        const WalkArgs ah = walk_args<false>(src, C, out_h, h, w, pitch, plane_stride, bits_stride, n);
        const WalkArgs av = walk_args<true>(src, C, out_v, h, w, pitch, plane_stride, bits_stride, n);
        const int lds = std::max(WalkCfgH<K>::LDS, WalkCfg<K>::V_LDS);
        static const int xcd = [] { const char* e = LT_EXP_ENV("LT_WALK_XCD"); return e && e[0] == '0' ? 0 : 1; }();   // A/B
        hipLaunchKernelGGL((k_bilateral_walk_hv<K>), dim3(ah.ntasks + av.ntasks), dim3(64), lds, s, ah, av, xcd);
        return;
    }
    if (passes & 1) launch_walk<K, false>(s, src, C, out_h, h, w, pitch, plane_stride, bits_stride, n);
    else (void)hipMemsetAsync(out_h, 0, bytes, s);
    if (passes & 2) launch_walk<K, true>(s, src, C, out_v, h, w, pitch, plane_stride, bits_stride, n);
    else (void)hipMemsetAsync(out_v, 0, bytes, s);
}

// The greenery-mask walks: window 65 over the raw Lab-b plane, the inRange term folded into the horizontal pass.
// out_h = !inRange | (left & right), out_v = up & down; their OR is the noise mask.
constexpr int K_NOISE = 65;
void launch_walk_noise(hipStream_t s, const uint8_t* src, int C, int noise_thresh, unsigned long long* out_h, unsigned long long* out_v,
                       int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n) {
    WalkArgs ah = walk_args<false>(src, C, out_h, h, w, pitch, plane_stride, bits_stride, n);
    const WalkArgs av = walk_args<true>(src, C, out_v, h, w, pitch, plane_stride, bits_stride, n);
    ah.rbias = (uint32_t)(0x8000 - std::min(std::max(noise_thresh, 0), 256)) * 0x10001u;
    const int lds = std::max(WalkCfgH<K_NOISE>::LDS, WalkCfg<K_NOISE>::V_LDS);
    static const int xcd = [] { const char* e = LT_EXP_ENV("LT_WALK_XCD"); return e && e[0] == '0' ? 0 : 1; }();   // A/B
    hipLaunchKernelGGL((k_bilateral_walk_hv<K_NOISE, true>), dim3(ah.ntasks + av.ntasks), dim3(64), lds, s, ah, av, xcd);
}

bool walk_supports(int k) { return k == 15 || k == 20 || k == 35; }

void dispatch_walk(int k, hipStream_t s, const uint8_t* src, int C, unsigned long long* out_h, unsigned long long* out_v, int h, int w,
                   int pitch, size_t plane_stride, size_t bits_stride, int n, int passes) {
    switch (k) {
        case 15: launch_walk_both<15>(s, src, C, out_h, out_v, h, w, pitch, plane_stride, bits_stride, n, passes); break;
        case 20: launch_walk_both<20>(s, src, C, out_h, out_v, h, w, pitch, plane_stride, bits_stride, n, passes); break;
        default: launch_walk_both<35>(s, src, C, out_h, out_v, h, w, pitch, plane_stride, bits_stride, n, passes); break;
    }
}

}  // namespace

bool bilateral_walk_supported(int k_r, int C_r, int k_b, int C_b, int h, int w, int pitch) {
    static const bool off = [] { const char* e = LT_EXP_ENV("LT_BILATERAL_TILES"); return e && e[0] == '1'; }();
    if (off || !walk_supports(k_r) || !walk_supports(k_b) || C_r < 0 || C_b < 0) return false;
    if ((w & 3) || (pitch & 63) || pitch < w || w < 8 || h < 1) return false;
    return (long long)k_r * (255 + C_r) < 32768 && (long long)k_b * (255 + C_b) < 32768;
}

// Both bilateral thresholds through the walking kernels: four partial bit planes (merged, s1, s2, s3), OR-ed into `merged`
// here when `merge` is set -- otherwise the caller merges them (launch_merge_open5 does it on the way into the 5x5 open).
// Returns 0 when it ran, -1 when the parameters are outside its limits.
int launch_bilateral_walk(hipStream_t s, const uint8_t* thr, int k_r, int C_r, const uint8_t* thb, int k_b, int C_b,
                          unsigned long long* merged, unsigned long long* s1, unsigned long long* s2, unsigned long long* s3,
                          int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n, bool merge) {
    if (n <= 0 || !bilateral_walk_supported(k_r, C_r, k_b, C_b, h, w, pitch) || (plane_stride & 63)) return -1;
    static const int passes = [] { const char* e = LT_EXP_ENV("LT_WALK_PASSES"); return e ? std::atoi(e) : 15; }();
    dispatch_walk(k_r, s, thr, C_r, merged, s1, h, w, pitch, plane_stride, bits_stride, n, passes & 3);
    dispatch_walk(k_b, s, thb, C_b, s2, s3, h, w, pitch, plane_stride, bits_stride, n, (passes >> 2) & 3);
    if (merge) launch_or4_bits(s, merged, s1, s2, s3, h, w, bits_stride, n);
    return 0;
}

bool noise_walk_supported(int k_n, int C_n, int h, int w, int pitch) {
    static const bool off = [] { const char* e = LT_EXP_ENV("LT_BILATERAL_TILES"); return e && e[0] == '1'; }();
    if (off || k_n != K_NOISE || C_n < 0 || (long long)k_n * (255 + C_n) >= 32768) return false;
    return !((w & 3) || (pitch & 63) || pitch < w || w < 8 || h < 1);
}

// noise_h | noise_v = the greenery mask  !inRange(b, noise_thresh, 255) | bilateral(b, 65, C_n)  of lane_tracker.py:223-225;
// `braw` is the RAW Lab-b plane with the padded pitch.  0 = ran, -1 = outside its limits.
int launch_noise_walk(hipStream_t s, const uint8_t* braw, int k_n, int C_n, int noise_thresh, unsigned long long* noise_h,
                      unsigned long long* noise_v, int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n) {
    if (n <= 0 || !noise_walk_supported(k_n, C_n, h, w, pitch) || (plane_stride & 63)) return -1;
    launch_walk_noise(s, braw, C_n, noise_thresh, noise_h, noise_v, h, w, pitch, plane_stride, bits_stride, n);
    return 0;
}

void launch_or4_bits(hipStream_t s, unsigned long long* merged, const unsigned long long* s1, const unsigned long long* s2,
                     const unsigned long long* s3, int h, int w, size_t bits_stride, int n, const unsigned long long* n0,
                     const unsigned long long* n1) {
    if (n <= 0) return;
    const size_t words = (size_t)(n - 1) * bits_stride + (size_t)h * ((w + 63) / 64);
    if (!n0 || !n1) n0 = n1 = nullptr;
    hipLaunchKernelGGL(k_or4_bits, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, merged, s1, s2, s3, n0, n1, merged, words);
}

// Code objects load on the first launch of one of their kernels (a few ms each, once per process and device): lt_create launches
// this no-op so that no stream's first window pays for it (lt_api.cpp: preload_kernels).
namespace { __global__ void k_preload_k_threshold_walk() {} }
void preload_k_threshold_walk(hipStream_t s) { hipLaunchKernelGGL(k_preload_k_threshold_walk, dim3(1), dim3(1), 0, s); }

}  // namespace lt

