// Internal declarations shared by the C-ABI layer (lt_api.cpp), the host table builders
// (lt_tables.cpp) and the kernel launchers (*.hip).  Not installed; the public ABI is
// include/lane_tracker_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include <cstdlib>

#include "../../include/lane_tracker_amd.h"

// Measurement switches -- alternative kernels and launch shapes for A/B runs, each held bit-exact by the parity suite -- exist in
// the EXPERIMENTS build only (`make EXPERIMENTS=1` -> liblane_tracker_amd_exp.so, -DLT_EXPERIMENTS; tools/* and
// tests/test_gpu_parity.py::test_alternative_kernel_paths_keep_parity load that one).  In the release library LT_EXP_ENV(...) is a
// null pointer constant: the path behind a switch is dead code the compiler drops, and the switch's name does not reach the
// binary (tests/test_native_abi.py counts the LT_* strings of the release .so against INTEGRATION.md section E).
#ifdef LT_EXPERIMENTS
#define LT_EXP_ENV(name) std::getenv(name)
#else
#define LT_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace lt {

// ---- host tables (lt_tables.cpp) ----------------------------------------------------------------
struct RemapTable {              // cv::remap fixed-point maps: integer tap + 5+5-bit fraction
    std::vector<int16_t> xy;     // (sx, sy) interleaved
    std::vector<uint16_t> frac;  // fy*32 + fx
    int rows = 0, cols = 0;
};
void build_warp_table(const lt_calib& c, RemapTable& t);
void warp_source_rows(const lt_calib& c, const RemapTable& warp, int& r0, int& r1);
void build_undistort_table(const lt_calib& c, int r0, int r1, RemapTable& t);
void build_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int32_t coeffs[9]);
int  ellipse_halfwidths(int k, int* dx);  // returns tap count

struct EllipseSE {               // one horizontal run per row
    int k;
    int8_t dx[64];
};

// ---- kernel launchers ----------------------------------------------------------------------------
struct FrontEndGeom {
    int img_h, img_w, warp_h, warp_w, r0, nrows;
};
// The undistorted rows of slots 2p and 2p+1 are interleaved per pixel (k_frontend.hip): dword index of pixel 0 of a slot;
// its pixel i is 2 i dwords further.  The buffer holds ceil(slots / 2) pairs of 2 * und_px dwords.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t und_slot_base(size_t und_px, int slot) { return (size_t)(slot >> 1) * 2 * und_px + (size_t)(slot & 1); }
// `und` is the base of the whole buffer, `first_slot` the absolute slot of frame 0 of the call; frames / planes point at
// that slot's data
void launch_undistort_rows(hipStream_t s, const uint8_t* frames, size_t frame_stride, const int16_t* uxy,
                           const uint16_t* ufrac, FrontEndGeom g, uint32_t* und, size_t und_px, int first_slot, int n);
void launch_warp_split(hipStream_t s, const uint32_t* und, size_t und_px, int first_slot, const int16_t* wxy,
                       const uint16_t* wfrac, FrontEndGeom g, const uint16_t* gamma_tab, const uint16_t* cbrt_tab,
                       const int32_t* coeffs, uint8_t* planeR, uint8_t* planeB, size_t plane_stride, int n);
void launch_split_bev(hipStream_t s, const uint8_t* bev, size_t bev_stride, int npix, const uint16_t* gamma_tab,
                      const uint16_t* cbrt_tab, const int32_t* coeffs, uint8_t* planeR, uint8_t* planeB,
                      size_t plane_stride, int n);
void launch_undistorted_to_rgb(hipStream_t s, const uint32_t* und, size_t und_px, int first_slot, int nrows, int w, uint8_t* out,
                               int n);

// one no-op launch per kernel translation unit: loads its code object (lt_create)
void preload_k_frontend(hipStream_t s);
void preload_k_filter(hipStream_t s);
void preload_k_tophat(hipStream_t s);
void preload_k_threshold(hipStream_t s);
void preload_k_threshold_walk(hipStream_t s);
void preload_k_adaptive_walk(hipStream_t s);
void preload_k_search(hipStream_t s);
void preload_k_overlay(hipStream_t s);

// presentation stage (k_overlay.hip)
// strip mode: only the camera rows [row0, row1) of every slot's annotated frame, packed, strip_stride bytes per slot; false: the
// geometry does not allow the four-pixel kernel (nothing launched)
bool launch_overlay_lane_strip(hipStream_t s, const uint8_t* frames, size_t frame_stride, uint8_t* strips, size_t strip_stride,
                               const int16_t* oxy, const uint16_t* ofrac, const int16_t* spans, size_t span_stride_rows, int img_w,
                               int row0, int row1, int bh, int bw, float alpha, int n);
void launch_overlay_lane(hipStream_t s, const uint8_t* frames, uint8_t* out, size_t frame_stride, const int16_t* oxy,
                         const uint16_t* ofrac, const int16_t* spans, size_t span_stride_rows, int img_h, int img_w,
                         int bh, int bw, float alpha, int n, const int* rows4 = nullptr);
// one frame, the row intervals (host memory, bh pairs) passed as a kernel argument; rows4 = nullptr: the whole frame, else two
// runs of camera rows {a0, a1, b0, b1} (the others are not written); false: not launched (bh above LT_SPAN_ARG_ROWS, a row
// length that is no multiple of 4, or the runtime refused the argument block)
constexpr int LT_SPAN_ARG_ROWS = 1104;
void launch_store_word(hipStream_t s, unsigned* dev_word, unsigned value);
bool launch_lane_spans_from_regions(hipStream_t s, const double* ploty, const double* ploty2, int n_rows, int bh, int bw, int16_t* spans, int n);
bool launch_lane_spans_from_fit(hipStream_t s, const lt_lane_record* rec, const double* prev_sum, int count, const double* ploty,
                                const double* ploty2, int n_rows, int bh, int bw, int16_t* spans);
bool launch_overlay_lane_one(hipStream_t s, const uint8_t* frame, uint8_t* out, const int16_t* oxy, const uint16_t* ofrac,
                             const int16_t* spans_host, int img_h, int img_w, int bh, int bw, float alpha, const int* rows4);
void launch_overlay_text(hipStream_t s, uint8_t* out, size_t frame_stride, int img_h, int img_w, const uint8_t* atlas,
                         const uint8_t* advance, int first_char, int n_glyphs, int gw, int gh, const uint8_t* lines,
                         const int16_t* xpos, int nl, int len, int slot_chars, int y0, int step, int n);   // slot_chars: characters between two slots' lines
// host -> device copy of a few hundred KB out of page-locked memory as a kernel launch (never blocks the caller)
void launch_copy_from_pinned(hipStream_t s, void* dst, const void* src_pinned, size_t bytes);
bool launch_copy_to_pinned(hipStream_t s, void* dst_pinned, const void* src, size_t bytes);   // false: not page-locked / aligned
bool launch_copy_rows_to_pinned(hipStream_t s, void* dst_pinned, const void* src, size_t pitch, size_t off, size_t bytes, int n);
bool launch_copy_words_to_pinned(hipStream_t s, void* dst_pinned, const void* src, size_t bytes);          // small, 4-byte granular
bool launch_mirror_record(hipStream_t s, void* dst_pinned, const void* src_record, unsigned ticket);       // a 64-byte record + the ticket word behind it
void launch_warp_rgb(hipStream_t s, const uint32_t* und, size_t und_px, int first_slot, const int16_t* wxy, const uint16_t* wfrac,
                     FrontEndGeom g, uint8_t* bev, size_t bev_stride, int n);

// dst = erode/dilate(src) with the ellipse; if minuend != nullptr: dst = sat(minuend - result)
void launch_morph_ellipse(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w,
                          const EllipseSE& se, bool dilate, size_t plane_stride, int n);
// decomposed 29x29 / 55x55 ellipses (k_tophat.hip); same contract as launch_morph_ellipse
// dpitch > 0: the destination has its own row pitch and per-frame stride (the padded top-hat planes the threshold walks read)
// copy_dst (55x55 top-hat with dpitch > 0 only): the minuend is stored there as well, in the destination's layout; false
// when that form is not available for the geometry (nothing was launched)
bool launch_morph_runs(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* minuend, int h, int w, int k,
                       bool dilate, size_t plane_stride, int n, int dpitch = 0, size_t dst_stride = 0, uint8_t* copy_dst = nullptr);
// one or two frames: the same step of the 55x55 chain of one plane and of the 29x29 chain of another in one launch; false: not launched
bool launch_morph_one_pair(hipStream_t s, const uint8_t* src55, uint8_t* dst55, const uint8_t* min55, const uint8_t* src29, uint8_t* dst29,
                           const uint8_t* min29, int h, int w, bool dilate, size_t plane_stride, int n, int dpitch, size_t dst_stride);
bool tophat_tables_match(const EllipseSE& se29, const EllipseSE& se55);
void launch_bilateral(hipStream_t s, const uint8_t* src, uint8_t* dst, int h, int w, int ksize, int C, int mode,
                      int tv, int fv, size_t plane_stride, int n);
void launch_adaptive_mean(hipStream_t s, const uint8_t* src, uint8_t* dst, int h, int w, int bs, int C,
                          size_t plane_stride, int n);
// the 'neighborhood' filter (lane_tracker.py:217-218) of both planes as running box sums (k_adaptive_walk.hip): out_r / out_b
// are the two thresholded bit planes.  false = outside its limits (nothing launched).
bool adaptive_walk_supported(int bs_r, int bs_b, int h, int w, size_t plane_stride);
bool launch_adaptive_walk(hipStream_t s, const uint8_t* R, int bs_r, int C_r, unsigned long long* out_r, const uint8_t* B, int bs_b,
                          int C_b, unsigned long long* out_b, int h, int w, size_t plane_stride, size_t bits_stride, int n);
// merged = ((tr | tb) & (use_noise ? (!(labb >= thresh) | noise_bil) : 1)) ? 255 : 0
void launch_merge(hipStream_t s, const uint8_t* tr, const uint8_t* tb, const uint8_t* labb, const uint8_t* noise_bil,
                  int noise_thresh, int use_noise, uint8_t* merged, size_t npix, size_t plane_stride, int n);

// bit-plane path (k_threshold.hip): fused bilateral thresholds + merge, u8->bits merge, 5x5 open on bits
int launch_bilateral_bits(hipStream_t s, const uint8_t* thr, int k_r, int C_r, const uint8_t* thb, int k_b, int C_b,
                          const uint8_t* labb, int k_n, int C_n, int noise_thresh, int use_noise,
                          unsigned long long* bits, int h, int w, size_t plane_stride, size_t bits_stride, int n,
                          unsigned long long* bits_v = nullptr);   // bits_v: the V phases' verdicts as a partial plane of their own (two workgroups per tile)
void launch_pack_merge(hipStream_t s, const uint8_t* tr, const uint8_t* tb, const uint8_t* labb, const uint8_t* nb,
                       int noise_thresh, int use_noise, unsigned long long* bits, int h, int w, size_t plane_stride,
                       size_t bits_stride, int n);
void launch_open5_bits(hipStream_t s, const unsigned long long* merged, unsigned long long* eroded, uint8_t* mask, int h,
                       int w, size_t plane_stride, size_t bits_stride, int n);
// both bilateral thresholds + merge through the long-walk kernels (k_threshold_walk.hip).  The top-hat planes have the row
// pitch `pitch` (a multiple of 64) and `plane_stride` bytes per frame.  0 = ran, -1 = outside its limits.
bool bilateral_walk_supported(int k_r, int C_r, int k_b, int C_b, int h, int w, int pitch);
int launch_bilateral_walk(hipStream_t s, const uint8_t* thr, int k_r, int C_r, const uint8_t* thb, int k_b, int C_b,
                          unsigned long long* merged, unsigned long long* s1, unsigned long long* s2, unsigned long long* s3,
                          int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n, bool merge);
void launch_or4_bits(hipStream_t s, unsigned long long* merged, const unsigned long long* s1, const unsigned long long* s2,
                     const unsigned long long* s3, int h, int w, size_t bits_stride, int n, const unsigned long long* n0 = nullptr,
                     const unsigned long long* n1 = nullptr);
// the greenery mask (lane_tracker.py:223-225) through the walking kernels: noise_h | noise_v = !inRange(b, thresh, 255) |
// bilateral(b, 65, C_n); `braw` = the raw Lab-b plane with the padded pitch.  0 = ran, -1 = outside its limits.
bool noise_walk_supported(int k_n, int C_n, int h, int w, int pitch);
int launch_noise_walk(hipStream_t s, const uint8_t* braw, int k_n, int C_n, int noise_thresh, unsigned long long* noise_h,
                      unsigned long long* noise_v, int h, int w, int pitch, size_t plane_stride, size_t bits_stride, int n);
// erode + dilate with the 5x5 ellipse, bit plane in, bit plane out
// p1 alone: two partial planes.  n0 / n1 (both or neither; with p1..p3 only): the merged plane is (p0 | p1 | p2 | p3) & (n0 | n1)
bool launch_merge_open5(hipStream_t s, unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                        const unsigned long long* p3, unsigned long long* opened, int h, int w, size_t bits_stride, int n,
                        const unsigned long long* n0 = nullptr, const unsigned long long* n1 = nullptr);
// ... for a few frames: one launch of small workgroups (k_or_open5_small)
bool launch_or_open5_small(hipStream_t s, unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                        const unsigned long long* p3, unsigned long long* opened, int h, int w, size_t bits_stride, int n,
                        const unsigned long long* n0 = nullptr, const unsigned long long* n1 = nullptr);
void launch_open5_to_bits(hipStream_t s, const unsigned long long* merged, unsigned long long* eroded,
                          unsigned long long* opened, int h, int w, size_t bits_stride, int n);
void launch_bits_to_u8(hipStream_t s, const unsigned long long* bits, uint8_t* out, int h, int w, size_t plane_stride,
                       size_t bits_stride, int n);


struct SearchGeom {
    int h, w;                    // mask size
    int ww, wh, hw;              // window
    int img_height;              // h - ignore_bottom
    int img_center, y_start, nlevels, limit;
    int nbands;                  // column-sum bands per frame: max(nlevels, 1); band 0 = start slice
    int ignore_sides, search_range, def_left, def_right;
    int band_top, band_bottom;   // band search rows [top, bottom)
    double mu, bandwidth;
    int maxpix, maxlev;
};
// Per-frame block k_sws_fit2 leaves in the pixel buffer (u32 units): [0] nlev, [1] window height, [2] image
// height minus ignore_bottom, [3] 0; [4 ...) the (a, b) column range per (side, level); then, 8-byte aligned,
// one 64-bit column mask per (side, level, row): bit j = pixel (row, a + j) is set.
__host__ __device__ inline int sws2_mask_offset(int nlev) { return (4 + 4 * nlev + 1) & ~1; }
__host__ __device__ inline long long sws2_block_words(int nlev, int wh) { return sws2_mask_offset(nlev) + 4LL * nlev * wh; }
// k_band_fit2's per-frame block: [0] rows per side, [1] first row, [2..3] 0; int32 a per (side, row); then,
// 8-byte aligned, one u64 column mask per (side, row).  LDS: masks, a, width per row, 4 counters, 16 moments.
__host__ __device__ inline int band2_mask_offset(int nrows) { return (4 + 2 * nrows + 1) & ~1; }
__host__ __device__ inline long long band2_block_words(int nrows) { return band2_mask_offset(nrows) + 4LL * nrows; }
__host__ __device__ inline size_t band2_mom_offset(int nrows) { return ((size_t)2 * nrows * 16 + 16 + 15) & ~(size_t)15; }
// Mask input of the searches: u8 planes, or -- when `bits` is not null -- the opened bit plane the mask chain
// leaves (wpr words per row, bits_stride words per frame).  *_takes_bits tells whether the kernel version that
// would run for this geometry reads bit planes (the first-version kernels need u8 masks).
struct MaskBits {
    const unsigned long long* bits;
    size_t bits_stride;
    int wpr;
};
bool sws_fit_takes_bits(const SearchGeom& g, size_t mask_stride);
bool band_fit_takes_bits(const SearchGeom& g, size_t mask_stride);
void launch_sws_fit(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, uint32_t* band_sums,
                    uint32_t* pix, int32_t* cent, lt_lane_record* rec, int n);
// previous coefficients of a single frame travel as a kernel argument (no upload, no synchronisation);
// batches read them from device memory
struct BandPrev {
    double c[6];
    int by_value;
};
void launch_band_fit(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, const double* prev,
                     const BandPrev& bp, uint32_t* pix, lt_lane_record* rec, int n);

// the chained band search of one stream (k_band_chain2): slots [first, first + n) in order, frame k+1 around frame k's fit;
// the first frame around `seed` (by value) or, with seed.by_value == 0, around the fit in *seed_rec (device memory)
bool band_chain_supported(const SearchGeom& g, size_t mask_stride);
bool launch_band_fit_one(hipStream_t s, MaskBits mb, SearchGeom g, const BandPrev& bp, uint32_t* pix, lt_lane_record* rec,
                         size_t mask_stride, const int* zero, lt_lane_record* mirror, unsigned mirror_ticket);
void launch_band_chain(hipStream_t s, const uint8_t* masks, size_t mask_stride, MaskBits mb, SearchGeom g, const lt_lane_record* seed_rec,
                       const BandPrev& seed, uint32_t* pix, lt_lane_record* rec, int n, const int* cancel_epoch, int my_epoch);

// fit of one explicit pixel list (packed (y<<16)|x); out: 3 doubles + 1 flag double (1.0 = rank deficient)
void launch_fit_list(hipStream_t s, const uint32_t* pix, int n, int h, int w, double* out4);

// ---- internal view of a context for lt_gather.cpp (defined in lt_api.cpp) ----------------------------
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // fills lt_last_error(), returns code
int ctx_device(lt_ctx* c);
int ctx_streams(lt_ctx* c, hipStream_t* out, int cap);        // the streams the slots currently run on; returns the count
int ctx_sync(lt_ctx* c);
int ctx_enqueue_records(lt_ctx* c, int first, int n, lt_lane_record* dst_device);
// every stream of the library comes from a per-process pool and goes back to it idle: none is ever destroyed (lt_api.cpp, StreamPool)
enum StreamKind { SK_COMPUTE = 0, SK_PLAIN, SK_CU_SET, SK_PRIORITY };
hipError_t stream_get(hipStream_t* st, StreamKind kind, int param);
void stream_put(hipStream_t st);

}  // namespace lt
