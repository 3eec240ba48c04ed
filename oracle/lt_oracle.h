/*
 * lt_oracle.h -- CPU restatement of the lane_tracker hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle and the CPU baseline ("port").  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * What it restates (citations are into the upstream reference, /root/reference):
 *   lane_tracker.py:832      cv2.undistort(img, K, D, None, K)
 *   lane_tracker.py:834      cv2.warpPerspective(img, M, warped_size, INTER_LINEAR, BORDER_CONSTANT)
 *   lane_tracker.py:14-83    bilateral_adaptive_threshold()
 *   lane_tracker.py:183-240  LaneTracker.filter_lane_points()
 *   lane_tracker.py:242-447  LaneTracker.sliding_window_search()
 *   lane_tracker.py:449-500  LaneTracker.band_search()
 *   lane_tracker.py:502-509  LaneTracker.fit_poly()  (np.polyfit degree 2)
 *
 * PARITY STATUS
 *   - sliding_window_search / band_search / fit_poly: PINNED.  The reference's own NumPy code was
 *     imported in the build container (tests/gen_golden.py) and its outputs are committed under
 *     tests/golden/; tests/test_oracle_golden.py checks this file against them.
 *   - every cv2-backed stage (undistort, warpPerspective, RGB2LAB, morphologyEx, filter2D,
 *     adaptiveThreshold): PARITY UNPINNED.  OpenCV (unversioned dependency of the reference, a
 *     2017-era 3.x) is absent from /root/reference and from the build image and the reference holds
 *     no tests or golden vectors.  These functions restate OpenCV's published algorithms as
 *     described in SURVEY.md Appendix A; the reference's call sites fix the arguments.
 */
#ifndef LT_ORACLE_H
#define LT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lto_calib {
    int32_t img_w, img_h;     /* camera frame size  (reference: 1280 x 720) */
    int32_t warp_w, warp_h;   /* bird's-eye size    (reference: 1080 x 1100) */
    double  K[9];             /* camera matrix, row major */
    double  D[5];             /* k1 k2 p1 p2 k3 */
    double  M[9];             /* camera -> bird's-eye homography (NOT the inverse) */
} lto_calib;

typedef struct lto_filter_params {
    int32_t filter_type;      /* 0 = 'bilateral', 1 = 'neighborhood' */
    int32_t ksize_r, C_r, ksize_b, C_b;
    int32_t mask_noise, noise_thresh, ksize_noise, C_noise;
} lto_filter_params;

typedef struct lto_search_params {
    int32_t window_width, window_height, search_range, no_success_limit;
    int32_t ignore_sides, ignore_bottom, bandwidth, _pad;
    double  mu, start_slice, partial;
} lto_search_params;

/* ---- geometric front end ------------------------------------------------------------------ */
/* Fixed-point remap tables for rows [r0, r1) of cv2.undistort's output (3-row stripe semantics). */
void lto_undistort_map(const lto_calib* c, int r0, int r1, int16_t* xy, uint16_t* alpha);
/* Fixed-point remap tables of cv2.warpPerspective for the whole warp_h x warp_w output. */
void lto_warp_map(const lto_calib* c, int16_t* xy, uint16_t* alpha);
/* cv::remap, CV_16SC2 + CV_16UC1 maps, INTER_LINEAR, BORDER_CONSTANT(0), 8UC3. */
void lto_remap_bilinear_c3(const uint8_t* src, int sh, int sw, const int16_t* xy,
                           const uint16_t* alpha, int dh, int dw, uint8_t* dst);
void lto_undistort(const lto_calib* c, const uint8_t* frame, uint8_t* out);
void lto_warp(const lto_calib* c, const uint8_t* und, uint8_t* bev);
/* undistort restricted to the rows the warp samples, then warp: same BEV as the two calls above */
void lto_front_end(const lto_calib* c, const uint8_t* frame, uint8_t* bev);
void lto_warp_source_rows(const lto_calib* c, int* r0, int* r1);

/* ---- colour ------------------------------------------------------------------------------- */
void lto_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int32_t coeffs[9]);
void lto_lab_b(const uint8_t* rgb, int npix, uint8_t* b);
void lto_channel_r(const uint8_t* rgb, int npix, uint8_t* r);

/* ---- morphology --------------------------------------------------------------------------- */
/* half-widths dx[i], i = 0..k-1, of cv2.getStructuringElement(MORPH_ELLIPSE, (k,k)); returns taps */
int  lto_ellipse_halfwidths(int k, int* dx);
void lto_ellipse_kernel(int k, uint8_t* elem);               /* k*k 0/1 footprint */
/* is_dilate = 0: erode (min), 1: dilate (max); out-of-image taps are ignored */
void lto_morph_ellipse_brute(const uint8_t* src, int h, int w, int k, int is_dilate, uint8_t* dst);
void lto_morph_ellipse(const uint8_t* src, int h, int w, int k, int is_dilate, uint8_t* dst);
void lto_tophat(const uint8_t* src, int h, int w, int k, uint8_t* dst);
void lto_open(const uint8_t* src, int h, int w, int k, uint8_t* dst);

/* ---- thresholds --------------------------------------------------------------------------- */
/* mode: 0 = 'floor', 1 = 'ceil'.  Returns 0, or -1 for a bad mode (reference raises ValueError). */
int  lto_bilateral_adaptive_threshold(const uint8_t* img, int h, int w, int ksize, int C, int mode,
                                      int true_value, int false_value, uint8_t* out);
/* cv2.adaptiveThreshold(src,255,MEAN_C,THRESH_BINARY,bs,-C): 255 iff src - boxmean > C */
void lto_adaptive_mean_threshold(const uint8_t* src, int h, int w, int bs, int C, uint8_t* out);
/* The same two with running sums (O(1) per pixel).  Not the parity checker: they exist so that the CPU baseline of
 * bench.py is a fair CPU path; tests/test_oracle_units.py checks them equal to the functions above. */
int  lto_bilateral_adaptive_threshold_fast(const uint8_t* img, int h, int w, int ksize, int C, int mode,
                                           int true_value, int false_value, uint8_t* out);
void lto_adaptive_mean_threshold_fast(const uint8_t* src, int h, int w, int bs, int C, uint8_t* out);

/* ---- filter_lane_points and the whole mask stage ------------------------------------------ */
/* returns 0, or -1 for a bad filter_type.  planes (optional, may be NULL): 4 planes h*w each:
 * R, Lab-b, tophat(R), tophat(b)  (the last two only written for filter_type 0) */
int  lto_filter_lane_points(const uint8_t* bev_rgb, int h, int w, const lto_filter_params* p,
                            uint8_t* mask, uint8_t* planes);
int  lto_mask_from_frame(const lto_calib* c, const uint8_t* frame, const lto_filter_params* p,
                         uint8_t* mask);
int  lto_filter_lane_points_fast(const uint8_t* bev_rgb, int h, int w, const lto_filter_params* p,
                                 uint8_t* mask, uint8_t* planes);   /* running-sum thresholds; equal results */

/* ---- search + fit ------------------------------------------------------------------------- */
/* Pixel lists are (y, x) int32 pairs, reference order.  Capacity per side must be >= h*w for
 * band search and >= nlevels*window_height*window_width for the sliding-window search.
 * Centroid arrays need capacity >= nlevels+1.  Returns detected_pixels (0/1). */
int  lto_sliding_window_search(const uint8_t* mask, int h, int w, const lto_search_params* p,
                               int32_t* ly, int32_t* lx, int32_t* nl,
                               int32_t* ry, int32_t* rx, int32_t* nr,
                               int32_t* lcent, int32_t* nlc, int32_t* rcent, int32_t* nrc);
int  lto_band_search(const uint8_t* mask, int h, int w, const lto_search_params* p,
                     const double lcoef[3], const double rcoef[3],
                     int32_t* ly, int32_t* lx, int32_t* nl,
                     int32_t* ry, int32_t* rx, int32_t* nr);
/* np.polyfit(y, x, 2) restated: column-scaled Vandermonde, least squares by Householder QR.
 * Returns the numerical rank found (3 = full rank). */
int  lto_polyfit2(const int32_t* y, const int32_t* x, int n, double coef[3]);

/* One independent frame, fresh tracker (BASELINE configs 2-3): mask -> sws -> fit.
 * record: 6 doubles (left a,b,c, right a,b,c) ; counts: nl, nr, detected.  scratch-free. */
int  lto_frame_sws_fit(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp,
                       const lto_search_params* sp, uint8_t* mask_out, double coef[6],
                       int32_t counts[3]);
int  lto_frame_sws_fit_fast(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp,
                            const lto_search_params* sp, uint8_t* mask_out, double coef[6],
                            int32_t counts[3]);   /* running-sum thresholds (cpu_baseline); equal results */

/* ---- presentation (cv2 calls, parity unpinned) ------------------------------ */
/* cv2.fillPoly(img, [pts], color) for ONE polygon on an interleaved u8 image with `ch` channels,
 * lineType 8, shift 0: every edge drawn with the 8-connected line iterator, interior filled by
 * the even-odd scanline rule.  pts = npts (x, y) int32 pairs; pixels outside the image are clipped. */
void lto_fill_poly(uint8_t* img, int h, int w, int ch, const int32_t* pts, int npts, const uint8_t* color);
/* cv2.addWeighted(a, alpha, b, beta, gamma) on u8: f32 arithmetic, round-half-even, saturate */
void lto_add_weighted_u8(const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma,
                         size_t n, uint8_t* out);
/* draw_lane (lane_tracker.py:629-662) without the text: polygon between the averaged lane points,
 * filled (0,255,0) on a blank bird's-eye image, warpPerspective(.., Minv, (img_w,img_h)), then
 * addWeighted(img, 1, lane, 0.3, 0).  c->M is unused; Minv is the pickled inverse matrix. */
void lto_draw_lane(const lto_calib* c, const double Minv[9], const uint8_t* img,
                   const int32_t* left_y, const int32_t* left_x, int n_left,
                   const int32_t* right_y, const int32_t* right_x, int n_right, uint8_t* out);
/* cv2.resize(src, (dw, dh)) INTER_LINEAR on u8 with `ch` interleaved channels (fixed-point 11-bit
 * coefficients, half-pixel centres) -- used by create_split_view (utils.py:89) */
void lto_resize_linear_u8(const uint8_t* src, int sh, int sw, int ch, int dh, int dw, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif
