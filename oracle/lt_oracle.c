/*
 * lt_oracle.c -- CPU restatement of the lane_tracker hot path.  TEST INFRASTRUCTURE ONLY.
 * See lt_oracle.h for scope, citations and parity status (cv2-backed stages: PARITY UNPINNED;
 * NumPy-only stages: pinned by tests/golden/ generated from the reference itself).
 *
 * Build: see oracle/Makefile.  Must be compiled with -ffp-contract=off: the f64 coordinate
 * generators follow OpenCV's operation order and an FMA contraction changes the last bit.
 */
#include "lt_oracle.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE 32
#define REMAP_COEF_BITS 15

/* cvRound(double): round half to even (lrint under the default rounding mode), saturating. */
static int round_half_even_sat(double v) {
    if (!(v > -2147483648.0)) return INT_MIN; /* also NaN */
    if (!(v < 2147483647.0)) return INT_MAX;
    return (int)nearbyint(v);
}
static int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static uint16_t sat_u16_round(double v) {
    int r = round_half_even_sat(v);
    return (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
}

/* closed-form 3x3 inverse used by cv::invert / Matx33d::inv for 3x3 (SURVEY App. A.1/A.2 [M]) */
static int inv3x3(const double* a, double* b) {
    double d = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) +
               a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (d == 0.0) return 0;
    d = 1.0 / d;
    double t[9];
    t[0] = (a[4] * a[8] - a[5] * a[7]) * d;
    t[1] = (a[2] * a[7] - a[1] * a[8]) * d;
    t[2] = (a[1] * a[5] - a[2] * a[4]) * d;
    t[3] = (a[5] * a[6] - a[3] * a[8]) * d;
    t[4] = (a[0] * a[8] - a[2] * a[6]) * d;
    t[5] = (a[2] * a[3] - a[0] * a[5]) * d;
    t[6] = (a[3] * a[7] - a[4] * a[6]) * d;
    t[7] = (a[1] * a[6] - a[0] * a[7]) * d;
    t[8] = (a[0] * a[4] - a[1] * a[3]) * d;
    memcpy(b, t, sizeof t);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* cv2.undistort (lane_tracker.py:832), SURVEY App. A.1.  R = I, new camera matrix = K.          */
/* undistort() walks the image in stripes of max(1, 4096/cols) rows; for the stripe that starts  */
/* at row y0 it shifts cy of the NEW camera matrix by -y0 and builds maps for local rows i.      */
void lto_undistort_map(const lto_calib* c, int r0, int r1, int16_t* xy, uint16_t* alpha) {
    const int w = c->img_w, h = c->img_h;
    int stripe = 4096 / (w > 1 ? w : 1);
    if (stripe < 1) stripe = 1;
    if (stripe > h) stripe = h;
    const double u0 = c->K[2], v0 = c->K[5], fx = c->K[0], fy = c->K[4];
    const double k1 = c->D[0], k2 = c->D[1], p1 = c->D[2], p2 = c->D[3], k3 = c->D[4];
    for (int row = r0; row < r1; ++row) {
        const int y0 = (row / stripe) * stripe, i = row - y0;
        double Ar[9], ir[9];
        memcpy(Ar, c->K, sizeof Ar);
        Ar[5] = c->K[5] - y0;
        inv3x3(Ar, ir);
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        int16_t* m1 = xy + (size_t)(row - r0) * w * 2;
        uint16_t* m2 = alpha + (size_t)(row - r0) * w;
        for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double ww = 1. / _w, x = _x * ww, y = _y * ww;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / 1.0; /* k4..k6 = 0 */
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            double u = fx * xd + u0;
            double v = fy * yd + v0;
            int iu = round_half_even_sat(u * INTER_TAB_SIZE);
            int iv = round_half_even_sat(v * INTER_TAB_SIZE);
            m1[j * 2] = (int16_t)(iu >> INTER_BITS);
            m1[j * 2 + 1] = (int16_t)(iv >> INTER_BITS);
            m2[j] = (uint16_t)((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1)));
        }
    }
}

/* cv2.warpPerspective (lane_tracker.py:834), SURVEY App. A.2.  OpenCV inverts M itself.         */
void lto_warp_map(const lto_calib* c, int16_t* xy, uint16_t* alpha) {
    double m[9];
    if (!inv3x3(c->M, m)) memset(m, 0, sizeof m);
    const int W = c->warp_w, H = c->warp_h, BW = 64;
    for (int y = 0; y < H; ++y) {
        for (int xb = 0; xb < W; xb += BW) {
            int bw = W - xb < BW ? W - xb : BW;
            double X0 = m[0] * xb + m[1] * y + m[2];
            double Y0 = m[3] * xb + m[4] * y + m[5];
            double W0 = m[6] * xb + m[7] * y + m[8];
            for (int x1 = 0; x1 < bw; ++x1) {
                double Wd = W0 + m[6] * x1;
                Wd = Wd != 0.0 ? INTER_TAB_SIZE / Wd : 0;
                double fX = fmax((double)INT_MIN, fmin((double)INT_MAX, (X0 + m[0] * x1) * Wd));
                double fY = fmax((double)INT_MIN, fmin((double)INT_MAX, (Y0 + m[3] * x1) * Wd));
                int X = round_half_even_sat(fX), Y = round_half_even_sat(fY);
                size_t o = (size_t)y * W + xb + x1;
                xy[o * 2] = sat_s16(X >> INTER_BITS);
                xy[o * 2 + 1] = sat_s16(Y >> INTER_BITS);
                alpha[o] = (uint16_t)((Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1)));
            }
        }
    }
}

/* 15-bit fixed-point bilinear weights are exact integers; border constant 0.   */
void lto_remap_bilinear_c3(const uint8_t* src, int sh, int sw, const int16_t* xy,
                           const uint16_t* alpha, int dh, int dw, uint8_t* dst) {
    for (size_t o = 0; o < (size_t)dh * dw; ++o) {
        int sx = xy[o * 2], sy = xy[o * 2 + 1];
        int fx = alpha[o] & 31, fy = alpha[o] >> 5;
        int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
        int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        for (int ch = 0; ch < 3; ++ch) {
            int v00 = 0, v01 = 0, v10 = 0, v11 = 0;
            if (sy >= 0 && sy < sh) {
                if (sx >= 0 && sx < sw) v00 = src[((size_t)sy * sw + sx) * 3 + ch];
                if (sx + 1 >= 0 && sx + 1 < sw) v01 = src[((size_t)sy * sw + sx + 1) * 3 + ch];
            }
            if (sy + 1 >= 0 && sy + 1 < sh) {
                if (sx >= 0 && sx < sw) v10 = src[((size_t)(sy + 1) * sw + sx) * 3 + ch];
                if (sx + 1 >= 0 && sx + 1 < sw) v11 = src[((size_t)(sy + 1) * sw + sx + 1) * 3 + ch];
            }
            int s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
            dst[o * 3 + ch] = (uint8_t)((s + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS);
        }
    }
}

void lto_undistort(const lto_calib* c, const uint8_t* frame, uint8_t* out) {
    size_t n = (size_t)c->img_w * c->img_h;
    int16_t* xy = malloc(n * 4);
    uint16_t* al = malloc(n * 2);
    lto_undistort_map(c, 0, c->img_h, xy, al);
    lto_remap_bilinear_c3(frame, c->img_h, c->img_w, xy, al, c->img_h, c->img_w, out);
    free(xy);
    free(al);
}

void lto_warp(const lto_calib* c, const uint8_t* und, uint8_t* bev) {
    size_t n = (size_t)c->warp_w * c->warp_h;
    int16_t* xy = malloc(n * 4);
    uint16_t* al = malloc(n * 2);
    lto_warp_map(c, xy, al);
    lto_remap_bilinear_c3(und, c->img_h, c->img_w, xy, al, c->warp_h, c->warp_w, bev);
    free(xy);
    free(al);
}

/* rows [r0, r1) of the undistorted image that the warp touches (taps sy and sy+1, in-image) */
void lto_warp_source_rows(const lto_calib* c, int* r0, int* r1) {
    size_t n = (size_t)c->warp_w * c->warp_h;
    int16_t* xy = malloc(n * 4);
    uint16_t* al = malloc(n * 2);
    lto_warp_map(c, xy, al);
    int lo = INT_MAX, hi = INT_MIN;
    for (size_t o = 0; o < n; ++o) {
        int sx = xy[o * 2], sy = xy[o * 2 + 1];
        if (sx + 1 < 0 || sx >= c->img_w) continue;
        for (int t = 0; t < 2; ++t) {
            int yy = sy + t;
            if (yy < 0 || yy >= c->img_h) continue;
            if (yy < lo) lo = yy;
            if (yy > hi) hi = yy;
        }
    }
    if (lo > hi) { lo = 0; hi = -1; }
    *r0 = lo;
    *r1 = hi + 1;
    free(xy);
    free(al);
}

/* Cached per-calibration tables: the CPU baseline must not regenerate them for every frame. */
static struct {
    lto_calib c;
    int valid, r0, r1;
    int16_t *uxy, *wxy;
    uint16_t *ual, *wal;
} g_fe;

static void front_end_tables(const lto_calib* c) {
    if (g_fe.valid && memcmp(&g_fe.c, c, sizeof *c) == 0) return;
    free(g_fe.uxy); free(g_fe.ual); free(g_fe.wxy); free(g_fe.wal);
    g_fe.c = *c;
    size_t nw = (size_t)c->warp_w * c->warp_h;
    g_fe.wxy = malloc(nw * 4);
    g_fe.wal = malloc(nw * 2);
    lto_warp_map(c, g_fe.wxy, g_fe.wal);
    lto_warp_source_rows(c, &g_fe.r0, &g_fe.r1);
    size_t nu = (size_t)(g_fe.r1 - g_fe.r0) * c->img_w;
    g_fe.uxy = malloc(nu * 4 + 4);
    g_fe.ual = malloc(nu * 2 + 2);
    lto_undistort_map(c, g_fe.r0, g_fe.r1, g_fe.uxy, g_fe.ual);
    g_fe.valid = 1;
}

void lto_front_end(const lto_calib* c, const uint8_t* frame, uint8_t* bev) {
    front_end_tables(c);
    const int r0 = g_fe.r0, r1 = g_fe.r1, w = c->img_w;
    uint8_t* und = calloc((size_t)c->img_h * w * 3, 1);
    lto_remap_bilinear_c3(frame, c->img_h, w, g_fe.uxy, g_fe.ual, r1 - r0, w, und + (size_t)r0 * w * 3);
    lto_remap_bilinear_c3(und, c->img_h, w, g_fe.wxy, g_fe.wal, c->warp_h, c->warp_w, bev);
    free(und);
}

/* ------------------------------------------------------------------------------------------- */
/* cv2.cvtColor(RGB2LAB) 8-bit integer path, b channel only (lane_tracker.py:208), App. A.4 [M].  */
void lto_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int32_t coeffs[9]) {
    for (int i = 0; i < 256; ++i) {
        float x = i * (1.f / 255.f);
        float g = x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow((double)(x + 0.055) * (1. / 1.055), 2.4);
        gamma_tab[i] = sat_u16_round((double)(255.f * (1 << 3) * g));
    }
    for (int i = 0; i < 3072; ++i) {
        float x = i * (1.f / (255.f * (1 << 3)));
        float f = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : (float)cbrt((double)x);
        cbrt_tab[i] = sat_u16_round((double)((1 << 15) * f));
    }
    static const float m[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                               0.072169f, 0.019334f, 0.119193f, 0.950227f};
    static const float wp[3] = {0.950456f, 1.f, 1.088754f};
    float scale[3] = {(1 << 12) / wp[0], (float)(1 << 12), (1 << 12) / wp[2]};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) coeffs[i * 3 + j] = round_half_even_sat((double)(m[i * 3 + j] * scale[i]));
}

#define DESCALE(v, n) (((v) + (1 << ((n)-1))) >> (n))

void lto_lab_b(const uint8_t* rgb, int npix, uint8_t* b) {
    static uint16_t gt[256], ct[3072];
    static int32_t C[9];
    static int init;
    if (!init) { lto_lab_tables(gt, ct, C); init = 1; }
    for (int i = 0; i < npix; ++i) {
        int R = gt[rgb[i * 3]], G = gt[rgb[i * 3 + 1]], B = gt[rgb[i * 3 + 2]];
        int iy = DESCALE(R * C[3] + G * C[4] + B * C[5], 12);
        int iz = DESCALE(R * C[6] + G * C[7] + B * C[8], 12);
        if (iy > 3071) iy = 3071; /* cannot happen for the sRGB matrix; guards the table */
        if (iz > 3071) iz = 3071;
        int fY = ct[iy], fZ = ct[iz];
        int v = DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15);
        b[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

void lto_channel_r(const uint8_t* rgb, int npix, uint8_t* r) {
    for (int i = 0; i < npix; ++i) r[i] = rgb[i * 3];
}

/* ------------------------------------------------------------------------------------------- */
/* cv2.getStructuringElement(MORPH_ELLIPSE,(k,k)) (lane_tracker.py:203-205), App. A.5.          */
int lto_ellipse_halfwidths(int k, int* dx) {
    int r = k / 2, c = k / 2, taps = 0;
    double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < k; ++i) {
        int dy = i - r;
        int d = round_half_even_sat(c * sqrt((r * r - dy * dy) * inv_r2));
        int j1 = c - d > 0 ? c - d : 0, j2 = c + d + 1 < k ? c + d + 1 : k;
        dx[i] = d;
        taps += j2 - j1;
    }
    return taps;
}

void lto_ellipse_kernel(int k, uint8_t* elem) {
    int* dx = malloc(sizeof(int) * k);
    lto_ellipse_halfwidths(k, dx);
    int c = k / 2;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) elem[i * k + j] = (j >= c - dx[i] && j <= c + dx[i]) ? 1 : 0;
    free(dx);
}

/* definition: min / max over the footprint, out-of-image taps ignored */
void lto_morph_ellipse_brute(const uint8_t* src, int h, int w, int k, int is_dilate, uint8_t* dst) {
    uint8_t* el = malloc((size_t)k * k);
    lto_ellipse_kernel(k, el);
    int r = k / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int acc = is_dilate ? 0 : 255;
            for (int i = 0; i < k; ++i) {
                int yy = y + i - r;
                if (yy < 0 || yy >= h) continue;
                for (int j = 0; j < k; ++j) {
                    int xx = x + j - r;
                    if (!el[i * k + j] || xx < 0 || xx >= w) continue;
                    int v = src[(size_t)yy * w + xx];
                    acc = is_dilate ? (v > acc ? v : acc) : (v < acc ? v : acc);
                }
            }
            dst[(size_t)y * w + x] = (uint8_t)acc;
        }
    free(el);
}

/* Same result, cheaper: the footprint is one horizontal run per row.  For every distinct half-
 * width d a horizontally min/max-filtered plane is built (runs nest, so each plane is derived from
 * the previous one with one op per pixel where the overlap allows it), then the result is the
 * min/max over the k rows.  Cross-checked against the brute-force definition in tests. */
void lto_morph_ellipse(const uint8_t* src, int h, int w, int k, int is_dilate, uint8_t* dst) {
    int r = k / 2;
    int* dx = malloc(sizeof(int) * k);
    lto_ellipse_halfwidths(k, dx);
    /* distinct half widths, ascending */
    int nd = 0, dist[64], slot_of[256];
    for (int d = 0; d <= r; ++d) {
        int used = 0;
        for (int i = 0; i < k; ++i) used |= dx[i] == d;
        slot_of[d] = -1;
        if (used) { slot_of[d] = nd; dist[nd++] = d; }
    }
    const uint8_t neutral = is_dilate ? 0 : 255;
    const int pw = w + 2 * r; /* padded row */
    uint8_t* planes = malloc((size_t)nd * h * w);
    uint8_t* cur = malloc(pw);
    uint8_t* nxt = malloc(pw);
    for (int y = 0; y < h; ++y) {
        memset(cur, neutral, pw);
        memcpy(cur + r, src + (size_t)y * w, w);
        int have = 0; /* cur holds the centred filter of half-width `have` on the padded row */
        for (int s = 0; s < nd; ++s) {
            int d = dist[s];
            while (have < d) {
                /* two windows of half-width `have` centred at x-step and x+step cover
                 * [x-have-step, x+have+step] without a gap iff step <= have; the first step
                 * (have == 0) therefore takes the centre tap as well. */
                int step = have == 0 ? 1 : (d - have < have ? d - have : have);
                memset(nxt, neutral, pw);
                for (int x = step; x < pw - step; ++x) {
                    uint8_t a = cur[x - step], b = cur[x + step];
                    uint8_t m = is_dilate ? (a > b ? a : b) : (a < b ? a : b);
                    if (have == 0) m = is_dilate ? (cur[x] > m ? cur[x] : m) : (cur[x] < m ? cur[x] : m);
                    nxt[x] = m;
                }
                /* cells within `step` of the padded ends stay neutral; they are never read for
                 * an in-image output because the pad is r >= d wide */
                uint8_t* t = cur; cur = nxt; nxt = t;
                have += step;
            }
            memcpy(planes + ((size_t)s * h + y) * w, cur + r, w);
        }
    }
    for (int y = 0; y < h; ++y) {
        uint8_t* o = dst + (size_t)y * w;
        memset(o, neutral, w);
        for (int i = 0; i < k; ++i) {
            int yy = y + i - r;
            if (yy < 0 || yy >= h) continue;
            const uint8_t* p = planes + ((size_t)slot_of[dx[i]] * h + yy) * w;
            if (is_dilate) { for (int x = 0; x < w; ++x) o[x] = p[x] > o[x] ? p[x] : o[x]; }
            else           { for (int x = 0; x < w; ++x) o[x] = p[x] < o[x] ? p[x] : o[x]; }
        }
    }
    free(planes); free(cur); free(nxt); free(dx);
}

/* morphologyEx(TOPHAT) = src - dilate(erode(src))  (lane_tracker.py:210-211) */
void lto_tophat(const uint8_t* src, int h, int w, int k, uint8_t* dst) {
    size_t n = (size_t)h * w;
    uint8_t* t = malloc(n);
    lto_morph_ellipse(src, h, w, k, 0, dst);
    lto_morph_ellipse(dst, h, w, k, 1, t);
    for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(src[i] > t[i] ? src[i] - t[i] : 0);
    free(t);
}

/* morphologyEx(OPEN) = dilate(erode(src))  (lane_tracker.py:238) */
void lto_open(const uint8_t* src, int h, int w, int k, uint8_t* dst) {
    uint8_t* t = malloc((size_t)h * w);
    lto_morph_ellipse(src, h, w, k, 0, t);
    lto_morph_ellipse(t, h, w, k, 1, dst);
    free(t);
}

/* ------------------------------------------------------------------------------------------- */
/* bilateral_adaptive_threshold (lane_tracker.py:14-83); filter2D = correlation, zero border.     */
int lto_bilateral_adaptive_threshold(const uint8_t* img, int h, int w, int ksize, int C, int mode,
                                     int true_value, int false_value, uint8_t* out) {
    if (mode != 0 && mode != 1) return -1;
    const int delta = mode == 0 ? C * ksize : -C * ksize; /* :67-70 */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int p = img[(size_t)y * w + x];
            int sl = 0, sr = 0, su = 0, sd = 0;
            for (int i = 1; i <= ksize; ++i) {
                if (x - i >= 0) sl += img[(size_t)y * w + x - i];
                if (x + i < w) sr += img[(size_t)y * w + x + i];
                if (y - i >= 0) su += img[(size_t)(y - i) * w + x];
                if (y + i < h) sd += img[(size_t)(y + i) * w + x];
            }
            int l = sl - ksize * p + delta, r = sr - ksize * p + delta; /* :61-64, :73-76 */
            int u = su - ksize * p + delta, d = sd - ksize * p + delta;
            int pass = mode == 0 ? ((0 > l && 0 > r) || (0 > u && 0 > d))  /* :79 */
                                 : ((0 < l && 0 < r) || (0 < u && 0 < d)); /* :81 */
            out[(size_t)y * w + x] = (uint8_t)(pass ? true_value : false_value);
        }
    return 0;
}

/* cv2.adaptiveThreshold(src, 255, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY, bs, -C)
 * (lane_tracker.py:217-218), App. A.6 [H/M]: box mean with replicated border, rounded to u8
 * (bs*bs is odd so sum/bs^2 never ties), output 255 iff src - mean > C. */
void lto_adaptive_mean_threshold(const uint8_t* src, int h, int w, int bs, int C, uint8_t* out) {
    const int r = bs / 2, area = bs * bs;
    int32_t* rows = malloc(sizeof(int32_t) * (size_t)h * w); /* horizontal sums */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int j = -r; j <= r; ++j) {
                int xx = x + j;
                xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                s += src[(size_t)y * w + xx];
            }
            rows[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int i = -r; i <= r; ++i) {
                int yy = y + i;
                yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
                s += rows[(size_t)yy * w + x];
            }
            int mean = (2 * s + area) / (2 * area);
            out[(size_t)y * w + x] = (uint8_t)((int)src[(size_t)y * w + x] - mean > C ? 255 : 0);
        }
    free(rows);
}

/* The same two thresholds with running sums, O(1) per pixel instead of O(k): NOT the parity checker (that stays on
 * the loops above, which read like the reference's kernels) -- these exist so that bench.py's cpu_baseline is a fair
 * statement of a CPU path; tests/test_oracle_units.py checks them equal to the loops. */
int lto_bilateral_adaptive_threshold_fast(const uint8_t* img, int h, int w, int ksize, int C, int mode,
                                          int true_value, int false_value, uint8_t* out) {
    if (mode != 0 && mode != 1) return -1;
    const int k = ksize, delta = mode == 0 ? C * k : -C * k;
    int32_t* su = calloc((size_t)w, sizeof(int32_t)); /* sum of the k pixels above, per column */
    int32_t* sd = calloc((size_t)w, sizeof(int32_t)); /* ... below */
    int32_t* pre = malloc(sizeof(int32_t) * ((size_t)w + 1));
    for (int i = 1; i <= k && i < h; ++i)
        for (int x = 0; x < w; ++x) sd[x] += img[(size_t)i * w + x];
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = img + (size_t)y * w;
        pre[0] = 0;
        for (int x = 0; x < w; ++x) pre[x + 1] = pre[x] + row[x];
        for (int x = 0; x < w; ++x) {
            const int p = row[x], kp = k * p;
            const int xl = x - k < 0 ? 0 : x - k, xr = x + 1 + k > w ? w : x + 1 + k;
            const int l = pre[x] - pre[xl] - kp + delta, r = pre[xr] - pre[x + 1] - kp + delta;
            const int u = su[x] - kp + delta, d = sd[x] - kp + delta;
            const int pass = mode == 0 ? ((0 > l && 0 > r) || (0 > u && 0 > d)) : ((0 < l && 0 < r) || (0 < u && 0 < d));
            out[(size_t)y * w + x] = (uint8_t)(pass ? true_value : false_value);
        }
        if (y + 1 < h) { /* slide both column windows to row y + 1 */
            const uint8_t* next = img + (size_t)(y + 1) * w;
            const uint8_t* leave = y - k >= 0 ? img + (size_t)(y - k) * w : NULL;
            const uint8_t* enter = y + 1 + k < h ? img + (size_t)(y + 1 + k) * w : NULL;
            for (int x = 0; x < w; ++x) {
                su[x] += row[x] - (leave ? leave[x] : 0);
                sd[x] += (enter ? enter[x] : 0) - next[x];
            }
        }
    }
    free(su); free(sd); free(pre);
    return 0;
}

void lto_adaptive_mean_threshold_fast(const uint8_t* src, int h, int w, int bs, int C, uint8_t* out) {
    const int r = bs / 2, area = bs * bs;
    int32_t* hs = malloc(sizeof(int32_t) * (size_t)h * w); /* horizontal window sums, replicated border */
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = src + (size_t)y * w;
        int s = 0;
        for (int j = -r; j <= r; ++j) s += row[j < 0 ? 0 : (j >= w ? w - 1 : j)];
        for (int x = 0; x < w; ++x) {
            hs[(size_t)y * w + x] = s;
            const int in = x + r + 1, outx = x - r;
            s += row[in >= w ? w - 1 : in] - row[outx < 0 ? 0 : outx];
        }
    }
    int32_t* vs = calloc((size_t)w, sizeof(int32_t));
    for (int i = -r; i <= r; ++i) {
        const int32_t* q = hs + (size_t)(i < 0 ? 0 : (i >= h ? h - 1 : i)) * w;
        for (int x = 0; x < w; ++x) vs[x] += q[x];
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const int mean = (2 * vs[x] + area) / (2 * area);
            out[(size_t)y * w + x] = (uint8_t)((int)src[(size_t)y * w + x] - mean > C ? 255 : 0);
        }
        const int in = y + r + 1, outy = y - r;
        const int32_t* qi = hs + (size_t)(in >= h ? h - 1 : in) * w;
        const int32_t* qo = hs + (size_t)(outy < 0 ? 0 : outy) * w;
        for (int x = 0; x < w; ++x) vs[x] += qi[x] - qo[x];
    }
    free(hs); free(vs);
}

/* ------------------------------------------------------------------------------------------- */
/* LaneTracker.filter_lane_points (lane_tracker.py:183-240); fast != 0: the running-sum thresholds (cpu_baseline only) */
static int filter_impl(const uint8_t* bev, int h, int w, const lto_filter_params* p, uint8_t* mask, uint8_t* planes, int fast) {
    int (*bil)(const uint8_t*, int, int, int, int, int, int, int, uint8_t*) =
        fast ? lto_bilateral_adaptive_threshold_fast : lto_bilateral_adaptive_threshold;
    void (*ada)(const uint8_t*, int, int, int, int, uint8_t*) = fast ? lto_adaptive_mean_threshold_fast : lto_adaptive_mean_threshold;
    if (p->filter_type != 0 && p->filter_type != 1) return -1; /* :220 ValueError */
    const size_t n = (size_t)h * w;
    uint8_t* buf = malloc(n * 8);
    uint8_t *R = buf, *B = buf + n, *thR = buf + 2 * n, *thB = buf + 3 * n;
    uint8_t *tr = buf + 4 * n, *tb = buf + 5 * n, *merged = buf + 6 * n, *tmp = buf + 7 * n;
    lto_channel_r(bev, (int)n, R);                      /* :207 */
    lto_lab_b(bev, (int)n, B);                          /* :208 */
    if (p->filter_type == 0) {
        lto_tophat(R, h, w, 29, thR);                   /* :210 (computed on both branches upstream; unused by 'neighborhood') */
        lto_tophat(B, h, w, 55, thB);                   /* :211 */
        bil(thR, h, w, p->ksize_r, p->C_r, 0, 255, 0, tr); /* :214 */
        bil(thB, h, w, p->ksize_b, p->C_b, 0, 255, 0, tb); /* :215 */
    } else {
        ada(R, h, w, p->ksize_r, p->C_r, tr); /* :217 */
        ada(B, h, w, p->ksize_b, p->C_b, tb); /* :218 */
    }
    if (p->mask_noise) {                                /* :221-231 */
        bil(B, h, w, p->ksize_noise, p->C_noise, 0, 255, 0, tmp); /* :224 */
        for (size_t i = 0; i < n; ++i) {
            int part1 = B[i] >= p->noise_thresh;        /* inRange(b, thresh, 255) :223 */
            int noise = (!part1) || tmp[i];             /* :225 */
            merged[i] = (uint8_t)(((tr[i] || tb[i]) && noise) ? 255 : 0); /* :229-231 */
        }
    } else {
        for (size_t i = 0; i < n; ++i) merged[i] = (uint8_t)((tr[i] || tb[i]) ? 255 : 0); /* :233-235 */
    }
    lto_open(merged, h, w, 5, mask);                    /* :238 */
    if (planes) {
        memcpy(planes, R, n);
        memcpy(planes + n, B, n);
        if (p->filter_type == 0) { memcpy(planes + 2 * n, thR, n); memcpy(planes + 3 * n, thB, n); }
    }
    free(buf);
    return 0;
}

int lto_filter_lane_points(const uint8_t* bev, int h, int w, const lto_filter_params* p, uint8_t* mask, uint8_t* planes) {
    return filter_impl(bev, h, w, p, mask, planes, 0);
}
int lto_filter_lane_points_fast(const uint8_t* bev, int h, int w, const lto_filter_params* p, uint8_t* mask, uint8_t* planes) {
    return filter_impl(bev, h, w, p, mask, planes, 1);
}

static int mask_impl(const lto_calib* c, const uint8_t* frame, const lto_filter_params* p, uint8_t* mask, int fast) {
    uint8_t* bev = malloc((size_t)c->warp_w * c->warp_h * 3);
    lto_front_end(c, frame, bev);                       /* :832, :834 */
    int rc = filter_impl(bev, c->warp_h, c->warp_w, p, mask, NULL, fast); /* :837 */
    free(bev);
    return rc;
}
int lto_mask_from_frame(const lto_calib* c, const uint8_t* frame, const lto_filter_params* p, uint8_t* mask) {
    return mask_impl(c, frame, p, mask, 0);
}

/* ------------------------------------------------------------------------------------------- */
/* LaneTracker.sliding_window_search (lane_tracker.py:242-447), SURVEY App. B.                    */

/* np.convolve(ones(ww), cnt) 'full', then the reference's first/last-argmax logic on conv[lo:hi].
 * Returns 0 if the slice is empty or all zero (np.any false); else stores first/last argmax
 * (indices relative to lo). */
static int box_argmax(const int64_t* prefix, int ncnt, int ww, int lo, int hi, int* first, int* last) {
    /* conv[k] = sum cnt[k-ww+1 .. k] clipped to [0,ncnt) ; prefix[i] = sum cnt[0..i) */
    int64_t best = 0;
    int f = -1, l = -1;
    for (int k = lo; k < hi; ++k) {
        int a = k - ww + 1, b = k + 1;
        if (a < 0) a = 0;
        if (b > ncnt) b = ncnt;
        int64_t v = b > a ? prefix[b] - prefix[a] : 0;
        if (v > best) { best = v; f = l = k; }
        else if (v == best && best > 0) l = k;
    }
    if (best == 0) return 0;
    *first = f - lo;
    *last = l - lo;
    return 1;
}

/* img[r0:r1, c-hw:c+hw].nonzero() with NumPy slice semantics (a negative start wraps => empty
 * for every case reachable here; stop is clipped to w).  Appends in row-major order. */
static void roi_nonzero(const uint8_t* mask, int w, int r0, int r1, int c, int hw, int32_t* ys,
                        int32_t* xs, int32_t* n) {
    int a = c - hw, b = c + hw;
    if (a < 0) {
        /* Python: start = w + a (if still < 0 it clips to 0); stop b: negative wraps too */
        int start = w + a < 0 ? 0 : w + a;
        int stop = b < 0 ? (w + b < 0 ? 0 : w + b) : (b > w ? w : b);
        if (start >= stop) return;
        /* non-empty wrapped slice: offsets are then added back as `+ c - hw` by the reference,
         * i.e. reported x = (index within slice) + a. */
        for (int y = r0; y < r1; ++y)
            for (int x = start; x < stop; ++x)
                if (mask[(size_t)y * w + x]) { ys[*n] = y; xs[*n] = (x - start) + a; ++*n; }
        return;
    }
    if (b > w) b = w;
    for (int y = r0; y < r1; ++y)
        for (int x = a; x < b; ++x)
            if (mask[(size_t)y * w + x]) { ys[*n] = y; xs[*n] = x; ++*n; }
}

static void col_counts(const uint8_t* mask, int w, int r0, int r1, int c0, int c1, int32_t* cnt,
                       int64_t* prefix) {
    int n = c1 - c0;
    for (int j = 0; j < n; ++j) cnt[j] = 0;
    for (int y = r0; y < r1; ++y)
        for (int j = 0; j < n; ++j) cnt[j] += mask[(size_t)y * w + c0 + j]; /* np.sum of pixel values (:290, :350) */
    prefix[0] = 0;
    for (int j = 0; j < n; ++j) prefix[j + 1] = prefix[j] + cnt[j];
}

typedef struct side_state {
    int c, ns, lo, hi, ndiff, last_diff, ncent, nroi;
    int32_t *cent, *ys, *xs, *n;
} side_state;

int lto_sliding_window_search(const uint8_t* mask, int h, int w, const lto_search_params* p,
                              int32_t* ly, int32_t* lx, int32_t* nl,
                              int32_t* ry, int32_t* rx, int32_t* nr,
                              int32_t* lcent, int32_t* nlc, int32_t* rcent, int32_t* nrc) {
    const int ww = p->window_width, wh = p->window_height, hw = (int)(ww / 2.0); /* int(window_width/2) */
    const int img_width = w;
    const int img_height = h - p->ignore_bottom;                                /* :277 */
    const int img_center = (int)(img_width / 2.0);                              /* :278 */
    const int y_start = (int)((1 - p->start_slice) * img_height);               /* :279 */
    const int nlevels = (int)((p->partial * img_height) / wh);                  /* :282 */
    const int limit = p->no_success_limit;
    int32_t* cnt = malloc(sizeof(int32_t) * (size_t)(w + 1));
    int64_t* prefix = malloc(sizeof(int64_t) * (size_t)(w + 2));
    *nl = *nr = 0;
    side_state S[2];
    memset(S, 0, sizeof S);
    S[0].cent = lcent; S[0].ys = ly; S[0].xs = lx; S[0].n = nl;
    S[1].cent = rcent; S[1].ys = ry; S[1].xs = rx; S[1].n = nr;

    /* level 0 (:290-332) */
    for (int s = 0; s < 2; ++s) {
        int c0 = s == 0 ? p->ignore_sides : img_center;
        int c1 = s == 0 ? img_center : img_width - p->ignore_sides;
        int off = s == 0 ? p->ignore_sides : img_center;
        int found = 0, first = 0, last = 0;
        if (c1 > c0 && img_height > y_start) {
            col_counts(mask, w, y_start < 0 ? 0 : y_start, img_height, c0, c1, cnt, prefix);
            found = box_argmax(prefix, c1 - c0, ww, 0, (c1 - c0) + ww - 1, &first, &last);
        }
        if (found) {
            int max_center = (int)((first + last) / 2.0);                        /* :296 */
            S[s].c = max_center - hw + off;                                     /* :297 / :317 */
            roi_nonzero(mask, w, img_height - wh, img_height, S[s].c, hw, S[s].ys, S[s].xs, S[s].n);
            S[s].nroi++;
        } else {
            S[s].c = (int)(img_width * (s == 0 ? 0.4 : 0.6));                    /* :308 / :328 */
        }
        S[s].cent[S[s].ncent++] = S[s].c;                                        /* :331-332 */
        S[s].lo = -p->search_range;                                              /* :340-343 */
        S[s].hi = p->search_range;
    }

    for (int level = 1; level < nlevels; ++level) {                              /* :346 */
        const int r0 = img_height - (1 + level) * wh, r1 = img_height - level * wh;
        col_counts(mask, w, r0, r1, 0, w, cnt, prefix);                          /* :350 */
        const int conv_len = w + ww - 1;                                         /* :351 */
        for (int s = 0; s < 2; ++s) {                                            /* left, then right */
            side_state* me = &S[s];
            side_state* other = &S[1 - s];
            if (me->ns >= limit) continue;                                       /* :354 / :395 */
            int lo_i = me->c + me->lo + hw; if (lo_i < 0) lo_i = 0;              /* :356 */
            int hi_i = me->c + me->hi + hw; if (hi_i > img_width) hi_i = img_width; /* :357 */
            /* conv[lo_i:hi_i] with Python slice semantics */
            int a = lo_i > conv_len ? conv_len : lo_i;
            int b = hi_i < 0 ? (conv_len + hi_i < 0 ? 0 : conv_len + hi_i) : (hi_i > conv_len ? conv_len : hi_i);
            int first = 0, last = 0;
            int found = b > a ? box_argmax(prefix, w, ww, a, b, &first, &last) : 0; /* :360 */
            if (found) {
                int max_center = (int)ceil((first + last) / 2.0);                /* :363 */
                int newc = max_center + lo_i - hw;                               /* :364 */
                me->cent[me->ncent++] = newc;                                    /* :365 */
                me->last_diff = newc - me->c;                                    /* :366 */
                me->ndiff++;
                me->c = newc;
                me->ns = 0;                                                      /* :368 */
                roi_nonzero(mask, w, r0, r1, me->c, hw, me->ys, me->xs, me->n);  /* :371-378 */
                me->nroi++;
                int t = (int)(p->mu * me->last_diff);                            /* :380-381 */
                me->lo += t;
                me->hi += t;
            } else {
                if (other->ndiff > 0 && other->ns == 0) me->c += other->last_diff; /* :385-387 */
                me->cent[me->ncent++] = me->c;
                me->ns++;                                                        /* :390 */
                if (me->ns >= limit) {                                           /* :391-392 */
                    /* del cent[-limit:]  (limit == 0 would be del cent[0:] = everything) */
                    int del = limit > 0 ? (limit < me->ncent ? limit : me->ncent) : me->ncent;
                    me->ncent -= del;
                }
            }
        }
    }
    free(cnt);
    free(prefix);
    *nlc = S[0].ncent;
    *nrc = S[1].ncent;
    /* :432-447 */
    return (S[0].nroi > 0 && S[1].nroi > 0 && *nl > 0 && *nr > 0) ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------- */
/* LaneTracker.band_search (lane_tracker.py:449-500) */
int lto_band_search(const uint8_t* mask, int h, int w, const lto_search_params* p,
                    const double lc[3], const double rc[3],
                    int32_t* ly, int32_t* lx, int32_t* nl, int32_t* ry, int32_t* rx, int32_t* nr) {
    /* :465  img1[h-ignore_bottom:, :] = 0  -> rows [h-ignore_bottom, h) are dropped */
    int bottom = h - p->ignore_bottom;
    if (bottom > h) bottom = h;
    if (bottom < 0) bottom = 0;
    /* :466  img1[:h*(1-partial), :] = 0    -> rows [0, int(h*(1-partial))) are dropped */
    int top = (int)(h * (1 - p->partial));
    if (top < 0) top = 0;
    const double bw = p->bandwidth;
    *nl = *nr = 0;
    for (int y = top; y < bottom; ++y) {
        double y2 = (double)((int64_t)y * y), yd = (double)y;
        double tl = lc[0] * y2 + lc[1] * yd + lc[2];     /* :474-476 evaluation order */
        double tr = rc[0] * y2 + rc[1] * yd + rc[2];
        double llo = tl - bw, lhi = tl + bw, rlo = tr - bw, rhi = tr + bw;
        for (int x = 0; x < w; ++x) {
            if (!mask[(size_t)y * w + x]) continue;
            double xd = (double)x;
            if (xd > llo && xd < lhi) { ly[*nl] = y; lx[*nl] = x; ++*nl; }
            if (xd > rlo && xd < rhi) { ry[*nr] = y; rx[*nr] = x; ++*nr; }
        }
    }
    return (*nl != 0 && *nr != 0) ? 1 : 0;               /* :491 */
}

/* ------------------------------------------------------------------------------------------- */
/* np.polyfit(y, x, 2) (lane_tracker.py:506-507): lhs = vander(y,3) with columns scaled to unit
 * 2-norm, least-squares solve, unscale.  NumPy solves with LAPACK gelsd (SVD); a Householder QR
 * with column pivoting on the same scaled matrix gives the same solution to rounding for full
 * rank.  Rank-deficient inputs (fewer than 3 distinct y) get the minimum-norm solution through a
 * small one-sided Jacobi SVD, like gelsd with rcond = n*eps. */
static void jacobi_svd3(double A[3][3], double U[3][3], double S[3], double V[3][3]);

int lto_polyfit2(const int32_t* y, const int32_t* x, int n, double coef[3]) {
    /* normal equations are formed in long double from the scaled columns only to obtain R'R;
     * the actual solve below is on the QR of the scaled Vandermonde, done explicitly. */
    coef[0] = coef[1] = coef[2] = 0;
    if (n <= 0) return 0;
    double sc[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        double yy = y[i];
        sc[0] += (yy * yy) * (yy * yy);
        sc[1] += yy * yy;
        sc[2] += 1.0;
    }
    for (int j = 0; j < 3; ++j) sc[j] = sqrt(sc[j]);
    for (int j = 0; j < 3; ++j) if (sc[j] == 0) sc[j] = 1; /* all-zero column */
    /* thin QR by modified Gram-Schmidt with reorthogonalisation on an n x 3 matrix is enough
     * here, but to stay SVD-faithful reduce to the 3x3 problem R c = Q'b via Householder. */
    double* A = malloc(sizeof(double) * (size_t)n * 4);
    for (int i = 0; i < n; ++i) {
        double yy = y[i];
        A[i * 4 + 0] = yy * yy / sc[0];
        A[i * 4 + 1] = yy / sc[1];
        A[i * 4 + 2] = 1.0 / sc[2];
        A[i * 4 + 3] = x[i];
    }
    int m = n;
    for (int k = 0; k < 3 && k < m; ++k) {
        double nrm = 0;
        for (int i = k; i < m; ++i) nrm += A[i * 4 + k] * A[i * 4 + k];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        double alpha = A[k * 4 + k] > 0 ? -nrm : nrm;
        double v0 = A[k * 4 + k] - alpha;
        double vnorm2 = v0 * v0;
        for (int i = k + 1; i < m; ++i) vnorm2 += A[i * 4 + k] * A[i * 4 + k];
        if (vnorm2 == 0) continue;
        for (int j = k + 1; j < 4; ++j) {
            double dot = v0 * A[k * 4 + j];
            for (int i = k + 1; i < m; ++i) dot += A[i * 4 + k] * A[i * 4 + j];
            double f = 2 * dot / vnorm2;
            A[k * 4 + j] -= f * v0;
            for (int i = k + 1; i < m; ++i) A[i * 4 + j] -= f * A[i * 4 + k];
        }
        A[k * 4 + k] = alpha;
        for (int i = k + 1; i < m; ++i) A[i * 4 + k] = 0;
    }
    double R[3][3] = {{0}}, qb[3] = {0, 0, 0};
    for (int i = 0; i < 3 && i < m; ++i) {
        for (int j = i; j < 3; ++j) R[i][j] = A[i * 4 + j];
        qb[i] = A[i * 4 + 3];
    }
    free(A);
    /* minimum-norm least squares of the 3x3 triangular system through its SVD (rcond = n*eps) */
    double U[3][3], S[3], V[3][3];
    jacobi_svd3(R, U, S, V);
    double smax = S[0] > S[1] ? (S[0] > S[2] ? S[0] : S[2]) : (S[1] > S[2] ? S[1] : S[2]);
    double rcond = (double)n * 2.220446049250313e-16;
    double c[3] = {0, 0, 0};
    int rank = 0;
    for (int k = 0; k < 3; ++k) {
        if (!(S[k] > rcond * smax)) continue;
        ++rank;
        double ub = (U[0][k] * qb[0] + U[1][k] * qb[1] + U[2][k] * qb[2]) / S[k];
        for (int j = 0; j < 3; ++j) c[j] += V[j][k] * ub;
    }
    for (int j = 0; j < 3; ++j) coef[j] = c[j] / sc[j];
    return rank;
}

/* one-sided Jacobi SVD of a 3x3 matrix: A = U diag(S) V' */
static void jacobi_svd3(double A[3][3], double U[3][3], double S[3], double V[3][3]) {
    double B[3][3];
    memcpy(B, A, sizeof B);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < 3; ++i) { a += B[i][p] * B[i][p]; b += B[i][q] * B[i][q]; g += B[i][p] * B[i][q]; }
                off += fabs(g);
                if (g == 0 || fabs(g) <= 1e-300) continue;
                double zeta = (b - a) / (2 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                double cs = 1 / sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    double bp = B[i][p], bq = B[i][q];
                    B[i][p] = cs * bp - sn * bq;
                    B[i][q] = sn * bp + cs * bq;
                    double vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq;
                    V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-300) break;
        /* convergence: columns orthogonal to working precision */
        double ok = 1;
        for (int p = 0; p < 2 && ok; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < 3; ++i) { a += B[i][p] * B[i][p]; b += B[i][q] * B[i][q]; g += B[i][p] * B[i][q]; }
                if (fabs(g) > 1e-17 * sqrt(a * b)) ok = 0;
            }
        if (ok) break;
    }
    for (int k = 0; k < 3; ++k) {
        double s = 0;
        for (int i = 0; i < 3; ++i) s += B[i][k] * B[i][k];
        S[k] = sqrt(s);
        for (int i = 0; i < 3; ++i) U[i][k] = S[k] > 0 ? B[i][k] / S[k] : 0;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* One independent frame with a fresh tracker: find_lane_points (sliding window) + fit_poly.      */
static int frame_impl(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp, const lto_search_params* sp,
                      uint8_t* mask_out, double coef[6], int32_t counts[3], int fast);
int lto_frame_sws_fit(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp,
                      const lto_search_params* sp, uint8_t* mask_out, double coef[6],
                      int32_t counts[3]) {
    return frame_impl(c, frame, fp, sp, mask_out, coef, counts, 0);
}
/* the same with the running-sum thresholds: bench.py's cpu_baseline (results equal, tests/test_oracle_units.py) */
int lto_frame_sws_fit_fast(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp,
                           const lto_search_params* sp, uint8_t* mask_out, double coef[6],
                           int32_t counts[3]) {
    return frame_impl(c, frame, fp, sp, mask_out, coef, counts, 1);
}
static int frame_impl(const lto_calib* c, const uint8_t* frame, const lto_filter_params* fp, const lto_search_params* sp,
                      uint8_t* mask_out, double coef[6], int32_t counts[3], int fast) {
    const int h = c->warp_h, w = c->warp_w;
    uint8_t* mask = mask_out ? mask_out : malloc((size_t)h * w);
    int rc = mask_impl(c, frame, fp, mask, fast);
    if (rc) { if (!mask_out) free(mask); return rc; }
    int nlev = (int)((sp->partial * (h - sp->ignore_bottom)) / sp->window_height);
    if (nlev < 1) nlev = 1;
    size_t cap = (size_t)nlev * sp->window_height * sp->window_width + 16;
    int32_t* buf = malloc(sizeof(int32_t) * (cap * 4 + 2 * (size_t)(nlev + 2)));
    int32_t *ly = buf, *lx = buf + cap, *ry = buf + 2 * cap, *rx = buf + 3 * cap;
    int32_t *lcent = buf + 4 * cap, *rcent = lcent + nlev + 2;
    int32_t nl, nr, nlc, nrc;
    int det = lto_sliding_window_search(mask, h, w, sp, ly, lx, &nl, ry, rx, &nr, lcent, &nlc, rcent, &nrc);
    for (int i = 0; i < 6; ++i) coef[i] = 0;
    if (det) {
        lto_polyfit2(ly, lx, nl, coef);
        lto_polyfit2(ry, rx, nr, coef + 3);
    }
    counts[0] = nl; counts[1] = nr; counts[2] = det;
    free(buf);
    if (!mask_out) free(mask);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Presentation stage (draw_lane, lane_tracker.py:629-662; create_split_view, utils.py:57-103).   */
/* OpenCV drawing.cpp restated from memory [M]: Line() walks the 8-connected LineIterator from the
 * leftmost end point; fillPoly = CollectPolyEdges (draws every edge, records non-horizontal edges
 * in 16.16 fixed point) + FillEdgeCollection (even-odd spans, top-inclusive/bottom-exclusive).     */
static void put_px(uint8_t* img, int h, int w, int ch, int x, int y, const uint8_t* color) {
    if (x < 0 || x >= w || y < 0 || y >= h) return;
    for (int c = 0; c < ch; ++c) img[((size_t)y * w + x) * ch + c] = color[c];
}

static void draw_line8(uint8_t* img, int h, int w, int ch, int x0, int y0, int x1, int y1, const uint8_t* color) {
    if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }   /* leftToRight */
    int dx = x1 - x0, dy = y1 - y0;
    int sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    int steep = dy > dx;
    int major = steep ? dy : dx, minor = steep ? dx : dy;
    int err = major - 2 * minor, x = x0, y = y0;
    for (int i = 0; i <= major; ++i) {
        put_px(img, h, w, ch, x, y, color);
        int neg = err < 0;
        err += -2 * minor + (neg ? 2 * major : 0);
        if (steep) { y += sy; if (neg) x += 1; }
        else { x += 1; if (neg) y += sy; }
    }
}

typedef struct { int y0, y1; int64_t x, dx; } poly_edge;
static int cmp_i64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return x < y ? -1 : x > y;
}

void lto_fill_poly(uint8_t* img, int h, int w, int ch, const int32_t* pts, int npts, const uint8_t* color) {
    if (npts <= 0) return;
    poly_edge* edges = malloc(sizeof(poly_edge) * (size_t)npts);
    int ne = 0, ymin = INT_MAX, ymax = INT_MIN;
    for (int i = 0; i < npts; ++i) {
        int j = (i + npts - 1) % npts;                      /* edge from the previous vertex to this one */
        int xa = pts[2 * j], ya = pts[2 * j + 1], xb = pts[2 * i], yb = pts[2 * i + 1];
        draw_line8(img, h, w, ch, xa, ya, xb, yb, color);
        if (ya == yb) continue;
        poly_edge e;
        if (ya < yb) { e.y0 = ya; e.y1 = yb; e.x = (int64_t)xa << 16; }
        else { e.y0 = yb; e.y1 = ya; e.x = (int64_t)xb << 16; }
        e.dx = (((int64_t)(ya < yb ? xb - xa : xa - xb)) << 16) / (e.y1 - e.y0);
        edges[ne++] = e;
        if (e.y0 < ymin) ymin = e.y0;
        if (e.y1 > ymax) ymax = e.y1;
    }
    int64_t* xs = malloc(sizeof(int64_t) * (size_t)(ne + 1));
    for (int y = ymin; y < ymax; ++y) {
        int k = 0;
        for (int i = 0; i < ne; ++i)
            if (edges[i].y0 <= y && y < edges[i].y1) xs[k++] = edges[i].x + (int64_t)(y - edges[i].y0) * edges[i].dx;
        qsort(xs, (size_t)k, sizeof(int64_t), cmp_i64);
        if (y < 0 || y >= h) continue;
        for (int i = 0; i + 1 < k; i += 2) {
            int64_t xa = (xs[i] + 0xffff) >> 16, xb = xs[i + 1] >> 16;    /* ceil .. floor */
            for (int64_t x = xa < 0 ? 0 : xa; x <= xb && x < w; ++x) put_px(img, h, w, ch, (int)x, y, color);
        }
    }
    free(xs);
    free(edges);
}

void lto_add_weighted_u8(const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma,
                         size_t n, uint8_t* out) {
    const float fa = (float)alpha, fb = (float)beta, fg = (float)gamma;
    for (size_t i = 0; i < n; ++i) {
        volatile float t0 = (float)a[i] * fa;          /* volatile: no fused multiply-add, f32 roundings as written */
        volatile float t1 = (float)b[i] * fb;
        volatile float t = t0 + t1;
        t = t + fg;
        long r = lrintf(t);                            /* round-half-even (default rounding mode) */
        out[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

void lto_draw_lane(const lto_calib* c, const double Minv[9], const uint8_t* img,
                   const int32_t* left_y, const int32_t* left_x, int n_left,
                   const int32_t* right_y, const int32_t* right_x, int n_right, uint8_t* out) {
    const int bh = c->warp_h, bw = c->warp_w, ih = c->img_h, iw = c->img_w;
    uint8_t* lane = calloc((size_t)bh * bw, 3);
    int npts = n_left + n_right;
    int32_t* pts = malloc(sizeof(int32_t) * 2 * (size_t)(npts > 0 ? npts : 1));
    for (int i = 0; i < n_left; ++i) { pts[2 * i] = left_x[i]; pts[2 * i + 1] = left_y[i]; }
    for (int i = 0; i < n_right; ++i) {                          /* np.flipud of the right-hand points */
        pts[2 * (n_left + i)] = right_x[n_right - 1 - i];
        pts[2 * (n_left + i) + 1] = right_y[n_right - 1 - i];
    }
    const uint8_t green[3] = {0, 255, 0};
    lto_fill_poly(lane, bh, bw, 3, pts, npts, green);
    /* warpPerspective(lane, Minv, (iw, ih)): the generic map builder with M := Minv, output := camera size */
    lto_calib u = *c;
    memcpy(u.M, Minv, sizeof u.M);
    u.warp_w = iw;
    u.warp_h = ih;
    size_t n = (size_t)iw * ih;
    int16_t* xy = malloc(n * 4);
    uint16_t* al = malloc(n * 2);
    lto_warp_map(&u, xy, al);
    uint8_t* unwarped = malloc(n * 3);
    lto_remap_bilinear_c3(lane, bh, bw, xy, al, ih, iw, unwarped);
    lto_add_weighted_u8(img, 1.0, unwarped, 0.3, 0.0, n * 3, out);
    free(unwarped); free(al); free(xy); free(pts); free(lane);
}

/* cv::resize INTER_LINEAR, 8-bit: src coordinate (d + 0.5) * scale - 0.5, floor -> tap + fraction,
 * left clamp (fraction 0), right clamp (tap sw-1, fraction... taps replicate); coefficients
 * saturate_cast<short>(f * 2048); result (sum of 4 products + 2^21) >> 22 evaluated as OpenCV's
 * two-stage: rows in int with 11-bit x coefficients, then ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2. */
void lto_resize_linear_u8(const uint8_t* src, int sh, int sw, int ch, int dh, int dw, uint8_t* dst) {
    int* xo = malloc(sizeof(int) * (size_t)dw);
    int* yo = malloc(sizeof(int) * (size_t)dh);
    short* xa = malloc(sizeof(short) * 2 * (size_t)dw);
    short* ya = malloc(sizeof(short) * 2 * (size_t)dh);
    const double sx = (double)sw / dw, sy = (double)sh / dh;
    for (int d = 0; d < dw; ++d) {
        float f = (float)((d + 0.5) * sx - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= sw - 1) { s = sw - 1; f = 0; }
        xo[d] = s;
        xa[2 * d] = (short)lrintf((1.f - f) * 2048);
        xa[2 * d + 1] = (short)lrintf(f * 2048);
    }
    for (int d = 0; d < dh; ++d) {
        float f = (float)((d + 0.5) * sy - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= sh - 1) { s = sh - 1; f = 0; }
        yo[d] = s;
        ya[2 * d] = (short)lrintf((1.f - f) * 2048);
        ya[2 * d + 1] = (short)lrintf(f * 2048);
    }
    for (int y = 0; y < dh; ++y) {
        const int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        for (int x = 0; x < dw; ++x) {
            const int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            for (int c = 0; c < ch; ++c) {
                int S0 = src[((size_t)y0 * sw + x0) * ch + c] * xa[2 * x] + src[((size_t)y0 * sw + x1) * ch + c] * xa[2 * x + 1];
                int S1 = src[((size_t)y1 * sw + x0) * ch + c] * xa[2 * x] + src[((size_t)y1 * sw + x1) * ch + c] * xa[2 * x + 1];
                int v = (((ya[2 * y] * (S0 >> 4)) >> 16) + ((ya[2 * y + 1] * (S1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * dw + x) * ch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    free(ya); free(xa); free(yo); free(xo);
}
