// Calibration-constant tables, built once per context on the host in f64 and uploaded.
//
// They replace per-call work hidden inside the reference's cv2 calls:
//   cv2.undistort            lane_tracker.py:832  -> undistort remap table (rows the warp needs only)
//   cv2.warpPerspective      lane_tracker.py:834  -> warp remap table
//   cv2.cvtColor(RGB2LAB)    lane_tracker.py:208  -> sRGB gamma LUT, cube-root LUT, fixed-point matrix
//   cv2.getStructuringElement lane_tracker.py:203-205 -> per-row half-widths of the ellipses
// The arithmetic follows OpenCV's published operation order; this translation
// unit must be compiled with -ffp-contract=off so that no FMA changes a last bit.
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

#include "lt_internal.h"

namespace lt {
namespace {

constexpr int kInterBits = 5;
constexpr int kInterTab = 1 << kInterBits;

inline int cv_round(double v) {  // cvRound + saturate_cast<int>
    if (!(v > (double)INT_MIN)) return INT_MIN;
    if (!(v < (double)INT_MAX)) return INT_MAX;
    return (int)std::nearbyint(v);
}
inline int16_t clamp_s16(int v) { return (int16_t)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v)); }

struct Mat3 {
    double m[9];
    double operator()(int r, int c) const { return m[r * 3 + c]; }
};

// closed-form inverse (cofactors * 1/det), the path OpenCV takes for 3x3 f64 matrices
bool invert3(const Mat3& a, Mat3& out) {
    const double c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
    const double c01 = a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0);
    const double c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
    double det = a(0, 0) * c00 - a(0, 1) * c01 + a(0, 2) * c02;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    out.m[0] = c00 * id;
    out.m[1] = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
    out.m[2] = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
    out.m[3] = (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) * id;
    out.m[4] = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
    out.m[5] = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
    out.m[6] = c02 * id;
    out.m[7] = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
    out.m[8] = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
    return true;
}

inline void store_fixed(RemapTable& t, size_t o, int X, int Y, bool saturate) {
    int sx = X >> kInterBits, sy = Y >> kInterBits;
    t.xy[o * 2] = saturate ? clamp_s16(sx) : (int16_t)sx;
    t.xy[o * 2 + 1] = saturate ? clamp_s16(sy) : (int16_t)sy;
    t.frac[o] = (uint16_t)((Y & (kInterTab - 1)) * kInterTab + (X & (kInterTab - 1)));
}

}  // namespace

// cv::warpPerspective without WARP_INVERSE_MAP: M is inverted, then for every destination pixel
// X = (m0 x + m1 y + m2) * 32/W etc., evaluated per 64-wide block as (row term) + m0*x1.
void build_warp_table(const lt_calib& c, RemapTable& t) {
    Mat3 M, m;
    std::memcpy(M.m, c.M, sizeof M.m);
    if (!invert3(M, m)) std::memset(m.m, 0, sizeof m.m);
    t.rows = c.warp_h;
    t.cols = c.warp_w;
    t.xy.assign((size_t)t.rows * t.cols * 2, 0);
    t.frac.assign((size_t)t.rows * t.cols, 0);
    constexpr int kBlockW = 64;
    for (int y = 0; y < t.rows; ++y)
        for (int x0 = 0; x0 < t.cols; x0 += kBlockW) {
            const double X0 = m.m[0] * x0 + m.m[1] * y + m.m[2];
            const double Y0 = m.m[3] * x0 + m.m[4] * y + m.m[5];
            const double W0 = m.m[6] * x0 + m.m[7] * y + m.m[8];
            const int bw = t.cols - x0 < kBlockW ? t.cols - x0 : kBlockW;
            for (int x1 = 0; x1 < bw; ++x1) {
                double w = W0 + m.m[6] * x1;
                w = w != 0.0 ? kInterTab / w : 0.0;
                double fx = (X0 + m.m[0] * x1) * w, fy = (Y0 + m.m[3] * x1) * w;
                fx = std::fmax((double)INT_MIN, std::fmin((double)INT_MAX, fx));
                fy = std::fmax((double)INT_MIN, std::fmin((double)INT_MAX, fy));
                store_fixed(t, (size_t)y * t.cols + x0 + x1, cv_round(fx), cv_round(fy), true);
            }
        }
}

// camera rows [r0, r1) that at least one in-image bilinear tap of the warp reads
void warp_source_rows(const lt_calib& c, const RemapTable& warp, int& r0, int& r1) {
    int lo = INT_MAX, hi = INT_MIN;
    const size_t n = (size_t)warp.rows * warp.cols;
    for (size_t o = 0; o < n; ++o) {
        const int sx = warp.xy[o * 2], sy = warp.xy[o * 2 + 1];
        if (sx + 1 < 0 || sx >= c.img_w) continue;
        for (int yy = sy; yy <= sy + 1; ++yy)
            if (yy >= 0 && yy < c.img_h) {
                lo = yy < lo ? yy : lo;
                hi = yy > hi ? yy : hi;
            }
    }
    if (lo > hi) {
        r0 = r1 = 0;
        return;
    }
    r0 = lo;
    r1 = hi + 1;
}

// cv::undistort = initUndistortRectifyMap (R = I, new camera matrix = K with cy shifted to the
// stripe origin) + remap, in stripes of max(1, 4096/cols) rows.
void build_undistort_table(const lt_calib& c, int r0, int r1, RemapTable& t) {
    const int w = c.img_w;
    int stripe = 4096 / (w > 1 ? w : 1);
    stripe = stripe < 1 ? 1 : (stripe > c.img_h ? c.img_h : stripe);
    t.rows = r1 - r0;
    t.cols = w;
    t.xy.assign((size_t)t.rows * w * 2 + 2, 0);
    t.frac.assign((size_t)t.rows * w + 1, 0);
    const double fx = c.cam_matrix[0], fy = c.cam_matrix[4], u0 = c.cam_matrix[2], v0 = c.cam_matrix[5];
    const double k1 = c.dist_coeffs[0], k2 = c.dist_coeffs[1], p1 = c.dist_coeffs[2], p2 = c.dist_coeffs[3],
                 k3 = c.dist_coeffs[4];
    int cached_origin = -1;
    Mat3 ir{};
    for (int row = r0; row < r1; ++row) {
        const int origin = row - row % stripe, i = row - origin;
        if (origin != cached_origin) {
            Mat3 Ar;
            std::memcpy(Ar.m, c.cam_matrix, sizeof Ar.m);
            Ar.m[5] = c.cam_matrix[5] - origin;
            invert3(Ar, ir);
            cached_origin = origin;
        }
        double xn = i * ir.m[1] + ir.m[2], yn = i * ir.m[4] + ir.m[5], wn = i * ir.m[7] + ir.m[8];
        for (int j = 0; j < w; ++j) {
            const double iw = 1. / wn, x = xn * iw, y = yn * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
            const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
            const double xd = x * kr + p1 * xy2 + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * xy2;
            const double u = fx * xd + u0, v = fy * yd + v0;
            store_fixed(t, (size_t)(row - r0) * w + j, cv_round(u * kInterTab), cv_round(v * kInterTab), false);
            xn += ir.m[0];
            yn += ir.m[3];
            wn += ir.m[6];
        }
    }
}

// 8-bit RGB2Lab integer path tables (gamma_shift 3, lab_shift 12, lab_shift2 15)
void build_lab_tables(uint16_t gamma_tab[256], uint16_t cbrt_tab[3072], int32_t coeffs[9]) {
    auto to_u16 = [](float v) {
        int r = cv_round((double)v);
        return (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
    };
    for (int i = 0; i < 256; ++i) {
        const float x = i * (1.f / 255.f);
        const float lin = x <= 0.04045f ? x * (1.f / 12.92f) : (float)std::pow((double)(x + 0.055) * (1. / 1.055), 2.4);
        gamma_tab[i] = to_u16(255.f * (1 << 3) * lin);
    }
    for (int i = 0; i < 3072; ++i) {
        const float x = i * (1.f / (255.f * (1 << 3)));
        const float f = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : (float)std::cbrt((double)x);
        cbrt_tab[i] = to_u16((1 << 15) * f);
    }
    const float rgb2xyz[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                              0.072169f, 0.019334f, 0.119193f, 0.950227f};
    const float white[3] = {0.950456f, 1.f, 1.088754f};
    for (int r = 0; r < 3; ++r) {
        const float scale = r == 1 ? (float)(1 << 12) : (1 << 12) / white[r];
        for (int col = 0; col < 3; ++col) coeffs[r * 3 + col] = cv_round((double)(rgb2xyz[r * 3 + col] * scale));
    }
}

// getStructuringElement(MORPH_ELLIPSE, (k,k)): row i spans columns [c-dx, c+dx]
int ellipse_halfwidths(int k, int* dx) {
    const int r = k / 2;
    const double inv_r2 = r ? 1. / ((double)r * r) : 0.;
    int taps = 0;
    for (int i = 0; i < k; ++i) {
        const int dy = i - r;
        int d = cv_round(r * std::sqrt((r * r - dy * dy) * inv_r2));
        dx[i] = d;
        const int j1 = r - d > 0 ? r - d : 0, j2 = r + d + 1 < k ? r + d + 1 : k;
        taps += j2 - j1;
    }
    return taps;
}

}  // namespace lt
