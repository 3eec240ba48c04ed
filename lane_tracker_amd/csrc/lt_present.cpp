// Presentation stage of liblane_tracker_amd.so (lane_tracker.py:629-793): lane overlay, text lines,
// annotated frames on their way back to the host.  See lt_ctx.h.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_ext.h>

#include "lt_ctx.h"

using namespace lt;

// ---- presentation stage: draw_lane() overlay and the bird's-eye image ----------------
namespace {

// Row intervals of cv2.fillPoly's result for a polygon whose two chains are functions of y: the
// union of the 8-connected edge lines and the even-odd interior is, per row, the hull of the edge
// pixels on that row.  The walk is OpenCV's LineIterator (left end point first, error term
// dx - 2 dy, one major-axis step per pixel).
static inline void span_point(int16_t* spans, int bh, int x, int y) {
    if (y >= 0 && y < bh) {
        const int16_t xc = (int16_t)std::min(std::max(x, -32768), 32767);
        if (xc < spans[2 * y]) spans[2 * y] = xc;
        if (xc > spans[2 * y + 1]) spans[2 * y + 1] = xc;
    }
}

static void span_line(int16_t* spans, int bh, int xa, int ya, int xb, int yb) {
    if (std::abs(xb - xa) <= 1 && std::abs(yb - ya) <= 1) {
        // neighbouring pixels (nearly every edge of a lane polygon: one plot point per row): the line is its two end points
        span_point(spans, bh, xa, ya);
        span_point(spans, bh, xb, yb);
        return;
    }
    if (xb < xa) { std::swap(xa, xb); std::swap(ya, yb); }
    const int adx = xb - xa, ady = std::abs(yb - ya), ystep = yb < ya ? -1 : 1;
    const bool tall = ady > adx;
    const int len = tall ? ady : adx, across = tall ? adx : ady;
    int err = len - 2 * across;
    for (int i = 0, x = xa, y = ya; i <= len; ++i) {
        if (y >= 0 && y < bh) {
            const int16_t xc = (int16_t)std::min(std::max(x, -32768), 32767);
            if (xc < spans[2 * y]) spans[2 * y] = xc;
            if (xc > spans[2 * y + 1]) spans[2 * y + 1] = xc;
        }
        const bool turn = err < 0;
        err -= 2 * across;
        if (turn) err += 2 * len;
        if (tall) { y += ystep; x += turn ? 1 : 0; }
        else { x += 1; y += turn ? ystep : 0; }
    }
}

static void lane_polygon_spans(int16_t* spans, int bh, const int32_t* lyx, int nl, const int32_t* ryx, int nr) {
    for (int y = 0; y < bh; ++y) { spans[2 * y] = 32767; spans[2 * y + 1] = -32768; }
    const int np = nl + nr;
    if (np <= 0) return;
    // vertex k of the closed polygon: the left points in order, then the right points reversed (np.flipud)
    const int32_t* last = nr ? ryx : lyx + 2 * (nl - 1);          // vertex np - 1: the first right point, or the last left one
    int px = last[1], py = last[0];
    // every vertex is the end point of the edge before it: an edge between neighbouring pixels (nearly all of them: one plot
    // point per row) adds nothing but its own end point -- its start is on record already (the first edge's start is the
    // polygon's last vertex, which the last edge ends on)
    auto edge = [&](int x, int y) {
        if (std::abs(x - px) <= 1 && std::abs(y - py) <= 1) span_point(spans, bh, x, y);
        else span_line(spans, bh, px, py, x, y);
        px = x;
        py = y;
    };
    for (int k = 0; k < nl; ++k) edge(lyx[2 * k + 1], lyx[2 * k]);
    for (int k = nr - 1; k >= 0; --k) edge(ryx[2 * k + 1], ryx[2 * k]);
}

}  // namespace

extern "C" {

int lt_lane_polygon_spans(int warp_h, const int32_t* left_yx, int n_left, const int32_t* right_yx, int n_right,
                          int16_t* spans) {
    if (warp_h < 1 || n_left < 0 || n_right < 0 || !spans || (n_left && !left_yx) || (n_right && !right_yx))
        return fail(LT_ERR_INVALID, "bad polygon arguments");
    lane_polygon_spans(spans, warp_h, left_yx, n_left, right_yx, n_right);
    return LT_OK;
}

int lt_overlay_configure(lt_ctx* c, const double* Minv) {
    if (!c || !Minv) return fail(LT_ERR_INVALID, "null argument");
    int rc = set_device(c);
    if (rc) return rc;
    TraceScope ts_all("lt_overlay_configure");
    if ((rc = sync_all(c))) return rc;
    // cv2.warpPerspective(lane, Minv, (img_w, img_h)): the same table builder with M := Minv and the
    // camera frame as the destination
    lt_calib u = c->calib;
    std::memcpy(u.M, Minv, sizeof u.M);
    u.warp_w = c->calib.img_w;
    u.warp_h = c->calib.img_h;
    RemapTable t;
    build_warp_table(u, t);
    dev_free(c->d_oxy);
    dev_free(c->d_ofrac);
    c->have_overlay = false;
    if ((rc = dev_alloc(&c->d_oxy, t.xy.size()))) return rc;
    if ((rc = dev_alloc(&c->d_ofrac, t.frac.size()))) return rc;
    HIP_TRY(hipMemcpy(c->d_oxy, t.xy.data(), t.xy.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_ofrac, t.frac.data(), t.frac.size() * 2, hipMemcpyHostToDevice));
    // Camera rows the lane can reach at all: a pixel's four taps are (sx, sy) .. (sx + 1, sy + 1), so only pixels with
    // -1 <= sx <= bw - 1 and -1 <= sy <= bh - 1 can see the bird's-eye image; every other pixel of the annotated frame is the
    // camera pixel whatever the polygon (lt_overlay_rows, lt_present_frame).
    c->ov_r0 = c->ov_r1 = 0;
    for (int y = 0; y < t.rows; ++y) {
        bool any = false;
        for (int x = 0; x < t.cols && !any; ++x) {
            const int sx = t.xy[2 * ((size_t)y * t.cols + x)], sy = t.xy[2 * ((size_t)y * t.cols + x) + 1];
            any = sx >= -1 && sx <= c->calib.warp_w - 1 && sy >= -1 && sy <= c->calib.warp_h - 1;
        }
        if (any) {
            if (c->ov_r1 == 0) c->ov_r0 = y;
            c->ov_r1 = y + 1;
        }
    }
    // The rows the mask chain uploads (lt_upload_frame_rows: what the undistortion reads) and the rows the lane can reach are
    // nearly the same run (457-695 and 458-696 of 720 with the reference calibration): a row or two more in the former, and an
    // annotated frame needs no second upload for the lane's run (lt_upload_frame_rest_rows: 12 us per frame of process()).
    // (a 1920x1080 camera scaled from the reference calibration: 685-1042 against a lane run that ends a dozen rows lower -- the
    // second upload was 26-29 us of every process() frame there: tools/process_trace.py x; up to 32 rows either side are taken along)
    if (c->cam_r1 > c->cam_r0 && c->ov_r1 > c->ov_r0 && c->ov_r0 >= c->cam_r0 - 32 && c->ov_r1 <= c->cam_r1 + 32) {
        c->cam_r0 = std::min(c->cam_r0, c->ov_r0);
        c->cam_r1 = std::max(c->cam_r1, c->ov_r1);
    }
    c->have_overlay = true;
    return LT_OK;
}

// A call is about to overwrite the page-locked staging regions of slots [first, first + n): if a copy out of those regions
// may still be in flight (an earlier call of the same kind over the same slots), wait for it; then widen the busy range.
static int staging_claim(lt_ctx::StagingBusy& b, int first, int n) {
    if (b.hi > b.lo && first < b.hi && first + n > b.lo && b.done) {
        HIP_TRY(hipEventSynchronize(b.done));
        b.lo = b.hi = 0;
    }
    if (b.hi <= b.lo) { b.lo = first; b.hi = first + n; }
    else { b.lo = std::min(b.lo, first); b.hi = std::max(b.hi, first + n); }
    return LT_OK;
}
static int staging_mark(lt_ctx::StagingBusy& b, hipStream_t st) {
    if (!b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(b.done, st));
    return LT_OK;
}

static int present_stream(lt_ctx* c) {
    if (!c->present && create_compute_stream(&c->present, c->search_cus) != hipSuccess) return fail(LT_ERR_HIP, "hipStreamCreate failed");
    return LT_OK;
}
static int ensure_strips(lt_ctx* c);
static int ensure_span_staging(lt_ctx* c) {
    const int bh = c->calib.warp_h;
    if (c->h_spans_cap >= c->capacity) return LT_OK;
    int r = sync_all(c);
    if (r) return r;
    if (c->h_spans) (void)hipHostFree(c->h_spans);
    c->h_spans = nullptr;
    c->h_spans_cap = 0;
    TraceScope ts_("overlay:hipHostMalloc(spans)", (size_t)c->capacity * bh * 2 * sizeof(int16_t));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_spans), (size_t)c->capacity * bh * 2 * sizeof(int16_t), hipHostMallocDefault));
    c->h_spans_cap = c->capacity;
    return LT_OK;
}
}  // extern "C"
namespace lt {
// everything the presentation stage of a window allocates lazily, ahead of the window (lt_warm)
int warm_presentation(lt_ctx* c, bool strips) {
    if (!c->have_overlay) return LT_OK;
    int rc = present_stream(c);
    if (rc) return rc;
    const int bh = c->calib.warp_h;
    if (!c->d_spans && (rc = dev_alloc(&c->d_spans, (size_t)c->capacity * bh * 2))) return rc;
    if ((rc = ensure_span_staging(c))) return rc;
    if (strips) {
        if ((rc = ensure_strips(c))) return rc;
    } else if (!c->d_annot && (rc = dev_alloc(&c->d_annot, (size_t)c->capacity * c->frame_bytes))) return rc;
    if (!c->dl) HIP_TRY(stream_get(&c->dl, SK_PRIORITY, 0));
    return LT_OK;
}
}  // namespace lt
extern "C" {

int lt_poly_points(int warp_w, int warp_h, const double* coeffs, int n, const double* ploty, const double* ploty2, int n_rows,
                   int32_t* left_n, int32_t* right_n, int32_t* left_yx, int32_t* right_yx) {
    if (warp_w < 1 || warp_h < 1 || n < 0 || n_rows < 0 || !coeffs || !left_n || !right_n || !left_yx || !right_yx ||
        (n_rows && (!ploty || !ploty2)))
        return fail(LT_ERR_INVALID, "bad arguments");
    // get_poly_points (lane_tracker.py:511-528) for n pairs of parabolas: fitx = a * ploty**2 + b * ploty + c evaluated as NumPy
    // does (two products, two sums, no contraction: this file is built with -ffp-contract=off), the points with
    // 0 <= fitx <= W - 1 kept, x truncated (astype(int)), and -- as upstream -- y = H - count .. H - 1 whatever rows they were
    const double xmax = (double)(warp_w - 1);
    size_t ol = 0, orr = 0;
    for (int i = 0; i < n; ++i) {
        for (int side = 0; side < 2; ++side) {
            const double a = coeffs[6 * i + 3 * side], b = coeffs[6 * i + 3 * side + 1], cc = coeffs[6 * i + 3 * side + 2];
            int32_t* out = side ? right_yx + 2 * orr : left_yx + 2 * ol;
            int cnt = 0;
            for (int r = 0; r < n_rows; ++r) {
                const double t1 = a * ploty2[r], t2 = b * ploty[r];
                const double x = (t1 + t2) + cc;
                if (x <= xmax && x >= 0.0) out[2 * cnt++ + 1] = (int32_t)(long long)x;
            }
            for (int k = 0; k < cnt; ++k) out[2 * k] = warp_h - cnt + k;
            if (side) { right_n[i] = cnt; orr += (size_t)cnt; }
            else { left_n[i] = cnt; ol += (size_t)cnt; }
        }
    }
    return LT_OK;
}

// The host's arithmetic between a frame's record and its text, for the frame LaneTracker.process() sees nearly always: a first try
// whose fit is valid.  One call instead of five Python functions (fit_poly, _lane_ahead, check_validity, get_curve_radius,
// get_eccentricity: 29 us of a 1280x720 frame's 56 us between the record and the return, tools/process_points.py) -- the same IEEE
// operations in the same order, on the same libm (`pow` is the function CPython's float ** calls), and a flag instead of an answer
// wherever the Python path does anything this function does not reproduce (the caller then takes that path; nothing was changed):
//   in[0..2], in[3..5]   this frame's left / right fit (a, b, c)
//   in[6..11], in[12]    the sum of the valid fits still in the averaging window (left, right; lt_present_lane_from_fit_async's
//                        prev_sum) and the number of fits the average divides by, this frame's included
//   in[13..19]           check_validity's limits: min / max distance at y1, y2, y3, tangent threshold (lane_tracker.py:588-593, :617)
//   in[20], in[21]       metres per pixel, vertical and horizontal
//   ploty_v / ploty2_v   the plot rows of partial = 1 (check_validity counts the points inside the image, :565-569);
//   ploty / ploty2       those of the frame's `partial`: the averaged curves' points go to left_yx / right_yx (get_poly_points)
//   avg6                 the averaged coefficients (np.average over the window, :1186-1187)
//   out[0]               1 valid, 0 invalid (check_validity); out[1] = 1: not reproduced here (no plot point inside the image, a
//                        radius that is not finite / is within 1e-8 of an integer -- upstream's refit of the pixels decides its
//                        int() -- / is beyond 2e15), out[2], out[3] the left / right radius (int() of it, as a double),
//                        out[4] the eccentricity in metres (:551-559)
int lt_frame_tail(int warp_w, int warp_h, const double* in, const double* ploty_v, const double* ploty2_v, int n_rows_v, const double* ploty,
                  const double* ploty2, int n_rows, double* avg6, int32_t* left_n, int32_t* right_n, int32_t* left_yx, int32_t* right_yx,
                  double* out) {
    if (warp_w < 1 || warp_h < 1 || !in || !avg6 || !left_n || !right_n || !left_yx || !right_yx || !out || n_rows < 0 || n_rows_v < 0 ||
        (n_rows && (!ploty || !ploty2)) || (n_rows_v && (!ploty_v || !ploty2_v)))
        return fail(LT_ERR_INVALID, "lt_frame_tail: bad arguments");
    const double* lf = in;
    const double* rf = in + 3;
    out[0] = out[1] = out[2] = out[3] = out[4] = 0.0;
    // ---- check_validity (:561-627) ----
    const double xmax = (double)(warp_w - 1);
    int cnt_v[2];
    for (int side = 0; side < 2; ++side) {
        const double a = in[3 * side], b = in[3 * side + 1], cc = in[3 * side + 2];
        int cnt = 0;
        for (int r = 0; r < n_rows_v; ++r) {
            const double x = (a * ploty2_v[r] + b * ploty_v[r]) + cc;
            cnt += (x <= xmax && x >= 0.0) ? 1 : 0;
        }
        cnt_v[side] = cnt;
    }
    const int nv = std::min(cnt_v[0], cnt_v[1]);
    const long long y1 = warp_w - 1, y2 = warp_w - (long long)((double)nv * 0.35), y3 = warp_w - (long long)((double)nv * 0.75);   // (the WIDTH, as upstream)
    auto at = [](const double* c, long long y) { return (c[0] * (double)(y * y) + c[1] * (double)y) + c[2]; };
    auto slope = [](const double* c, long long y) { return (2.0 * c[0]) * (double)y + c[1]; };
    const double x1d = std::fabs(at(lf, y1) - at(rf, y1)), x2d = std::fabs(at(lf, y2) - at(rf, y2)), x3d = std::fabs(at(lf, y3) - at(rf, y3));
    const double* lim = in + 13;
    bool valid = !((x1d < lim[0]) || (x1d > lim[1]) || (x2d < lim[2]) || (x2d > lim[3]) || (x3d < lim[4]) || (x3d > lim[5]));
    if (x1d != x1d || x2d != x2d || x3d != x3d) { out[1] = 1.0; return LT_OK; }     // (NaN: the Python path's comparisons decide)
    if (valid) {
        const double n1 = std::fabs(slope(lf, y1) - slope(rf, y1)), n2 = std::fabs(slope(lf, y3) - slope(rf, y3));
        if (n1 != n1 || n2 != n2) { out[1] = 1.0; return LT_OK; }
        valid = !((n1 >= lim[6]) || (n2 >= lim[6]));
    }
    out[0] = valid ? 1.0 : 0.0;
    if (!valid) return LT_OK;
    // ---- the running average with this fit last (:1186-1187: np.average over the window = sequential sum, one division) ----
    const double count = in[12];
    if (!(count >= 1.0)) { out[1] = 1.0; return LT_OK; }
    for (int k = 0; k < 6; ++k) avg6[k] = (count > 1.0 ? in[6 + k] + in[k] : in[k]) / count;
    // ---- get_poly_points of the averaged curves (:511-528) ----
    int rc = lt_poly_points(warp_w, warp_h, avg6, 1, ploty, ploty2, n_rows, left_n, right_n, left_yx, right_yx);
    if (rc) return rc;
    if (left_n[0] < 1 || right_n[0] < 1) { out[1] = 1.0; return LT_OK; }
    // ---- get_curve_radius (:530-549), through the pixel fit's coefficients in metric units ----
    // (libm's pow through a pointer the compiler cannot see through: `x ** 2` in Python is pow(x, 2.0) at run time, and a compiler
    // that rewrites the call as x * x -- always allowed, always done -- may differ from it in the last bit)
    static double (*volatile const libm_pow)(double, double) = ::pow;
    const double mppv = in[20], mpph = in[21], y_eval = (double)warp_h;
    for (int side = 0; side < 2; ++side) {
        const double* c = in + 3 * side;
        const double a = c[0] * mpph / libm_pow(mppv, 2.0), b = c[1] * mpph / mppv;
        if (a == 0.0) { out[1] = 1.0; return LT_OK; }
        const double t = ((2.0 * a) * y_eval) * mppv + b;
        const double val = libm_pow(1.0 + libm_pow(t, 2.0), 1.5) / std::fabs(2.0 * a);
        if (!std::isfinite(val) || std::fabs(val) >= 2e15 || std::fabs(val - std::nearbyint(val)) <= 1e-8 * std::max(1.0, std::fabs(val))) {
            out[1] = 1.0;
            return LT_OK;
        }
        out[2 + side] = std::trunc(val);
    }
    // ---- get_eccentricity (:551-559): the last plot point of each averaged curve ----
    const long long mid = (long long)((double)warp_w / 2.0);
    const long long xl = left_yx[2 * (left_n[0] - 1) + 1], xr = right_yx[2 * (right_n[0] - 1) + 1];
    out[4] = ((double)((mid - xl) - (xr - mid)) / 2.0) * mpph;
    return LT_OK;
}

// lt_overlay_run; rows4: two runs of camera rows {a0, a1, b0, b1} outside which the annotated frames are not needed
// (lt_present_frame, lt_overlay_run_rows), nullptr = all of them
// strip mode: the rows the lane can reach (lt_overlay_rows) of every slot, packed, into the context's strip buffer
static int ensure_strips(lt_ctx* c) {
    const size_t sb = (size_t)std::max(c->ov_r1 - c->ov_r0, 0) * c->calib.img_w * 3;
    if (c->d_strip && c->strip_bytes == sb) return LT_OK;
    int rc = sync_all(c);
    if (rc) return rc;
    dev_free(c->d_strip);
    c->strip_bytes = sb;
    return dev_alloc(&c->d_strip, (size_t)c->capacity * std::max<size_t>(sb, 4));
}

// direct_out (one frame, row runs): the device-visible address of a page-locked frame in HOST memory the kernel stores the drawn
// rows into itself, instead of the context's annotated-frame buffer and a copy kernel behind it (lt_present_lane_async)
// the plot rows (ploty, ploty ** 2 of get_poly_points) on the device: sent when they change (they depend on `partial` and the image
// height only); waits for the whole context when they do
static int ensure_ploty(lt_ctx* c, const double* ploty, const double* ploty2, int n_rows) {
    if ((int)c->h_ploty.size() == 2 * n_rows && std::memcmp(c->h_ploty.data(), ploty, (size_t)n_rows * sizeof(double)) == 0 &&
        std::memcmp(c->h_ploty.data() + n_rows, ploty2, (size_t)n_rows * sizeof(double)) == 0)
        return LT_OK;
    int rc = sync_all(c);
    if (rc) return rc;
    dev_free(c->d_ploty);
    c->d_ploty = nullptr;
    c->h_ploty.assign(ploty, ploty + n_rows);
    c->h_ploty.insert(c->h_ploty.end(), ploty2, ploty2 + n_rows);
    uint8_t* raw = nullptr;
    if ((rc = dev_alloc(&raw, (size_t)2 * n_rows * sizeof(double)))) { c->h_ploty.clear(); return rc; }
    c->d_ploty = reinterpret_cast<double*>(raw);
    HIP_TRY(hipMemcpyAsync(c->d_ploty, c->h_ploty.data(), (size_t)2 * n_rows * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

// averaged coefficients instead of points (lt_overlay_run_strip_coeffs): the device forms plot points and row intervals itself
struct CoeffInput { const double* coeffs; const uint8_t* draw; const double* ploty; const double* ploty2; int n_rows; };

static int overlay_run_impl(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                            const int32_t* right_yx, double alpha, const int* rows4, bool strip = false, uint8_t* direct_out = nullptr,
                            bool* went_direct = nullptr, const CoeffInput* ci = nullptr) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_overlay_run before lt_overlay_configure");
    if (n == 0) return LT_OK;
    if (!ci) {
        if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
        long long tl = 0, tr = 0;
        for (int i = 0; i < n; ++i) {
            if (left_n[i] < 0 || right_n[i] < 0) return fail(LT_ERR_INVALID, "negative point count");
            tl += left_n[i];
            tr += right_n[i];
        }
        if ((tl && !left_yx) || (tr && !right_yx)) return fail(LT_ERR_INVALID, "null point list");
    }
    if (!rows4 && !strip) {
        const int bad = first_partial(c->frame_full, first, n);
        if (bad >= 0)
            return fail(LT_ERR_STATE, "slot %d holds only part of its camera frame (lt_upload_frame_rows without lt_upload_frame_rest): a whole-frame "
                                      "overlay would show rows of the block's previous occupant", bad);
    }
    if ((rc = set_device(c))) return rc;
    const int bh = c->calib.warp_h;
    if (!c->d_spans && (rc = dev_alloc(&c->d_spans, (size_t)c->capacity * bh * 2))) return rc;
    if (strip) {
        if ((rc = ensure_strips(c))) return rc;
    } else if (!c->d_annot && (rc = dev_alloc(&c->d_annot, (size_t)c->capacity * c->frame_bytes))) return rc;
    // One frame (process()): the intervals travel as a kernel argument -- no staging buffer, no copy launch, no events
    static const bool arg_ok = [] { const char* e = LT_EXP_ENV("LT_SPANS_ARG"); return !(e && e[0] == '0'); }();
    bool one = arg_ok && !strip && !ci && n == 1 && bh <= LT_SPAN_ARG_ROWS && (c->calib.img_w & 3) == 0;
    if (ci && (rc = ensure_ploty(c, ci->ploty, ci->ploty2, ci->n_rows))) return rc;
    int16_t one_spans[2 * LT_SPAN_ARG_ROWS];
    auto claim_staging = [&]() -> int {
        int r = staging_claim(c->spans_busy, first, n);
        if (r) return r;
        if ((r = ensure_span_staging(c))) return r;
        return (int)LT_OK;
    };
    if (!one && (rc = claim_staging())) return rc;
    int16_t* hs = one ? one_spans : c->h_spans + (size_t)first * bh * 2;
    static const bool timing = LT_EXP_ENV("LT_OVERLAY_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // ~18 us of edge walking per polygon: a window's piece of 32 .. 128 polygons is shared among a few threads (the caller is
    // the one thread that feeds the device)
    const int workers = std::max(1, std::min({n / 8, 8, (int)std::thread::hardware_concurrency()}));
    auto some = [&](int w) {
        size_t ol = 0, orr = 0;
        for (int i = 0; i < n; ++i) {
            if (i * (long long)workers / n == w)
                lane_polygon_spans(hs + (size_t)i * bh * 2, bh, left_yx ? left_yx + 2 * ol : nullptr, left_n[i],
                                   right_yx ? right_yx + 2 * orr : nullptr, right_n[i]);
            ol += (size_t)left_n[i];
            orr += (size_t)right_n[i];
        }
    };
    if (ci) {                                  // six doubles and a draw byte at the start of every slot's interval region
        for (int i = 0; i < n; ++i) {
            uint8_t* reg = reinterpret_cast<uint8_t*>(hs + (size_t)i * bh * 2);
            std::memcpy(reg, ci->coeffs + (size_t)6 * i, 6 * sizeof(double));
            reg[48] = ci->draw ? ci->draw[i] : 1;
        }
    } else if (workers == 1) some(0);
    else {
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; ++w) pool.emplace_back(some, w);
        some(0);
        for (auto& t : pool) t.join();
    }
    // the rows of the frame the path does not read came on the copy stream (lt_upload_frame_rest): the overlay is their reader
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = present_stream(c))) return rc;
    hipStream_t ps = c->present;
    if (c->rest_pending) {
        bool precise = true;
        if ((rc = wait_range(c->rests, ps, first, first + n, &precise))) return rc;
        if (!precise) HIP_TRY(hipStreamWaitEvent(ps, c->rest_done, 0));
    }
    {   // the camera rows of these slots: behind the launches that wrote their masks (which waited for the rows' upload)
        bool precise = true;
        if ((rc = wait_range(c->writers, ps, first, first + n, &precise))) return rc;
        if (!precise) {
            rc = for_each_slice(c, first, n, [&](hipStream_t st, int, int) {
                hipEvent_t e = next_order_event(c);
                if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(ps, e, 0));
                return (int)LT_OK;
            });
            if (rc) return rc;
        }
    }
    // an asynchronous download may still be reading the annotated frames this call overwrites
    if (c->annot_busy.hi > c->annot_busy.lo && first < c->annot_busy.hi && first + n > c->annot_busy.lo && c->annot_busy.done)
        HIP_TRY(hipStreamWaitEvent(ps, c->annot_busy.done, 0));
    if (one) {
        uint8_t* dst = direct_out ? direct_out : c->d_annot + (size_t)first * c->frame_bytes;
        if (launch_overlay_lane_one(ps, c->d_frames + (size_t)first * c->frame_bytes, dst,
                                    c->d_oxy, c->d_ofrac, hs, c->calib.img_h, c->calib.img_w, bh, c->calib.warp_w, (float)alpha, rows4)) {
            HIP_TRY(hipGetLastError());
            if (went_direct) *went_direct = direct_out != nullptr;
            if (!direct_out) mark_annot(c, first, n, rows4 ? 0 : 1);
            return note_range(c->readers, ps, first, first + n);
        }
        // not launched (the runtime refused the argument block): the staged way after all, with the intervals already built
        one = false;
        if ((rc = claim_staging())) return rc;
        std::memcpy(c->h_spans + (size_t)first * bh * 2, one_spans, (size_t)bh * 2 * sizeof(int16_t));
        hs = c->h_spans + (size_t)first * bh * 2;
    }
    launch_copy_from_pinned(ps, c->d_spans + (size_t)first * bh * 2, hs, (size_t)n * bh * 2 * sizeof(int16_t));
    if (ci && !launch_lane_spans_from_regions(ps, c->d_ploty, c->d_ploty + ci->n_rows, ci->n_rows, bh, c->calib.warp_w, c->d_spans + (size_t)first * bh * 2, n))
        return fail(LT_ERR_STATE, "lt_overlay_run_strip_coeffs: not available for this bird's-eye height");
    const auto t2 = std::chrono::steady_clock::now();
    if (strip) {
        if (!launch_overlay_lane_strip(ps, c->d_frames + (size_t)first * c->frame_bytes, c->frame_bytes, c->d_strip + (size_t)first * c->strip_bytes,
                                       c->strip_bytes, c->d_oxy, c->d_ofrac, c->d_spans + (size_t)first * bh * 2, (size_t)bh, c->calib.img_w,
                                       c->ov_r0, c->ov_r1, bh, c->calib.warp_w, (float)alpha, n))
            return fail(LT_ERR_STATE, "strip overlay needs a frame width that is a multiple of 4");
    } else
    launch_overlay_lane(ps, c->d_frames + (size_t)first * c->frame_bytes, c->d_annot + (size_t)first * c->frame_bytes,
                        c->frame_bytes, c->d_oxy, c->d_ofrac, c->d_spans + (size_t)first * bh * 2, (size_t)bh,
                        c->calib.img_h, c->calib.img_w, bh, c->calib.warp_w, (float)alpha, n, rows4);
    HIP_TRY(hipGetLastError());
    if (!strip) mark_annot(c, first, n, rows4 ? 0 : 1);
    if ((rc = staging_mark(c->spans_busy, ps))) return rc;
    rc = note_range(c->readers, ps, first, first + n);
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        std::fprintf(stderr, "overlay_run n=%d: spans %ld us, wait+memcpy %ld us, launch+events %ld us\n", n, us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return rc;
}

int lt_overlay_run(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                   const int32_t* right_yx, double alpha) {
    return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, nullptr);
}

static int ordered_rows(lt_ctx* c, const int32_t* rows4, int r[4]) {
    for (int k = 0; k < 4; ++k) r[k] = rows4[k];
    if (!(0 <= r[0] && r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3] && r[3] <= c->calib.img_h))
        return fail(LT_ERR_INVALID, "row runs must be ordered and inside the frame");
    return LT_OK;
}

int lt_overlay_run_rows(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                        const int32_t* right_yx, double alpha, const int32_t* rows4) {
    if (!rows4) return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, nullptr);
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int r[4];
    const int rc = ordered_rows(c, rows4, r);
    if (rc) return rc;
    return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, r);
}

int lt_overlay_run_strip(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                         const int32_t* right_yx, double alpha) {
    return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, nullptr, true);
}

// lt_overlay_run_strip from the lanes' AVERAGED coefficients (n x 6 doubles: left a, b, c, right a, b, c) instead of their plot points:
// get_poly_points and the polygons' row intervals are formed on the device (k_lane_spans_from_fit, one workgroup per frame: the
// host's f64 operations in the host's order), which takes 11 us per frame off the thread that drives an annotated stream.
// draw: n bytes, 0 = no lane in that frame (plain copy); nullptr: all drawn.
int lt_overlay_run_strip_coeffs(lt_ctx* c, int first, int n, const double* coeffs, const uint8_t* draw, const double* ploty, const double* ploty2,
                                int n_rows, double alpha) {
    if (!c || (n > 0 && !coeffs) || !ploty || !ploty2 || n_rows < 1) return fail(LT_ERR_INVALID, "lt_overlay_run_strip_coeffs: bad arguments");
    const int bh = c->calib.warp_h;
    if ((bh & 1) || bh * 4 < 56 || ((size_t)2 * bh + (size_t)2 * n_rows) * sizeof(int) > 60 * 1024)
        return fail(LT_ERR_STATE, "lt_overlay_run_strip_coeffs: not available for this bird's-eye height");
    const CoeffInput ci{coeffs, draw, ploty, ploty2, n_rows};
    return overlay_run_impl(c, first, n, nullptr, nullptr, nullptr, nullptr, alpha, nullptr, true, nullptr, nullptr, &ci);
}

// The strips of slots [first, first + n) on their way into the caller's frames: one contiguous copy per block of STRIP_BLOCK slots
// into page-locked staging (the copy engine at its full rate: 56 GB/s against the 37 GB/s of the kernel that stored row runs into a
// page-locked frame array, and no CUs taken from the mask chain), then -- on the library's copy threads, once the copy's event has
// fired -- from staging into rows [ov_r0, ov_r1) of out + i * out_frame_stride.  `out` is ordinary memory.  Complete when
// lt_host_copy_wait_group(group) returns.
int lt_strip_download_async(lt_ctx* c, int first, int n, uint8_t* out, size_t out_frame_stride, int group) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output frames");
    if (!c->d_strip || !c->present) return fail(LT_ERR_STATE, "lt_strip_download_async before lt_overlay_run_strip");
    if (out_frame_stride < c->frame_bytes) return fail(LT_ERR_INVALID, "frame stride below the frame size");
    if (n == 0 || c->strip_bytes == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    if (!c->dl) HIP_TRY(stream_get(&c->dl, SK_PRIORITY, 0));
    hipEvent_t e = next_order_event(c);
    if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(e, c->present));
    HIP_TRY(hipStreamWaitEvent(c->dl, e, 0));
    constexpr int STRIP_BLOCK = 32;
    const size_t sb = c->strip_bytes, block_bytes = (size_t)STRIP_BLOCK * sb;
    const size_t row_off = (size_t)c->ov_r0 * c->calib.img_w * 3;
    for (int at = 0; at < n; at += STRIP_BLOCK) {
        const int m = std::min(STRIP_BLOCK, n - at);
        if (host_reserve(group)) return fail(LT_ERR_INVALID, "lt_strip_download_async: unknown group %d", group);
        uint8_t* block = static_cast<uint8_t*>(pinned_block_acquire(block_bytes));
        hipEvent_t done = block ? pooled_event() : nullptr;
        if (!block || !done) {
            if (block) pinned_block_release(block, block_bytes);
            host_unreserve(group);
            return fail(LT_ERR_NOMEM, "no page-locked staging block for the strips");
        }
        hipError_t he = hipMemcpyAsync(block, c->d_strip + (size_t)(first + at) * sb, (size_t)m * sb, hipMemcpyDeviceToHost, c->dl);
        if (he == hipSuccess) he = hipEventRecord(done, c->dl);
        if (he != hipSuccess) {
            pinned_block_release(block, block_bytes);
            pooled_event_release(done);
            host_unreserve(group);
            return fail(LT_ERR_HIP, "strip download failed: %s", hipGetErrorString(he));
        }
        uint8_t* dst = out + (size_t)at * out_frame_stride + row_off;
        host_after_event(done, c->device, [=]() {
            std::shared_ptr<void> hold(block, [=](void* p) { pinned_block_release(p, block_bytes); });
            (void)host_submit_copy2d(group, dst, out_frame_stride, block, sb, sb, (size_t)m, hold);
            pooled_event_release(done);
            host_unreserve(group);           // the scatter pieces are queued: the group now waits for those
        });
    }
    if (c->annot_busy.hi <= c->annot_busy.lo) { c->annot_busy.lo = first; c->annot_busy.hi = first + n; }
    else { c->annot_busy.lo = std::min(c->annot_busy.lo, first); c->annot_busy.hi = std::max(c->annot_busy.hi, first + n); }
    return staging_mark(c->annot_busy, c->dl);
}

int lt_overlay_rows(lt_ctx* c, int* row0, int* row1) {
    if (!c || !row0 || !row1) return fail(LT_ERR_INVALID, "null argument");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_overlay_rows before lt_overlay_configure");
    *row0 = c->ov_r0;
    *row1 = c->ov_r1;
    return LT_OK;
}

int lt_overlay_set_font(lt_ctx* c, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs, int glyph_w,
                        int glyph_h) {
    if (!c || !atlas || !advance) return fail(LT_ERR_INVALID, "null argument");
    if (n_glyphs < 1 || n_glyphs > 256 || glyph_w < 1 || glyph_w > 255 || glyph_h < 1 || glyph_h > 255 || first_char < 0)
        return fail(LT_ERR_INVALID, "bad font geometry");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    dev_free(c->d_atlas);
    dev_free(c->d_advance);
    c->font_glyphs = 0;
    const size_t bytes = (size_t)n_glyphs * glyph_w * glyph_h;
    if ((rc = dev_alloc(&c->d_atlas, bytes))) return rc;
    if ((rc = dev_alloc(&c->d_advance, (size_t)n_glyphs))) return rc;
    HIP_TRY(hipMemcpy(c->d_atlas, atlas, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_advance, advance, (size_t)n_glyphs, hipMemcpyHostToDevice));
    c->h_advance.assign(advance, advance + n_glyphs);
    c->font_first = first_char;
    c->font_glyphs = n_glyphs;
    c->font_gw = glyph_w;
    c->font_gh = glyph_h;
    return LT_OK;
}

int lt_overlay_text(lt_ctx* c, int first, int n, const char* lines, int n_lines, int line_len, int x0, int y0, int step) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->font_glyphs) return fail(LT_ERR_STATE, "lt_overlay_text before lt_overlay_set_font");
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_overlay_text before lt_overlay_run");
    if (n == 0 || n_lines <= 0 || line_len <= 0) return LT_OK;
    if (!lines) return fail(LT_ERR_INVALID, "null text");
    if ((rc = set_device(c))) return rc;
    const size_t per = (size_t)n_lines * line_len;
    if (per > c->text_per_slot || c->text_slots < c->capacity) {      // (re)size the per-slot text buffers: rare, synchronises
        if ((rc = sync_all(c))) return rc;
        const size_t per_new = (std::max(per, c->text_per_slot) + 3) & ~(size_t)3, total = per_new * (size_t)c->capacity;
        dev_free(c->d_lines);
        dev_free(c->d_xpos);
        if (c->h_lines) (void)hipHostFree(c->h_lines);
        if (c->h_xpos) (void)hipHostFree(c->h_xpos);
        c->h_lines = nullptr;
        c->h_xpos = nullptr;
        c->text_per_slot = 0;
        c->text_slots = 0;
        if ((rc = dev_alloc(&c->d_lines, total))) return rc;
        if ((rc = dev_alloc(&c->d_xpos, total))) return rc;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_lines), total, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_xpos), total * sizeof(int16_t), hipHostMallocDefault));
        c->text_per_slot = per_new;
        c->text_slots = c->capacity;
    }
    if ((rc = staging_claim(c->text_busy, first, n))) return rc;
    // A slot's lines sit at the slot's own position in the per-slot buffers, at their FIXED stride text_per_slot -- not at
    // this call's n_lines * line_len: staging_claim orders calls by slot range, and two calls in flight over disjoint
    // slots with different line counts (a 'fail' piece has one line, a lane piece two) must not meet in bytes.
    const size_t stride = c->text_per_slot;
    uint8_t* hl = c->h_lines + (size_t)first * stride;
    int16_t* hx = c->h_xpos + (size_t)first * stride;
    for (int i = 0; i < n; ++i) {
        std::memcpy(hl + (size_t)i * stride, lines + (size_t)i * per, per);
        for (int l = 0; l < n_lines; ++l) {                // left edge of every character: running sum of advances
            const char* src = lines + (size_t)i * per + (size_t)l * line_len;
            int16_t* dst = hx + (size_t)i * stride + (size_t)l * line_len;
            int x = x0;
            bool ended = false;
            for (int k = 0; k < line_len; ++k) {
                const unsigned char ch = (unsigned char)src[k];
                ended = ended || ch == 0;
                dst[k] = (int16_t)std::min(x, 32767);
                const int g = (int)ch - c->font_first;
                if (!ended && g >= 0 && g < c->font_glyphs) x += c->h_advance[(size_t)g];
            }
        }
    }
    uint8_t* dl = c->d_lines + (size_t)first * stride;
    int16_t* dx = c->d_xpos + (size_t)first * stride;
    if ((rc = present_stream(c))) return rc;
    // A frame or two (process(), one frame per call): the kernel reads the few hundred bytes from the page-locked buffers
    // themselves -- two launches fewer between the record and the annotated frame.  A window's worth goes to the device first.
    const uint8_t* kl = dl;
    const int16_t* kx = dx;
    void *pl = nullptr, *px = nullptr;
    static const bool direct_ok = [] { const char* e = LT_EXP_ENV("LT_TEXT_DIRECT"); return !(e && e[0] == '0'); }();
    if (direct_ok && n <= 2 && hipHostGetDevicePointer(&pl, hl, 0) == hipSuccess && hipHostGetDevicePointer(&px, hx, 0) == hipSuccess &&
        pl && px) {
        kl = static_cast<const uint8_t*>(pl);
        kx = static_cast<const int16_t*>(px);
    } else {
        (void)hipGetLastError();
        launch_copy_from_pinned(c->present, dl, hl, (size_t)n * stride);
        launch_copy_from_pinned(c->present, dx, hx, (size_t)n * stride * sizeof(int16_t));
    }
    launch_overlay_text(c->present, c->d_annot + (size_t)first * c->frame_bytes, c->frame_bytes, c->calib.img_h, c->calib.img_w,
                        c->d_atlas, c->d_advance, c->font_first, c->font_glyphs, c->font_gw, c->font_gh, kl, kx,
                        n_lines, line_len, (int)stride, y0, step, n);
    HIP_TRY(hipGetLastError());
    return staging_mark(c->text_busy, c->present);
}

int lt_download_overlay(lt_ctx* c, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_download_overlay before lt_overlay_run");
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    {
        const int bad = first_partial(c->annot_full, first, n);
        if (bad >= 0) return fail(LT_ERR_STATE, "slot %d holds row runs of its annotated frame only (lt_overlay_run_rows / lt_present_*): no whole frame to download", bad);
    }
    const uint8_t* src = c->d_annot + (size_t)first * c->frame_bytes;
    const size_t bytes = (size_t)n * c->frame_bytes;
    if (!c->present || n == 0) return download(c, src, out, bytes);
    // The annotated frames are written on the presentation stream and nowhere else (lt_overlay_run, lt_overlay_text), behind
    // everything they depend on: the copy is enqueued there, behind them, and the host waits once -- not once for the overlay
    // and once more for a copy it issues only then (10 us of process()'s 0.4 ms per frame).
    if ((rc = set_device(c))) return rc;
    // a frame or two: by a copy kernel (no engine start-up: 11 us less per frame of process()); LT_DL1_KERNEL=0: the engine
    static const bool by_kernel = [] { const char* e = LT_EXP_ENV("LT_DL1_KERNEL"); return !(e && e[0] == '0'); }();
    if (!(by_kernel && n <= 2 && launch_copy_to_pinned(c->present, out, src, bytes)))
        HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->present));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

// process()'s tail for ONE frame in one call: lt_overlay_run + lt_overlay_text + the way back, the host waiting once.  With
// rows4 = {a0, a1, b0, b1} only those two runs of camera rows are drawn and written to `out` (the rows at their places in the
// frame): a pixel outside the rows the lane can reach (lt_overlay_rows) and outside the text lines is the camera pixel, which
// the caller has -- process() copies those rows from its input while the device is busy, and only half the frame crosses the
// bus behind the overlay.  The runs must cover the text lines and, for a non-empty polygon, lt_overlay_rows.
static int present_copy_rows(lt_ctx* c, int slot, uint8_t* out, int row0, int row1) {
    if (row1 <= row0) return LT_OK;
    const size_t row_bytes = (size_t)c->calib.img_w * 3, off = (size_t)row0 * row_bytes, bytes = (size_t)(row1 - row0) * row_bytes;
    const uint8_t* src = c->d_annot + (size_t)slot * c->frame_bytes;
    if (!launch_copy_to_pinned(c->present, out + off, src + off, bytes))
        HIP_TRY(hipMemcpyAsync(out + off, src + off, bytes, hipMemcpyDeviceToHost, c->present));
    HIP_TRY(hipGetLastError());
    return LT_OK;
}
// rows4 -> r[4] (nullptr: the whole frame as the first run); `split`: the text lines must lie in the first run and the rows the
// lane can reach in the second, the two apart -- the condition for drawing and sending the second run before the text exists
static int present_rows(lt_ctx* c, const int32_t* rows4, bool text, int n_lines, int y0, int step, bool lane, bool split, int r[4]) {
    const int H = c->calib.img_h;
    r[0] = 0; r[1] = H; r[2] = H; r[3] = H;
    if (!rows4) return split ? fail(LT_ERR_INVALID, "two row runs are needed") : (int)LT_OK;
    for (int k = 0; k < 4; ++k) r[k] = rows4[k];
    if (!(0 <= r[0] && r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3] && r[3] <= H))
        return fail(LT_ERR_INVALID, "row runs must be ordered and inside the frame");
    auto within = [&](int lo, int hi, int a, int b) { lo = std::max(lo, 0); hi = std::min(hi, H); return lo >= hi || (a <= lo && hi <= b); };
    const int t0 = y0, t1 = y0 + (n_lines - 1) * step + c->font_gh;
    if (split) {
        if (text && !within(t0, t1, r[0], r[1])) return fail(LT_ERR_INVALID, "the first row run does not cover the text lines");
        if (!within(c->ov_r0, c->ov_r1, r[2], r[3])) return fail(LT_ERR_INVALID, "the second row run does not cover the rows the lane can reach (lt_overlay_rows)");
        return LT_OK;
    }
    auto covered = [&](int lo, int hi) { return within(lo, hi, r[0], r[1]) || within(lo, hi, r[2], r[3]) || (r[1] == r[2] && within(lo, hi, r[0], r[3])); };
    if (text && !covered(t0, t1)) return fail(LT_ERR_INVALID, "the row runs do not cover the text lines");
    if (lane && !covered(c->ov_r0, c->ov_r1)) return fail(LT_ERR_INVALID, "the row runs do not cover the rows the lane can reach (lt_overlay_rows)");
    return LT_OK;
}

int lt_present_frame(lt_ctx* c, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx, const int32_t* right_yx,
                     double alpha, const char* lines, int n_lines, int line_len, int x0, int y0, int step, uint8_t* out,
                     const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_present_frame before lt_overlay_configure");
    const bool text = lines && n_lines > 0 && line_len > 0 && c->font_glyphs > 0;
    int r[4];
    if ((rc = present_rows(c, rows4, text, n_lines, y0, step, left_n[0] > 0 || right_n[0] > 0, false, r))) return rc;
    if ((rc = overlay_run_impl(c, slot, 1, left_n, right_n, left_yx, right_yx, alpha, rows4 ? r : nullptr))) return rc;
    if (text && (rc = lt_overlay_text(c, slot, 1, lines, n_lines, line_len, x0, y0, step))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[0], r[1]))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[2], r[3]))) return rc;
    HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

// lt_present_frame in two halves, for a caller that knows the polygon before it knows the text (LaneTracker.process(): the
// averaged curves follow from the record at once, radius, eccentricity and the verdict on the frame take the host another
// 25 us): the first half draws both row runs and sends the second one -- the rows the lane can reach -- on its way without
// waiting; the second half blends the text into the first run, sends that and waits for both.  A first half whose frame turns
// out invalid is simply followed by a whole lt_present_frame (same slot, same `out`): it draws and sends everything again.
int lt_present_lane_async(lt_ctx* c, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                          const int32_t* right_yx, double alpha, uint8_t* out, const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_present_lane_async before lt_overlay_configure");
    int r[4];
    if ((rc = present_rows(c, rows4, false, 0, 0, 0, true, true, r))) return rc;
    // The lane's rows straight into `out`: when the first run is empty (the text is the host's business, lt_text_blend_host) the
    // overlay kernel stores what it draws into the page-locked frame itself -- one launch instead of two, the rows cross the bus as
    // they are drawn (-13 us per frame of process(); LT_OVERLAY_DIRECT=0: draw into the context's buffer, then the copy kernel).
    static const bool direct_ok = [] { const char* e = LT_EXP_ENV("LT_OVERLAY_DIRECT"); return !(e && e[0] == '0'); }();
    void* dev = nullptr;
    bool direct = false;
    if (direct_ok && r[1] <= r[0] && ((uintptr_t)out & 15) == 0 && hipHostGetDevicePointer(&dev, out, 0) == hipSuccess && dev) {
        if ((rc = overlay_run_impl(c, slot, 1, left_n, right_n, left_yx, right_yx, alpha, r, false, static_cast<uint8_t*>(dev), &direct))) return rc;
        if (direct) return LT_OK;
        return present_copy_rows(c, slot, out, r[2], r[3]);      // (the staged path ran: the rows are in the context's buffer)
    }
    (void)hipGetLastError();
    if ((rc = overlay_run_impl(c, slot, 1, left_n, right_n, left_yx, right_yx, alpha, r))) return rc;
    return present_copy_rows(c, slot, out, r[2], r[3]);
}

// The lane of a frame drawn WITHOUT the host: enqueued on the slot's own stream right behind its search (call it after
// lt_band_fit_run / lt_sws_fit_run over that one slot).  One workgroup forms the running average with the fit of the slot's
// record, the averaged curves' plot points and the polygon's row intervals (k_lane_spans_from_fit: the host's operations, in the
// host's order), the overlay kernel behind it stores the rows the lane can reach into the page-locked frame `out` -- 40 us before
// the host, which has to see the record first, could have launched it.  A record without a usable fit draws nothing (the rows
// come back as the camera's); a frame the host then finds invalid is drawn again by whatever it presents (lt_present_frame /
// lt_present_lane_async wait for this overlay).  LT_ERR_STATE: not available for these arguments (rows4's first run not empty,
// `out` not page-locked, too many plot rows) -- the caller draws the lane itself once it has the record.
int lt_present_lane_from_fit_async(lt_ctx* c, int slot, const double* prev_sum, int count, const double* ploty, const double* ploty2,
                                   int n_rows, double alpha, uint8_t* out, const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out || !ploty || !ploty2 || n_rows < 1 || count < 1 || (count > 1 && !prev_sum)) return fail(LT_ERR_INVALID, "lt_present_lane_from_fit_async: bad arguments");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_present_lane_from_fit_async before lt_overlay_configure");
    int r[4];
    if ((rc = present_rows(c, rows4, false, 0, 0, 0, true, true, r))) return rc;
    if ((rc = set_device(c))) return rc;
    void* dev = nullptr;
    if (r[1] > r[0] || ((uintptr_t)out & 15) || (c->calib.img_w & 3) || hipHostGetDevicePointer(&dev, out, 0) != hipSuccess || !dev) {
        (void)hipGetLastError();
        return fail(LT_ERR_STATE, "lt_present_lane_from_fit_async: needs the lane's run alone and a page-locked, 16-byte aligned frame");
    }
    const int bh = c->calib.warp_h;
    if (!c->d_spans && (rc = dev_alloc(&c->d_spans, (size_t)c->capacity * bh * 2))) return rc;
    if ((rc = ensure_ploty(c, ploty, ploty2, n_rows))) return rc;
    return for_each_slice(c, slot, 1, [&](hipStream_t st, int f0, int) {
        if (c->rest_pending) {       // rows of the frame the mask chain's upload did not bring (lt_upload_frame_rest): the overlay reads them
            bool precise = true;
            int wrc = wait_range(c->rests, st, f0, f0 + 1, &precise);
            if (wrc) return wrc;
            if (!precise) HIP_TRY(hipStreamWaitEvent(st, c->rest_done, 0));
        }
        int16_t* sp = c->d_spans + (size_t)f0 * bh * 2;
        if (!launch_lane_spans_from_fit(st, c->d_rec + f0, prev_sum, count, c->d_ploty, c->d_ploty + n_rows, n_rows, bh, c->calib.warp_w, sp))
            return fail(LT_ERR_STATE, "lt_present_lane_from_fit_async: too many rows for one workgroup's LDS");
        launch_overlay_lane(st, c->d_frames + (size_t)f0 * c->frame_bytes, static_cast<uint8_t*>(dev), c->frame_bytes, c->d_oxy, c->d_ofrac, sp,
                            (size_t)bh, c->calib.img_h, c->calib.img_w, bh, c->calib.warp_w, (float)alpha, 1, r);
        c->lane_spec_stream = st;
        c->lane_spec_ticket = 0;
        // the completion word the host polls (the record mirror's page-locked block, 256 bytes: the record, its ticket at +64, this one
        // at +128): a one-thread launch behind the overlay.  (Round 6 tried to let the overlay's own last workgroup store it -- a
        // system-scope fence per thread, a count of finished workgroups --: the fences made the overlay 39 us instead of 20 + 4.)
        if (c->h_rec) {
            void* hdev = nullptr;
            if (hipHostGetDevicePointer(&hdev, c->h_rec, 0) == hipSuccess && hdev) {
                unsigned t = ++c->rec_ticket_counter;
                if (!t) t = ++c->rec_ticket_counter;
                launch_store_word(st, reinterpret_cast<unsigned*>(static_cast<char*>(hdev) + 128), t);
                c->lane_spec_ticket = t;
            } else (void)hipGetLastError();
        }
        HIP_TRY(hipGetLastError());
        int nrc = note_range_frame(c, c->readers, st, f0, f0 + 1);   // the next upload into this slot waits for the overlay's reads
        c->lane_spec_reader_seq = c->readers.lazy_seq;               // (what the completion word, once seen, proves finished)
        return nrc ? nrc : note_range_frame(c, c->writers, st, f0, f0 + 1);   // ... and an overlay on the presentation stream for this one's stores
    });
}

// test hook: the row intervals k_lane_spans_from_fit forms for a fit given by value (a record of its own in the slot's place)
int lt_lane_spans_from_fit(lt_ctx* c, const double* fit6, int detected, int fit_flags, const double* prev_sum, int count, const double* ploty,
                           const double* ploty2, int n_rows, int16_t* spans_out) {
    if (!c || !fit6 || !ploty || !ploty2 || !spans_out || n_rows < 1 || count < 1) return fail(LT_ERR_INVALID, "lt_lane_spans_from_fit: bad arguments");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    const int bh = c->calib.warp_h;
    lt_lane_record rec;
    std::memset(&rec, 0, sizeof rec);
    for (int k = 0; k < 3; ++k) { rec.left_coeffs[k] = fit6[k]; rec.right_coeffs[k] = fit6[3 + k]; }
    rec.detected = (uint8_t)(detected != 0);
    rec.fit_flags = (uint8_t)fit_flags;
    uint8_t *d_rec = nullptr, *d_pl = nullptr, *d_sp = nullptr;
    if ((rc = dev_alloc(&d_rec, sizeof rec)) || (rc = dev_alloc(&d_pl, (size_t)2 * n_rows * sizeof(double))) || (rc = dev_alloc(&d_sp, (size_t)bh * 4))) {
        dev_free(d_rec); dev_free(d_pl); dev_free(d_sp);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_rec, &rec, sizeof rec, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pl, ploty, (size_t)n_rows * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pl + (size_t)n_rows * sizeof(double), ploty2, (size_t)n_rows * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_sp, 0x55, (size_t)bh * 4, c->stream);
    bool launched = false;
    if (e == hipSuccess) {
        launched = launch_lane_spans_from_fit(c->stream, reinterpret_cast<const lt_lane_record*>(d_rec), prev_sum, count, reinterpret_cast<const double*>(d_pl),
                                              reinterpret_cast<const double*>(d_pl) + n_rows, n_rows, bh, c->calib.warp_w, reinterpret_cast<int16_t*>(d_sp));
        e = hipGetLastError();
    }
    if (e == hipSuccess && launched) e = hipMemcpyAsync(spans_out, d_sp, (size_t)bh * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_rec); dev_free(d_pl); dev_free(d_sp);
    if (e != hipSuccess) return fail(LT_ERR_HIP, "lt_lane_spans_from_fit failed: %s", hipGetErrorString(e));
    if (!launched) return fail(LT_ERR_STATE, "lt_lane_spans_from_fit: too many rows for one workgroup's LDS");
    return LT_OK;
}

int lt_present_finish(lt_ctx* c, int slot, const char* lines, int n_lines, int line_len, int x0, int y0, int step, uint8_t* out,
                      const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    const bool text = lines && n_lines > 0 && line_len > 0 && c->font_glyphs > 0;
    if ((!c->d_annot || !c->present) && (!c->lane_spec_stream || text || (rows4 && rows4[1] > rows4[0])))
        return fail(LT_ERR_STATE, "lt_present_finish before lt_present_lane_async");
    int r[4];
    if ((rc = present_rows(c, rows4, text, n_lines, y0, step, false, true, r))) return rc;
    if ((rc = set_device(c))) return rc;
    if (text && (rc = lt_overlay_text(c, slot, 1, lines, n_lines, line_len, x0, y0, step))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[0], r[1]))) return rc;
    if (c->lane_spec_stream) {       // the lane came from lt_present_lane_from_fit_async, on the slot's own stream
        bool seen = false;
        if (c->lane_spec_ticket && c->h_rec) {
            const volatile unsigned* word = reinterpret_cast<const volatile unsigned*>(reinterpret_cast<const char*>(c->h_rec) + 128);
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; !(seen = *word == c->lane_spec_ticket); ++spins) {
                __builtin_ia32_pause();
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(c->lane_spec_stream));
        // the word sits behind the frame's overlay on the slot's stream: every unrecorded reader noted up to its launch is done
        if (c->readers.lazy && c->lane_spec_stream == c->stream && c->lane_spec_reader_seq > c->readers.lazy_seen) c->readers.lazy_seen = c->lane_spec_reader_seq;
        c->lane_spec_stream = nullptr;
        c->lane_spec_ticket = 0;
    }
    if (c->present) HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

static void harvest_downloads(lt_ctx* c);
static int choose_download(lt_ctx* c);
static int download_overlay_async_impl(lt_ctx* c, int first, int n, uint8_t* out, const int* rows4);
int lt_download_overlay_async(lt_ctx* c, int first, int n, uint8_t* out) { return download_overlay_async_impl(c, first, n, out, nullptr); }
int lt_download_overlay_rows_async(lt_ctx* c, int first, int n, uint8_t* out, const int32_t* rows4) {
    if (!rows4) return download_overlay_async_impl(c, first, n, out, nullptr);
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int r[4];
    const int rc = ordered_rows(c, rows4, r);
    if (rc) return rc;
    return download_overlay_async_impl(c, first, n, out, r);
}
// rows4: only these two runs of rows of every frame (at their places in `out`), nullptr: whole frames
static int download_overlay_async_impl(lt_ctx* c, int first, int n, uint8_t* out, const int* rows4) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_download_overlay_async before lt_overlay_run");
    if (!rows4) {
        const int bad = first_partial(c->annot_full, first, n);
        if (bad >= 0) return fail(LT_ERR_STATE, "slot %d holds row runs of its annotated frame only (lt_overlay_run_rows / lt_present_*): no whole frame to download", bad);
    }
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // on a stream of its own, behind the overlay work enqueued so far: the copy neither holds up the kernels queued behind
    // it on the context's stream nor shares a queue with the uploads
    if (!c->dl) {
        // the reserved CUs but the first are the copy kernel's (lt_set_search_cus); otherwise the highest priority: the copy kernel's
        // few workgroups go ahead of the mask kernels'
        if (c->search_cus >= 2) HIP_TRY(stream_get(&c->dl, SK_CU_SET, -c->search_cus));
        else HIP_TRY(stream_get(&c->dl, SK_PRIORITY, 0));
    }
    hipEvent_t e = next_order_event(c);
    if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(e, c->present ? c->present : c->stream));
    HIP_TRY(hipStreamWaitEvent(c->dl, e, 0));
    // engine or kernel: by measurement (choose_download); LT_DL_KERNEL=1 / 0 and lt_set_download_method pin one of them
    static const int env_method = [] { const char* e = LT_EXP_ENV("LT_DL_KERNEL"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    if (env_method >= 0 && c->dl_forced < 0) c->dl_forced = env_method;
    harvest_downloads(c);
    int method = choose_download(c);
    auto timing_event = [&]() -> hipEvent_t {
        hipEvent_t ev = nullptr;
        if (!c->dl_event_pool.empty()) { ev = c->dl_event_pool.back(); c->dl_event_pool.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
        return ev;
    };
    hipEvent_t ta = timing_event(), tb = timing_event();
    if (ta && tb) HIP_TRY(hipEventRecord(ta, c->dl));
    size_t bytes = (size_t)n * c->frame_bytes;
    const uint8_t* src = c->d_annot + (size_t)first * c->frame_bytes;
    if (!rows4) {
        if (method == 1 && !launch_copy_to_pinned(c->dl, out, src, bytes)) method = 0;   // not page-locked / aligned
        if (method == 0) HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->dl));
    } else {
        const size_t row_bytes = (size_t)c->calib.img_w * 3;
        bytes = 0;
        // The kernel takes a run of rows of all the frames in one launch; the engine takes the run of ONE frame as an ordinary
        // copy (a pitched copy over the frames it takes row by row: 120-160 ms for a window of 256 frames), two copies per
        // frame -- 18.4 k frames/s of an annotated 1280x720 stream against the kernel's 22.6 k.  So the kernel, unless
        // lt_set_download_method(0) / LT_DL_KERNEL=0 ask for the engine.
        bool by_kernel = c->dl_forced != 0;
        for (int k = 0; k < 4 && by_kernel; k += 2)       // both runs the same way, so that the timing below means one thing
            by_kernel = rows4[k + 1] <= rows4[k] ||
                        ((((size_t)rows4[k] * row_bytes) | ((size_t)(rows4[k + 1] - rows4[k]) * row_bytes) | c->frame_bytes | (size_t)(uintptr_t)out) & 15) == 0;
        for (int k = 0; k < 4; k += 2) {
            if (rows4[k + 1] <= rows4[k]) continue;
            const size_t off = (size_t)rows4[k] * row_bytes, run = (size_t)(rows4[k + 1] - rows4[k]) * row_bytes;
            if (by_kernel && !launch_copy_rows_to_pinned(c->dl, out, src, c->frame_bytes, off, run, n)) by_kernel = false;
            bytes += run * (size_t)n;
        }
        if (!by_kernel)
            for (int f = 0; f < n; ++f)
                for (int k = 0; k < 4; k += 2) {
                    if (rows4[k + 1] <= rows4[k]) continue;
                    const size_t off = (size_t)f * c->frame_bytes + (size_t)rows4[k] * row_bytes;
                    HIP_TRY(hipMemcpyAsync(out + off, src + off, (size_t)(rows4[k + 1] - rows4[k]) * row_bytes, hipMemcpyDeviceToHost, c->dl));
                }
        method = by_kernel ? 1 : 0;
    }
    HIP_TRY(hipGetLastError());
    if (ta && tb) {
        HIP_TRY(hipEventRecord(tb, c->dl));
        c->dl_inflight.push_back({ta, tb, (double)bytes, method});
    } else {
        if (ta) c->dl_event_pool.push_back(ta);
        if (tb) c->dl_event_pool.push_back(tb);
    }
    if (c->annot_busy.hi <= c->annot_busy.lo) { c->annot_busy.lo = first; c->annot_busy.hi = first + n; }
    else { c->annot_busy.lo = std::min(c->annot_busy.lo, first); c->annot_busy.hi = std::max(c->annot_busy.hi, first + n); }
    return staging_mark(c->annot_busy, c->dl);
}

// Engine or kernel?  The copy engine moves the frames at 48-56 GB/s when the page-locked destination and the device buffer are
// laid out kindly, and at 28-30 GB/s when they are not -- a property of the memory the process happened to get (allocation
// history, the box), not of anything this library orders: tools/copy_engine_probe.py RAW=1 shows one lone download at 29 GB/s
// on the same engine, same code path, beside nothing.  (Rounds 2-3 read the resulting 9.3 k instead of 15 k frames/s of
// the annotated 1280x720 stream as uploads and downloads "taking turns"; they do overlap.)  A kernel storing 16 bytes per lane
// into the same destination is not affected (an annotated 1280x720 stream does 13.3 k frames/s that way in either regime: less
// than the engine at its best, 15 k, because the copy kernel shares the chip with the mask chain, far more than the engine at
// its worst).  So: every copy is timed with an event pair; the engine is the default; when its running rate drops below
// DL_SLOW GB/s the kernel takes over, and one copy in DL_REPROBE goes by the engine again so that a recovery is noticed.
static constexpr double DL_SLOW = 36.0;
static constexpr int DL_REPROBE = 48;
static void harvest_downloads(lt_ctx* c) {
    size_t keep = 0;
    for (size_t i = 0; i < c->dl_inflight.size(); ++i) {
        lt_ctx::DlTimed& d = c->dl_inflight[i];
        float ms = 0.f;
        if (hipEventQuery(d.b) == hipSuccess && hipEventElapsedTime(&ms, d.a, d.b) == hipSuccess) {
            if (ms > 0.f && d.bytes >= 8e6) {              // small copies time the launch, not the bus
                const double r = d.bytes / (ms * 1e-3) / 1e9;
                c->dl_rate[d.method] = c->dl_samples[d.method] ? 0.5 * c->dl_rate[d.method] + 0.5 * r : r;
                ++c->dl_samples[d.method];
            }
            c->dl_event_pool.push_back(d.a);
            c->dl_event_pool.push_back(d.b);
        } else {
            (void)hipGetLastError();
            c->dl_inflight[keep++] = d;
        }
    }
    c->dl_inflight.resize(keep);
}
static int choose_download(lt_ctx* c) {
    if (c->dl_forced >= 0) return c->dl_forced;
    const int cur = c->dl_method, other = 1 - cur;
    ++c->dl_since_probe;
    if (c->dl_samples[cur] >= 4) {       // (the first copies of a stream are short and wait for their overlays: not a verdict)
        const bool never = c->dl_samples[other] == 0;
        if (cur == 0 && c->dl_rate[0] < DL_SLOW && (never || c->dl_rate[1] > c->dl_rate[0])) { c->dl_method = 1; c->dl_since_probe = 0; }
        else if (cur == 1 && c->dl_rate[0] >= DL_SLOW) { c->dl_method = 0; c->dl_since_probe = 0; }   // the engine has recovered
        else if (c->dl_since_probe >= DL_REPROBE && (cur == 1 || c->dl_rate[0] < DL_SLOW)) {   // one copy the other way
            c->dl_since_probe = 0;
            return other;
        }
    }
    return c->dl_method;
}

int lt_set_download_method(lt_ctx* c, int method) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (method < -1 || method > 1) return fail(LT_ERR_INVALID, "download method: -1 = measured choice, 0 = copy engine, 1 = kernel");
    c->dl_forced = method;
    return LT_OK;
}

int lt_download_stats(lt_ctx* c, double* engine_gbs, int* engine_copies, double* kernel_gbs, int* kernel_copies, int* method) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    harvest_downloads(c);
    if (engine_gbs) *engine_gbs = c->dl_rate[0];
    if (engine_copies) *engine_copies = c->dl_samples[0];
    if (kernel_gbs) *kernel_gbs = c->dl_rate[1];
    if (kernel_copies) *kernel_copies = c->dl_samples[1];
    if (method) *method = c->dl_forced >= 0 ? c->dl_forced : c->dl_method;
    return LT_OK;
}

int lt_download_overlay_wait(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    if (!c->dl) return LT_OK;
    HIP_TRY(hipStreamSynchronize(c->dl));   // every copy of lt_download_overlay_async is behind its slots' overlay kernels
    c->annot_busy.lo = c->annot_busy.hi = 0;
    return LT_OK;
}

int lt_download_bev(lt_ctx* c, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!c->have_mask) return fail(LT_ERR_STATE, "lt_download_bev before lt_mask_run");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    if ((rc = ensure_bev(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    uint8_t* dst = c->d_bev + (size_t)first * c->bev_bytes;
    if (c->fe.nrows <= 0) HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * c->bev_bytes, c->stream));
    else
        launch_warp_rgb(c->stream, c->d_und, c->und_px, first, c->d_wxy, c->d_wfrac, c->fe, dst, c->bev_bytes, n);
    HIP_TRY(hipGetLastError());
    return download(c, dst, out, (size_t)n * c->bev_bytes);
}

}  // extern "C"
