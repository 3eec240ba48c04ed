#!/usr/bin/env python3
"""One-off: split csrc/lt_api.cpp (2915 lines at commit a6e66b7) into lt_ctx.h + lt_api.cpp + lt_memory.cpp +
lt_present.cpp + lt_chain.cpp by line ranges.  Kept for the record of how the split was made; not part of the build."""
import os
import re
import sys

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lane_tracker_amd", "csrc")
src = open(os.path.join(HERE, "lt_api.cpp")).read().split("\n")
assert len(src) >= 2915, len(src)


def L(a, b):          # 1-based inclusive
    return src[a - 1:b]


def unstatic(lines, names):
    out = []
    for ln in lines:
        for n in names:
            ln = re.sub(r"^static (.*\b%s\()" % re.escape(n), r"\1", ln)
        out.append(ln)
    return out


INCLUDES = """#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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
""".split("\n")

# ---------------------------------------------------------------- lt_ctx.h
ctx_h = []
ctx_h += """// The context behind the C ABI (struct lt_ctx) and the helpers its translation units share:
//   lt_api.cpp     -- create / destroy / reserve, uploads, downloads, the mask chain, the searches, measurement
//   lt_memory.cpp  -- device-memory cache, page-locked host memory, the host copy threads
//   lt_present.cpp -- presentation stage: lane overlay, text, annotated frames on their way back
//   lt_chain.cpp   -- the chained band search of a stream (tickets, cancel, collect)
// Not installed; the public ABI is include/lane_tracker_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "lt_internal.h"

namespace lt {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // fills lt_last_error(), returns code

#define HIP_TRY(expr)                                                                               \\
    do {                                                                                            \\
        hipError_t e_ = (expr);                                                                     \\
        if (e_ != hipSuccess) return lt::fail(LT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \\
    } while (0)

enum Stage {
    ST_UNDISTORT = 0, ST_WARP_SPLIT, ST_ERODE_R, ST_TOPHAT_R, ST_ERODE_B, ST_TOPHAT_B, ST_THRESHOLD, ST_MERGE,
    ST_OPEN, ST_SWS_FIT, ST_BAND_FIT, ST_SPLIT_BEV
};

enum Plane { P_R = 0, P_B, P_THR, P_THB, P_MERGED, P_MASK, P_T0, P_T1, P_T2, P_T3, P_COUNT };

}  // namespace lt

using lt::P_COUNT;
""".split("\n")
body = L(57, 211)
body = [ln.replace("FrontEndGeom fe{};", "lt::FrontEndGeom fe{};").replace("EllipseSE se5{}, se29{}, se55{};", "lt::EllipseSE se5{}, se29{}, se55{};")
        for ln in body]
ctx_h += body
ctx_h += """
namespace lt {

// ---- device memory (lt_memory.cpp): a cache in front of hipMalloc / hipFree ------------------------------
void* cached_alloc(size_t bytes);
void cached_free(void* p);
""".split("\n")
ctx_h += L(295, 306)
ctx_h += """
// ---- streams, slot ranges, ordering (lt_api.cpp) ---------------------------------------------------------
int sync_all(lt_ctx* c);
int note_range(lt_ctx::RangeEvents& r, hipStream_t st, int lo, int hi);
int wait_range(const lt_ctx::RangeEvents& r, hipStream_t waiter, int lo, int hi, bool* precise);
int note_written(lt_ctx* c, hipStream_t st, int lo, int hi);
hipEvent_t next_order_event(lt_ctx* c);
int wait_chains(lt_ctx* c, hipStream_t st, int lo, int hi);
int flush_stage_events(lt_ctx* c);
int check_slots(lt_ctx* c, int first, int n);
int set_device(lt_ctx* c);
hipError_t create_compute_stream(hipStream_t* st, int reserved = 0);
int download(lt_ctx* c, const void* src, void* dst, size_t bytes);
int ensure_bev(lt_ctx* c);
int ensure_search_buffers(lt_ctx* c, int maxpix, int maxlev);
bool masks_have_bits(const lt_ctx* c, int first, int n);
int ensure_u8_masks(lt_ctx* c, int first, int n);
int make_search_geom(lt_ctx* c, const lt_search_params* p, bool band, SearchGeom& g);

""".split("\n")
# for_each_slice template (365-428 minus the wait_chains definition 371-378 and the next_order_event forward decl)
ctx_h += L(365, 369)
ctx_h += L(379, 428)
ctx_h += [""]
ctx_h += L(444, 470)       # StageScope
ctx_h += ["", "}  // namespace lt", ""]
open(os.path.join(HERE, "lt_ctx.h"), "w").write("\n".join(ctx_h))

# ---------------------------------------------------------------- lt_memory.cpp
mem = ["// Device-memory cache, page-locked host memory and the host copy threads of liblane_tracker_amd.so", "// (see lt_ctx.h; public entry points: lt_device_cache_trim, lt_host_alloc / lt_host_free, lt_host_copy_*)."]
mem += INCLUDES
mem += ["namespace lt {", ""]
mem += unstatic(L(215, 293), ["dev_cache", "cached_alloc", "cached_free"])
mem += ["", "}  // namespace lt", "", 'extern "C" {', ""]
mem += L(1805, 1921)
mem += ["", '}  // extern "C"', ""]
open(os.path.join(HERE, "lt_memory.cpp"), "w").write("\n".join(mem))

# ---------------------------------------------------------------- lt_present.cpp
pre = ["// Presentation stage of liblane_tracker_amd.so (SURVEY 8(f) N1; lane_tracker.py:629-793): lane overlay, text lines,",
       "// annotated frames on their way back to the host.  See lt_ctx.h."]
pre += INCLUDES
pre += L(1358, 1418)
pre += ["", 'extern "C" {', ""]
pre += L(1464, 1803)
pre += [""]
pre += L(1923, 2217)
pre += ["", '}  // extern "C"', ""]
open(os.path.join(HERE, "lt_present.cpp"), "w").write("\n".join(pre))

# ---------------------------------------------------------------- lt_chain.cpp
ch = ["// The chained band search of one stream (lt_band_fit_chain_run / _cancel / _collect; lane_tracker.py:449-509, 851-872):",
      "// one workgroup walks the resident masks of consecutive frames and hands the fit on.  See lt_ctx.h."]
ch += INCLUDES
ch += ['extern "C" {', ""]
ch += L(2436, 2445)
ch += [""]
ch += L(2535, 2613)
ch += [""]
ch += L(2655, 2682)
ch += ["", '}  // extern "C"', ""]
open(os.path.join(HERE, "lt_chain.cpp"), "w").write("\n".join(ch))

# ---------------------------------------------------------------- lt_api.cpp (what stays)
api = L(1, 3)
api += INCLUDES
api += ["namespace {", "", "thread_local std::string g_err;", "", "}  // namespace", "",
        "namespace lt {", "",
        "int fail(int code, const char* fmt, ...) {", "    char buf[512];", "    va_list ap;", "    va_start(ap, fmt);",
        "    vsnprintf(buf, sizeof buf, fmt, ap);", "    va_end(ap);", "    g_err = buf;", "    return code;", "}", "",
        "const char* kStageNames[LT_NUM_STAGES] = {\"undistort_rows\", \"warp_split\", \"erode_r29\", \"tophat_r29\", \"erode_b55\",",
        "                                          \"tophat_b55\", \"threshold\", \"merge\", \"open5\", \"sws_fit\", \"band_fit\",",
        "                                          \"split_bev\"};", ""]
api += L(308, 363)
api += [""]
api += unstatic(L(371, 378), ["wait_chains"])
api += [""]
api += L(430, 442)
api += [""]
api += L(472, 817)
api += [""]
api += unstatic(L(835, 857), ["create_compute_stream"])
api += ["", "}  // namespace lt", ""]
api += L(821, 833)
api += [""]
api += L(859, 1271)
api += ["}  // extern \"C\"", "namespace lt {"]
api += unstatic(L(1272, 1297), ["download"])
api += ["}  // namespace lt", "extern \"C\" {", ""]
api += L(1299, 1356)
api += [""]
api += L(1420, 1462)
api += [""]
api += L(2219, 2434)
api += [""]
api += L(2447, 2533)
api += [""]
api += L(2615, 2653)
api += [""]
api += L(2684, 2915)
txt = "\n".join(api)
txt = txt.replace("hipError_t create_compute_stream(hipStream_t* st, int reserved = 0) {", "hipError_t create_compute_stream(hipStream_t* st, int reserved) {")
open(os.path.join(HERE, "lt_api.cpp"), "w").write(txt)
print("done")
