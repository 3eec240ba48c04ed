#!/usr/bin/env python3
"""Which SDMA engine does the runtime hand to the uploads and to the downloads of an annotated stream?  Runs the stream in child
processes with the runtime's own log on (AMD_LOG_LEVEL=4; rocclr prints one "HSA Copy copy_engine=0x.. engineType=.." line per
copy) -- once as the first thing a process does and once after the calls bench.py's stream leg makes before it (process(),
process_batch() plain and annotated, a plain stream) -- and tabulates (engineType, copy_engine) -> count and bytes, beside the
frames/s of each.  usage: copy_engine_probe.py [fresh|bench] (no argument: both, as children)"""
import collections, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def child(mode):
    node = os.environ.get("NODE")
    if node is not None:       # run (and first-touch / page-lock memory) on the CPUs of one NUMA node
        os.sched_setaffinity(0, _cpulist(open("/sys/devices/system/node/node%s/cpulist" % node).read()))
    import numpy as np
    from lane_tracker_amd import calib, synth
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.scaled_calibration(1.5) if mode.endswith("1080") else calib.reference_calibration()
    n = 256
    base = synth.stream_lanes(32, seed=5, cal=cal)
    frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
    if os.environ.get("NODE_IN") is not None:     # the source frames first-touched on one NUMA node, everything else wherever it lands
        full = os.sched_getaffinity(0)
        os.sched_setaffinity(0, _cpulist(open("/sys/devices/system/node/node%s/cpulist" % os.environ["NODE_IN"]).read()))
        frames = np.empty_like(frames)
        frames[...] = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n]
        os.sched_setaffinity(0, full)
    if os.environ.get("PINNED") == "1":           # page-locked source frames: no pin-on-the-fly in the upload path
        from lane_tracker_amd import _native
        pf = _native.pinned_empty(frames.shape)
        pf[...] = frames
        frames = pf
    lt = LaneTracker(**cal)
    if os.environ.get("RESERVE"):                 # the context at its final size before anything runs: no growth later
        lt._ctx.reserve(int(os.environ["RESERVE"]))
    pre = os.environ.get("PRE", "pbas") if mode.startswith("bench") else ""
    if "p" in pre:
        for f in frames[:40]:
            lt.process(f)
    if "b" in pre:
        lt.process_batch(frames, annotate=False)
    if "a" in pre:
        lt.process_batch(frames, annotate=True)
    if "s" in pre:
        list(lt.process_stream([frames] * 4, annotate=False))
    if "A" in pre:          # a short annotated stream, then a pause
        for _ in lt.process_stream([frames[:64]] * 2, annotate=True):
            pass
    wins = [frames] * 8
    if os.environ.get("LIKE_BENCH"):              # bench.py's stream windows: N distinct rendered frames, separately allocated windows
        import bench
        nb = int(os.environ["LIKE_BENCH"])
        prm = synth.stream_lane_params(nb, seed=5)
        r = synth.SceneRenderer(cal)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(16) as ex:
            base2 = np.stack(list(ex.map(lambda q: r.render(q[0], q[1], q[2], dashed_phase=q[3])[0], prm)), 0)
        wins = bench.stream_windows(base2, n, 8)
    if os.environ.get("DISTINCT") == "1":         # eight separately allocated windows (same content): nothing the runtime has pinned before
        wins = [frames.copy() for _ in range(8)]
    if os.environ.get("REGISTER") == "1":         # ... page-locked once by the caller instead of chunk by chunk inside every copy
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        t0 = time.perf_counter()
        for w in wins:
            assert hip.hipHostRegister(ctypes.c_void_p(w.ctypes.data), ctypes.c_size_t(w.nbytes), 0) == 0
        print("registered %d windows in %.1f ms" % (len(wins), (time.perf_counter() - t0) * 1e3), flush=True)
    for _ in lt.process_stream(wins[:4], annotate=True):
        pass
    if os.environ.get("RAW") != "1":
        print("MARK measured-part", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in lt.process_stream(wins, annotate=True):
        pass
    dt = time.perf_counter() - t0
    print("MARK end", file=sys.stderr, flush=True)
    if os.environ.get("HOSTPROF") == "1":
        import cProfile, io, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in lt.process_stream(wins, annotate=True):
            pass
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
        print(st.getvalue()[:3500], flush=True)
    print("RESULT %s: %.0f frames/s annotated stream  %s" % (mode, 8 * n / dt, lt._ctx.download_stats()), flush=True)
    if os.environ.get("RAW") == "1":              # the two copies by themselves and side by side, on the tracker's own buffers
        from lane_tracker_amd import _native
        ctx = lt._ctx
        out = _native.pinned_empty((128,) + frames.shape[1:])
        up = frames[:128]

        def t(fn, reps=6):
            fn(); ctx.sync(); ctx.download_overlay_wait()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.sync(); ctx.download_overlay_wait()
            return (time.perf_counter() - t0) / reps * 1e3
        if os.environ.get("PREHEAT"):                 # do many small copies into a new destination make it a fast one?
            k = int(os.environ["PREHEAT"])
            first = up.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(out, first=512), reps=1)
            for i in range(k):
                ctx.download_overlay_async(out[i % 128:i % 128 + 1], first=512 + i % 128)
            ctx.download_overlay_wait()
            after = up.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(out, first=512), reps=2)
            print("PREHEAT %s: first big copies %.1f GB/s, after %d one-frame copies %.1f GB/s" % (mode, first, k, after), flush=True)
        h2d = t(lambda: (ctx.upload_frame_rows_async(up, first=0), ctx.upload_frame_rest(up, first=0)))
        print("MARK measured-part", file=sys.stderr, flush=True)
        d2h = t(lambda: ctx.download_overlay_async(out, first=512))
        print("MARK end", file=sys.stderr, flush=True)
        both = t(lambda: (ctx.upload_frame_rows_async(up, first=0), ctx.upload_frame_rest(up, first=0), ctx.download_overlay_async(out, first=512)))
        # a second, fresh context in the same process: is the slow download a property of the first one's buffers or of the process?
        c2 = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=256)
        c2.overlay_configure(cal["warp_matrices"][1])
        c2.upload_frames(frames[:256])
        e = np.zeros(0, np.int64)
        c2.overlay_run([(e, e, e, e)] * 256)
        c2.sync()

        def t2(reps=6):
            c2.download_overlay_async(out, first=64); c2.sync(); c2.download_overlay_wait()
            t0 = time.perf_counter()
            for _ in range(reps):
                c2.download_overlay_async(out, first=64)
            c2.sync(); c2.download_overlay_wait()
            return (time.perf_counter() - t0) / reps * 1e3
        d2h_c2 = t2()
        out2 = _native.pinned_empty((128,) + frames.shape[1:])
        d2h_again = t(lambda: ctx.download_overlay_async(out2, first=256))
        print("RAW2 %s: second context down %.2f ms (%.1f GB/s); first context, other slots, other host buffer %.2f ms" % (mode, d2h_c2, up.nbytes / 1e6 / d2h_c2, d2h_again), flush=True)
        c2.close()
        grid = []
        for hb, name in ((out, "out"), (out2, "out2")):
            grid.append(name + ": " + " ".join("%d:%.1f" % (sl, up.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(hb, first=sl), reps=3))
                                               for sl in range(0, min(ctx.capacity, 1024) - 127, 128)))
        print("RAW3 %s: GB/s by first slot -- %s" % (mode, " | ".join(grid)), flush=True)
        if os.environ.get("SCAN") == "1":             # rate against the source slot and the destination offset, 64 frames per copy
            rs = ["%d:%.0f" % (sl, 64 * up[0].nbytes / 1e6 / t(lambda: ctx.download_overlay_async(out[:64], first=sl), reps=2)) for sl in range(0, 960, 37)]
            rd = ["%d:%.0f" % (k, 64 * up[0].nbytes / 1e6 / t(lambda: ctx.download_overlay_async(out[k:k + 64], first=300), reps=2)) for k in range(0, 64, 5)]
            print("SCAN %s: by source slot %s | by destination frame offset %s" % (mode, " ".join(rs), " ".join(rd)), flush=True)
        # user pages (a NumPy allocation: anonymous mmap, transparent huge pages where the kernel grants them), registered with the
        # runtime, against hipHostMalloc'ed blocks allocated at the same moment
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        res = []
        for k in range(4):
            raw = np.empty(up.nbytes + (2 << 20), np.uint8)
            off = (-raw.ctypes.data) % (2 << 20)
            reg = raw[off:off + up.nbytes].reshape(up.shape)
            reg[...] = 0
            rc = hip.hipHostRegister(ctypes.c_void_p(reg.ctypes.data), ctypes.c_size_t(reg.nbytes), 0)
            hm = _native.pinned_empty((128,) + frames.shape[1:])
            hm2 = _native.pinned_empty((127,) + frames.shape[1:])       # another size: never from the pool
            r_reg = up.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(reg, first=0), reps=3)
            r_hm = up.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(hm, first=0), reps=3)
            r_hm2 = hm2.nbytes / 1e6 / t(lambda: ctx.download_overlay_async(hm2, first=0), reps=3)
            res.append("registered(rc %d) %.1f, hipHostMalloc %.1f / %.1f" % (rc, r_reg, r_hm, r_hm2))
            hip.hipHostUnregister(ctypes.c_void_p(reg.ctypes.data))
        print("RAW4 %s: GB/s -- %s" % (mode, " | ".join(res)), flush=True)
        gb = up.nbytes / 1e9
        print("RAW %s: 128 frames up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s), both %.2f ms" % (mode, h2d, gb / h2d * 1e3, d2h, gb / d2h * 1e3, both), flush=True)
    lt.close()


def parent(mode):
    env = dict(os.environ)
    if os.environ.get("PROBE_LOG", "1") == "1":
        env["AMD_LOG_LEVEL"] = "4"
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    tab, on, other, raw = collections.OrderedDict(), False, {}, []
    pat = re.compile(r"HSA Copy copy_engine=(0x[0-9a-f]+).*size=(\d+), forceSDMA=(\d+), engineType=(\d+)")
    for line in p.stderr:
        if "MARK measured-part" in line:
            on = True
        elif "MARK end" in line:
            on = False
        elif on:
            if os.environ.get("RAW") == "1" or re.search(r"[Cc]opy|[Bb]lit|staging|SDMA|sdma|Pin|pin", line):          # every other message about copies, by its text
                key = re.sub(r"0x[0-9a-f]+|\d+", "#", line.split("]", 1)[-1].strip())[:110]
                other[key] = other.get(key, 0) + 1
            if "Query copy engine status" in line and len(raw) < 40:
                raw.append(line.split("]", 1)[-1].strip())
            m = pat.search(line)
            if m:
                k = (int(m.group(4)), m.group(1), int(m.group(3)))
                c = tab.setdefault(k, [0, 0])
                c[0] += 1
                c[1] += int(m.group(2))
    out = p.stdout.read()
    p.wait()
    print(out.strip())
    for (etype, eng, force), (cnt, byts) in sorted(tab.items()):
        print("   engineType=%d copy_engine=%s forceSDMA=%d: %6d copies, %8.1f MB" % (etype, eng, force, cnt, byts / 1e6))
    for k, v in collections.Counter(raw).most_common(8):
        print("   %4d x %s" % (v, k))
    for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:40]:
        print("   %6d x %s" % (v, k))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for m in ("fresh", "bench", "fresh1080", "bench1080"):
            parent(m)
