import json
for f in ("gpurun_out/process_soak_720.json","gpurun_out/process_soak_1080.json"):
    d=json.load(open(f))
    print(d["size"], d["frames"], d["reopened"], d["rss_mb_first_last_max"])
    for s in d["samples"]: print("  ", s["t_s"], s["frames_per_s"], s["rss_mb"], s["host"]["staging_bytes"], s["device_cache"]["kept_bytes"]+s["device_cache"]["live_bytes"], s["evicted_bytes"], s["aperture_uploads"], s["success"])
