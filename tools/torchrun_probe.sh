#!/bin/bash
# Batch rate of bench.py inside a process that also holds an RCCL process group (what the multi-GPU launch does),
# with the slot-slice streams at the highest priority (default) and as plain streams.  Plain streams share the
# runtime's pool of hardware queues with torch's and RCCL's streams: two slices end up on one queue and the rate
# drops by 7-9 % (1 MI355X: 87.6 ms vs 82.0 ms for 20 steps of 256 frames).
export LT_BENCH_VERBOSE=1
k=${1:-20}
D="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1"
run() { echo "$1"; shift; env "$@" timeout 300 python bench.py --gpus 1 --steps $k --warmup 3 --no-cpu-baseline 2>&1 | grep -E "timed region"; }
run plain-high X=1
run plain-normal LT_STREAM_PRIORITY=normal
run dist-high $D MASTER_PORT=29513
run dist-normal $D MASTER_PORT=29514 LT_STREAM_PRIORITY=normal
