O=gpurun_out/r05b; mkdir -p $O
for i in 1 2 3; do
  LT_TRACE_START=1 LT_BENCH_TRACE=1 LT_BENCH_OLD_FIRST=1 python bench.py > $O/bench_old_$i.json 2> $O/bench_old_$i.err
done
for i in 1 2; do
  LT_TRACE_START=1 LT_BENCH_TRACE=1 python bench.py > $O/bench_new_$i.json 2> $O/bench_new_$i.err
done
for i in 1 2 3 4; do timeout 200 python tools/close_hang.py --cache-gb 32 --limit 60 > $O/close_hang_$i.log 2>&1; done
timeout 300 python tools/close_hang.py --cache-gb 32 --limit 60 --size 1920x1080 > $O/close_hang_1080.log 2>&1
