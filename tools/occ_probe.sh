for l in 0 30000 50000; do echo extra_lds=$l; LT_MORPH_EXTRA_LDS=$l timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], {k: round(v, 3) for k, v in d['kernels_ms_per_step'].items() if 'r29' in k or 'b55' in k})"; done
