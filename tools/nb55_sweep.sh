for nb in 3 4 5 6 7 8; do
    r=$(env LT_MORPH_NB_55E=$nb LT_MORPH_NB_55D=$nb timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read())['kernels_ms_per_step']; print(d['erode_b55'], d['tophat_b55'])")
    echo "55 nb=$nb : $r"
done
