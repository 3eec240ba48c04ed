for st in 1 2 3 4 5; do for rep in 1 2; do
v=$(timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams $st | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"); echo "streams=$st : $v"; done; done
for f in 4 8 16 32; do v=$(LT_FRONTEND_FPB=$f timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"); echo "fpb_cap=$f : $v"; done
