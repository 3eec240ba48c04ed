#!/bin/bash
# Shader clock and power while the bench loop runs (GPU box): samples rocm-smi every 0.5 s during a long bench run
python bench.py --steps 3000 --warmup 3 --no-cpu-baseline > /tmp/clock_bench.json 2>/dev/null &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' '; echo
  sleep 0.5
done
wait $pid
python -c "import json; d=json.load(open('/tmp/clock_bench.json')); print('value', d['value'], 'ms', d['ms_per_step'])"
