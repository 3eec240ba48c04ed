for S in 3 4 5 6 8; do python bench.py --streams $S --steps 20 --no-cpu-baseline --no-host-fed --no-stream --no-settings 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('streams', d['config']['streams_per_gpu'], d['value'], d['overlapped_batches_frames_per_s'], d['ms_per_step'])
"; done
