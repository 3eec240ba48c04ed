O=gpurun_out/r05B; mkdir -p $O
cat /sys/class/drm/card*/device/numa_node > $O/gpu_nodes.txt 2>&1; cat /sys/devices/system/node/node*/cpulist >> $O/gpu_nodes.txt
python - >> $O/gpu_nodes.txt <<'PY'
import ctypes, os
print("HIP_VISIBLE_DEVICES", os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
PY
for rep in 1 2; do
for s in 1280x720 1920x1080; do
  timeout 1200 python tools/annot_ab.py $s True "" "AB_NUMA_NODE=0" "AB_NUMA_NODE=1" >> $O/ab.log 2>&1
done; done
