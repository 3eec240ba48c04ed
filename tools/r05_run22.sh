O=gpurun_out/r05v; mkdir -p $O
timeout 300 python tools/annot_cprofile.py 1280x720 True > $O/cprof_720_true.txt 2>&1
timeout 300 python tools/annot_cprofile.py 1280x720 inplace > $O/cprof_720_inplace.txt 2>&1
timeout 300 python tools/annot_cprofile.py 1280x720 False > $O/cprof_720_plain.txt 2>&1
timeout 300 python tools/annot_cprofile.py 1920x1080 True > $O/cprof_1080_true.txt 2>&1
