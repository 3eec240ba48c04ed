#!/bin/bash
# The two places where a workaround for a suspected hipcc code-generation problem is in the source ("Toolchain
# cases"): build the library WITHOUT each workaround, run the tests that caught the problem against that build, and keep
# the ISA of the affected kernel with and without the workaround.  Run on the GPU box from the repo root:
#     bash tools/toolchain_cases.sh          -> gpurun_out/cases/{report.txt, *.s.diffstat, ...}
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/cases
mkdir -p $out
cd $root/lane_tracker_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950"
OBJS="lt_api.o lt_memory.o lt_present.o lt_chain.o lt_gather.o lt_tables.o k_filter.o k_tophat.o k_threshold.o k_threshold_walk.o k_overlay.o"
report=$out/report.txt
/opt/rocm/bin/hipcc --version | head -2 > $report
build_variant() {   # name, file, define
  name=$1; file=$2; def=$3
  /opt/rocm/bin/hipcc $FLAGS -D$def -c $file.hip -o /tmp/$file.$name.o || return 1
  others="k_frontend.o k_search.o"
  others=${others/$file.o//tmp/$file.$name.o}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $OBJS $others || return 1
  /opt/rocm/bin/hipcc $FLAGS -D$def -S --cuda-device-only $file.hip -o $out/$file.$name.s 2>/dev/null
  /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only $file.hip -o $out/$file.shipped.s 2>/dev/null
}
cd $root/lane_tracker_amd/csrc && make -s -j8 >/dev/null
# ---- case 1: k_warp_split4 without the asm barrier
build_variant warp_no_barrier k_frontend LT_CASE_WARP_NO_BARRIER
cd $root
echo "== case 1: k_warp_split4 without the register barrier (tests: front end + mask chain parity)" >> $report
LANE_TRACKER_AMD_LIB=$out/lib_warp_no_barrier.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "front_end or mask_chain" 2>&1 | tail -4 >> $report
diff <(grep -v "^\s*;" $out/k_frontend.shipped.s | grep -v "^\s*\.") <(grep -v "^\s*;" $out/k_frontend.warp_no_barrier.s | grep -v "^\s*\.") > $out/k_frontend.isa.diff
echo "ISA diff lines (shipped vs without barrier): $(wc -l < $out/k_frontend.isa.diff)" >> $report
# ---- case 2: sws2_recurrence inlined
cd $root/lane_tracker_amd/csrc
build_variant sws2_inline k_search LT_CASE_SWS2_INLINE
cd $root
echo "== case 2: sws2_recurrence inlined into k_sws_fit2 (tests: golden sliding-window fixtures on the GPU)" >> $report
LANE_TRACKER_AMD_LIB=$out/lib_sws2_inline.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sliding_window_search or search_fuzz or search_kernel_limits" 2>&1 | tail -4 >> $report
diff <(grep -v "^\s*;" $out/k_search.shipped.s | grep -v "^\s*\.") <(grep -v "^\s*;" $out/k_search.sws2_inline.s | grep -v "^\s*\.") | head -4000 > $out/k_search.isa.diff
echo "ISA diff lines (shipped vs inlined, first 4000): $(wc -l < $out/k_search.isa.diff)" >> $report
# the shipped build on the same tests, for the record
echo "== shipped build on the same tests" >> $report
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "front_end or mask_chain or sliding_window_search or search_fuzz or search_kernel_limits" 2>&1 | tail -3 >> $report
rm -f $out/*.so $out/*.s     # keep the report and the diffs only (the merge back is limited to 64 MiB)
cat $report
