"""Guards on the generated gfx950 code (CPU: hipcc cross-compiles; no GPU needed).

ROCm 7.2's hipcc can turn two "shift right, clamp to 0..255" results packed into one word into V_ASHR_PK_U8_I32 and OR
further bytes in above bit 16 -- but the MI355X keeps bits 31:16 of that instruction's destination, so the word is
wrong (tests/cases/ashr_pk_u8.hip shows it on the GPU; DESIGN.md "Toolchain cases").  k_warp_split4 carries a register
barrier against it.  This test compiles every kernel file to assembly and fails if the instruction (or its signed
sibling) appears anywhere in the product, i.e. if a future edit or toolchain re-introduces the pattern."""
import glob
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lane_tracker_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(path):
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-S",
                        "--cuda-device-only", "-I", CSRC, path, "-o", "-"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_packed_clamp_shift_instruction_in_the_product():
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    assert files
    with ThreadPoolExecutor(4) as ex:
        listings = list(ex.map(_isa, files))
    for path, text in zip(files, listings):
        hits = [ln.strip() for ln in text.splitlines() if re.search(r"\bv_ashr_pk_[ui]8_i32\b", ln)]
        assert not hits, "%s: %d uses of v_ashr_pk_*8_i32, e.g. %s" % (os.path.basename(path), len(hits), hits[0])
        # the hand-written pieces of the walking threshold kernels must still be there (a guard against silent rewrites)
        if path.endswith("k_threshold_walk.hip"):
            assert "v_cmp_le_i16_e64" in text and "ds_read_b128" in text and "global_load_dwordx4" in text
            assert ".vgpr_spill_count: 0" in text or "vgpr_spill_count:     0" in text
        # the two-row top-hat kernels compare u8 pixels as f16 denormals: every one of them must run with f16 denormals
        # kept (kernel descriptor) and set the mode itself before its first min / max
        if path.endswith("k_tophat.hip"):
            kernels = re.split(r"\n(?=_ZN2lt[^\n]*k_morph_runs2[^\n]*:)", text)[1:]
            assert len(kernels) >= 12
            for k in kernels:
                body = k.split(".end_amdhsa_kernel")[0]
                assert re.search(r"s_setreg_imm32_b32 hwreg\(HW_REG_MODE, 6, 2\), 3", body), k.splitlines()[0]
                assert re.search(r"\.amdhsa_float_denorm_mode_16_64 3", body), k.splitlines()[0]
                assert "scratch_" not in body.split("s_endpgm")[0], k.splitlines()[0]
