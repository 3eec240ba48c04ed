"""The C restatement under AddressSanitizer + UndefinedBehaviorSanitizer ("run CPU restatement
under -fsanitize=address,undefined in tests").  `make -C oracle asan` builds liblt_oracle_asan.so; the golden-vector
and unit tests of the oracle then run in a child interpreter with the sanitizer runtime preloaded.  Any report makes
the child exit non-zero (halt_on_error, -fno-sanitize-recover is the default for ASan; UBSan is told to abort).
CPU only -- GPU sanitizers are not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_golden_and_unit_tests_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "asan"])
    env = dict(os.environ, LT_ORACLE_SANITIZED="1", LD_PRELOAD=asan + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=86",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=87")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_units.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout
