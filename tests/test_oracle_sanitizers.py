"""CPU: the C oracle (oracle/fa_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer: the golden
suite of tests/test_oracle_golden.py -- every env trajectory, the MT19937 known answers, the choice stream -- runs
against `make asan`'s build in a child interpreter with libasan preloaded; any report (heap overflow, use after free,
signed overflow, misaligned access, out-of-range shift ...) aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_suite_is_clean_under_asan_and_ubsan():
    orc = os.path.join(ROOT, "oracle")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", orc, "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", FA_ORACLE_LIB=os.path.join(orc, "libfa_oracle_asan.so"),
               OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]
    assert " passed" in text
