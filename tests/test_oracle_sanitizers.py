"""CPU: the oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C oracle asan`): the golden-vector suite, the strided-pointer entry points
(oracle_select_best_path, oracle_closed_loop) and the new spiral rollout run in a child interpreter
with libasan preloaded; any report aborts the child (-fno-sanitize-recover, halt_on_error)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_suite_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    env = dict(os.environ, LD_PRELOAD=libasan, VDYN_ORACLE_LIB=os.path.join(REPO, "oracle", "libvdyn_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=REPO, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
