"""CPU: the checker itself under AddressSanitizer + UndefinedBehaviorSanitizer.
`make -C oracle asan-test` builds oracle/liboracle_asan.so (-fsanitize=address,undefined, no OpenMP) and runs the CPU parts of
the oracle's own tests against it in a child interpreter with the sanitizer runtimes preloaded; any report aborts that run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    cc = shutil.which("cc") or shutil.which("gcc")
    if cc is None:
        pytest.skip("no C compiler")
    asan = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan is not installed")
    env = dict(os.environ)
    env.pop("HB_ORACLE_LIB", None)
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan-test"], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the instrumented library really was the one under test
    probe = subprocess.run(["python3", "-c", "import sys; sys.path.insert(0, '.'); from oracle import oracle_py as O; print(O.build())"],
                           capture_output=True, text=True, cwd=ROOT, env=dict(env, HB_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so")))
    assert probe.stdout.strip().endswith("liboracle_asan.so"), probe.stdout + probe.stderr
