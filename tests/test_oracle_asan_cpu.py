"""The C restatement (oracle/ancsh_oracle.c) under AddressSanitizer + UBSan: `make -C oracle asan`, then the whole oracle test file in
a python process that preloads libasan and loads the sanitizer build.  The oracle is what every
operator is held to bit for bit -- an out-of-bounds read in ITS loops would silently become the definition of "correct".
Negative control: an output array one row short IS reported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ANCSH_ORACLE_SO=os.path.join(ROOT, "oracle", "_build", "libancsh_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    return env


def test_oracle_suite_is_clean_under_asan_and_ubsan():
    env = _env()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_cpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert r.returncode == 0 and " passed" in out, out[-3000:]


def test_asan_catches_a_short_output_array():
    env = _env()
    code = ("import numpy as np, ctypes, os\n"
            "L = ctypes.CDLL(os.environ['ANCSH_ORACLE_SO'])\n"
            "inp = np.zeros((1, 8, 3), np.float32); idx = np.arange(8, dtype=np.int32).reshape(1, 8)\n"
            "out = np.zeros((1, 7, 3), np.float32)\n"                                  # one row short
            "p = lambda a: a.ctypes.data_as(ctypes.c_void_p)\n"
            "L.orc_gather_point(1, 8, 8, p(inp), p(idx), p(out))\n"
            "print('survived')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "heap-buffer-overflow" in (r.stdout + r.stderr) and "survived" not in r.stdout, (r.stdout + r.stderr)[-2000:]
