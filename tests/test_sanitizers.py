"""The host thread team of the transfer engine (csrc/host_pool.h) under ThreadSanitizer and ASan + UBSan: the
stand-alone stress of tools/sanitize/pool_stress.cpp (the full recipe, which also drives the oracle under ASan, is
tools/sanitize.sh)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags,name", [("-fsanitize=thread", "tsan"), ("-fsanitize=address,undefined -fno-sanitize-recover=all", "asan")])
def test_host_pool_under_sanitizer(tmp_path, flags, name):
    exe = str(tmp_path / f"pool_{name}")
    src = os.path.join(ROOT, "tools", "sanitize", "pool_stress.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", *flags.split(), "-pthread", src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "pool stress ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
