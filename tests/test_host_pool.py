"""The host worker pool of the batched binding (nanopolish_amd/csrc/np_pool.h) on the CPU: tests/host_pool_test.cpp compiled with g++ and
run -- every index of a loop exactly once, loops from four submitting threads at once, posted loops with completion callbacks and
drain(), a pool without worker threads -- plain, and under ThreadSanitizer when the toolchain links it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, extra, name):
    exe = str(tmp_path / name)
    cmd = ["g++", "-O1", "-g", "-std=c++11", "-pthread", "-I" + os.path.join(ROOT, "nanopolish_amd", "csrc")] + extra + [os.path.join(ROOT, "tests", "host_pool_test.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return exe if r.returncode == 0 else None, r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_pool_loops_cover_every_index_once(tmp_path):
    exe, err = _build(tmp_path, [], "pool")
    assert exe, err
    for _ in range(3):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "host pool ok" in r.stdout, r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_pool_is_clean_under_thread_sanitizer(tmp_path):
    exe, err = _build(tmp_path, ["-fsanitize=thread"], "pool_tsan")
    if not exe:
        pytest.skip("ThreadSanitizer does not link here: " + err[-200:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host pool ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-2000:]
