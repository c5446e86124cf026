"""The reference-side bindings and the host side of the C ABI under sanitizers (VERDICT r3 Missing 6): the drop-in test
files run again, in a child process, against
  * oracle/_ref/libnp_ref_full_batch_asan.so / libnp_ref_dropin_asan.so -- the product's four shims (np_dropin.cpp, np_batch_dropin.cpp,
    np_variants_dropin.cpp, np_eventalign_dropin.cpp) and the harness that drives them compiled with gcc's AddressSanitizer +
    UndefinedBehaviorSanitizer (`make -C oracle asan`; the reference's own objects are not instrumented), and
  * nanopolish_amd/variants/libnp_hip_ubsan.so -- the library with its host code (np_capi.hip, np_host.cpp: packing, size arithmetic,
    the host-buffer entry points) under UBSan in trap mode (`make -C nanopolish_amd/csrc ubsan`): undefined behaviour there kills the child.
Python is not instrumented, so libasan is preloaded; leak checking is off (the interpreter's own allocations).  Both builds travel to the
GPU box prebuilt; the test is skipped where they are missing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = os.path.join(ROOT, "oracle", "_ref", "libnp_ref_full_batch_asan.so")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "libnp_ref_dropin_asan.so")
HIP = os.path.join(ROOT, "nanopolish_amd", "variants", "libnp_hip_ubsan.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (os.path.exists(BATCH) and os.path.exists(DROPIN) and os.path.exists(HIP)),
                                                  reason="sanitizer builds missing (make -C nanopolish_amd/csrc ubsan && make -C oracle asan)")]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("files", [("tests/test_gpu_dropin.py",), ("tests/test_gpu_batch_dropin.py",),
                                   ("tests/test_gpu_variants_dropin.py", "tests/test_gpu_eventalign_dropin.py")])
def test_dropin_tests_are_clean_under_asan_and_ubsan(files):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed here")
    # (with libasan preloaded dlopen() is intercepted and the caller's RUNPATH is lost: torch's lazily loaded libraries need their directory spelled out)
    import importlib.util
    spec = importlib.util.find_spec("torch")
    libdirs = [os.path.join(os.path.dirname(spec.origin), "lib")] if spec and spec.origin else []
    libdirs += ["/opt/rocm/lib"] + [d for d in os.environ.get("LD_LIBRARY_PATH", "").split(":") if d]
    env = dict(os.environ, LD_LIBRARY_PATH=":".join(libdirs), LD_PRELOAD=asan + ":" + ubsan, NP_REF_BATCH_LIB=BATCH, NP_REF_DROPIN_LIB=DROPIN, NP_HIP_LIB=HIP,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:exitcode=99",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + list(files), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout, tail
