"""The eventalign chain kernel's burst list spilling to memory (csrc/np_eventalign_kernel.hip, ea_lds): the shipped kernel keeps 176
bursts of walk steps per segment in LDS and only an unusual segment (hundreds of events per 100 bases) overflows that, so the spill path
would be run by one test read.  nanopolish_amd/variants/libnp_hip_smalllist.so is the same library with a list of 8 bursts -- 80 walk
steps at most, where a segment's back-track takes ~250: EVERY segment spills, several times -- and the eventalign parity tests (DNA and
direct RNA, against the reference's own align_read_to_ref compiled in place) run against it in a child process.
`make -C nanopolish_amd/csrc smalllist` (done by __graft_entry__.build()); skipped where the build is missing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nanopolish_amd", "variants", "libnp_hip_smalllist.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(LIB), reason="make -C nanopolish_amd/csrc smalllist")]


def test_eventalign_parity_with_a_list_that_always_spills():
    env = dict(os.environ, NP_HIP_LIB=LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_gpu_eventalign_dropin.py",
                        "tests/test_gpu_rna.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail
