"""The libm dependency, repaired instead of reported (VERDICT r4 Missing 6).

The fused pass forms four kinds of per-read constants on the device with a restatement of this image's glibc 2.35 log / exp / logf
(csrc/np_log.h, np_logf.h): the aligner's lp_skip / lp_stay / lp_step / lp_trim (src/nanopolish_raw_loader.cpp:99-108), set4's log(var)
(src/nanopolish_squiggle_read.cpp:38-65), the HMM transitions (src/hmm/nanopolish_profile_hmm_r9.inl:17-76) and profile_hmm_score_set's
log(n).  np_create compares the restatement with the PROCESS's libm; when they differ (or with NP_HOST_CONSTANTS=1) those constants are
computed on the host with the process's own functions and uploaded.  Here:
  * host-constants mode forced on under the normal libm: the whole chain still equals the oracle (the mode itself is right);
  * the same chain in a child process whose libm is one ulp off for half of all arguments (tests/libm_skew.c, LD_PRELOAD; the oracle's C
    port calls the same skewed functions, as a reference built against that libm would): np_create notices, switches mode, and the
    chain equals the oracle bit for bit -- and with the repair forced OFF it does not (the test has teeth)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHAIN = ["tests/test_gpu_events.py::test_pass_from_raw_signal_matches_oracle",
         "tests/test_gpu_parity.py::test_calibrated_pass_matches_oracle",
         "tests/test_gpu_parity.py::test_fused_call_methylation_pass_matches_oracle",
         "tests/test_gpu_parity.py::test_variant_screening_scores_match_oracle",
         "tests/test_gpu_parity.py::test_hmm_score_matches_oracle"]


def _run(env_extra, tests):
    env = dict(os.environ, **env_extra)
    # (-s: np_create's NP_VERBOSE line goes to stderr, which pytest would swallow for a passing test)
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=900)


def test_host_constants_mode_equals_the_oracle_under_the_normal_libm():
    r = _run({"NP_HOST_CONSTANTS": "1", "NP_VERBOSE": "1"}, CHAIN)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "per-read constants on the HOST" in r.stderr + r.stdout


@pytest.fixture(scope="module")
def skew_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("libm") / "libm_skew.so")
    cc = subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-fno-builtin", os.path.join(ROOT, "tests", "libm_skew.c"), "-o", out, "-ldl", "-lm"],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("no C compiler for the skewed libm here: " + cc.stderr[-300:])
    return out


def test_a_different_libm_is_noticed_and_repaired(skew_lib):
    # (the port must be rebuilt? no: liboracle.so calls log / exp / logf through the PLT, the preloaded library comes first)
    r = _run({"LD_PRELOAD": skew_lib, "NP_VERBOSE": "1"}, CHAIN)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "per-read constants on the HOST" in r.stderr + r.stdout and " 0 of 300000" not in r.stderr + r.stdout
    # the same libm with the repair forced off: the device's glibc-2.35 constants no longer match what this "reference" computes
    r = _run({"LD_PRELOAD": skew_lib, "NP_HOST_CONSTANTS": "0"}, CHAIN[:2])
    assert r.returncode != 0, "a skewed libm made no difference: the check above proves nothing"
