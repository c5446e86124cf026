"""Shared seeded test cases (TEST side: may use the oracle freely)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nanopolish_amd.synth import synth_read  # noqa: E402,F401
from oracle.workloads import (K, HAF_PRE, HAF_POST, revcomp, methylation_jobs, eventalign_segments,  # noqa: E402,F401
                              call_methylation_read)
