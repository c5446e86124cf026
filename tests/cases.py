"""Shared seeded test cases (TEST side: may use the oracle freely)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nanopolish_amd.synth import synth_read  # noqa: E402,F401
from oracle.workloads import (K, HAF_PRE, HAF_POST, revcomp, methylation_jobs, eventalign_segments,  # noqa: E402,F401
                              call_methylation_read)


def vector_overload_case(orc, rd, pairs):
    """Inputs for profile_hmm_score(sequence, std::vector<HMMInputData>, flags) (src/hmm/nanopolish_profile_hmm.cpp:14-21):
    ONE sequence (the first CpG window of the read) against several "reads" -- the windows of the read's first groups, each
    with its own scalings.  Returns (epb, sequence, rc_sequence, list of data dicts)."""
    epb, jobs = methylation_jobs(orc, rd, pairs)
    js = jobs[:5]
    datas = [dict(events=rd["events"], e_start=j["e1"], e_stop=j["e2"], stride=j["stride"], rc=j["rc"], shift=rd["shift"] + 0.25 * i,
                  scale=rd["scale"] * (1.0 + 0.01 * i), var=rd["var"] + 0.05 * i, events_per_base=epb + 0.1 * i) for i, j in enumerate(js)]
    return epb, js[0]["subseq"], js[0]["rc_subseq"], datas
