#!/usr/bin/env python3
"""calls/s of `profile_hmm_score` through the per-call drop-in (oracle/_ref/libnp_ref_dropin.so: the reference with its hot-path
translation units replaced by nanopolish_amd/csrc/np_dropin.cpp) from 1 thread and from every CPU this process may use, beside the
reference's own function (oracle/_ref/libnp_ref.so) -- the way the reference calls it: from inside an OpenMP loop over reads
(src/nanopolish_scorereads.cpp:388, basemods.cpp:374,382 under src/common/nanopolish_bam_processor.cpp:99).  One JSON line.

    python tests/bench_percall_dropin.py [--reads 64] [--read-len 1500]
"""
import argparse
import ctypes as C
import json
import os
import time
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from nanopolish_amd.hostinfo import usable_cores  # noqa: E402


def jobs_of(orc, models, read_ids, L):
    """the call-methylation work items of synthetic reads, flattened for npref_score_many_reads"""
    from cases import methylation_jobs, synth_read
    mn = orc.model(models["nucleotide"])
    ev, eo, sh, sc, va, epb, job_off, seqs, rcs, e1, e2, st, rc = [], [0], [], [], [], [], [0], [], [], [], [], [], []
    for rid in read_ids:
        rd = synth_read(rid, models["nucleotide"], L=L)
        s, c = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
        pairs = orc.event_align(mn, orc.scalings(s, c, 1.0), rd["events"], rd["ranks"])
        e, mj = methylation_jobs(orc, rd, pairs)
        ev.append(rd["events"]); eo.append(eo[-1] + len(rd["events"]))
        sh.append(rd["shift"]); sc.append(rd["scale"]); va.append(rd["var"]); epb.append(e)
        for j in mj:
            for a, b in ((j["subseq"], j["rc_subseq"]), (j["m_subseq"], j["rc_m_subseq"])):
                seqs.append(a); rcs.append(b); e1.append(j["e1"]); e2.append(j["e2"]); st.append(j["stride"]); rc.append(j["rc"])
        job_off.append(len(seqs))
    return dict(events=np.concatenate(ev), event_off=eo, shift=sh, scale=sc, var=va, epb=epb, job_off=job_off, seqs=seqs, rc_seqs=rcs,
                e_start=e1, e_stop=e2, stride=st, rc=rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--read-len", type=int, default=1500)
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process)
    from oracle import Oracle, RefOracle, load_models
    models = load_models()
    orc = Oracle()
    J = jobs_of(orc, models, range(100, 100 + args.reads), args.read_len)
    n_calls = len(J["seqs"])
    cores = usable_cores()[2]
    dropin_path = os.path.join(ROOT, "oracle", "_ref", "libnp_ref_dropin.so")
    libs = (("reference", RefOracle()), ("dropin", RefOracle(dropin_path)))
    out = dict(metric="profile_hmm_score calls/sec, called per work item from an OpenMP loop over reads", unit="calls/s", calls=n_calls,
               reads=args.reads, read_len=args.read_len, cores=cores)
    want = None
    Ld = C.CDLL(dropin_path)
    Ld.np_dropin_combiner_flush_ns.restype = C.c_long

    def stats():
        r, c = C.c_long(0), C.c_long(0)
        Ld.np_dropin_combiner_stats(C.byref(r), C.byref(c))
        return r.value, c.value, int(Ld.np_dropin_combiner_flush_ns())
    rounds = {}
    for name, lib in libs:
        for th in (1, cores):
            best = 0.0
            s0, t0 = stats(), time.perf_counter()
            for _ in range(3 if name == "dropin" else 1):
                sc = lib.score_many_reads("cpg", J["events"], J["event_off"], J["shift"], J["scale"], J["var"], J["epb"], J["job_off"], J["seqs"],
                                          J["rc_seqs"], J["e_start"], J["e_stop"], J["stride"], J["rc"], 3, th)
                best = max(best, n_calls / lib.last_call_s)
            if name == "dropin":
                s1, wall = stats(), time.perf_counter() - t0
                nr = max(1, s1[0] - s0[0])
                rounds["t%d" % th] = dict(device_rounds=nr, calls_per_round=round((s1[1] - s0[1]) / nr, 2), us_per_round_inside_the_library=round((s1[2] - s0[2]) / nr / 1e3, 1),
                                          us_per_round_wall=round(wall / nr * 1e6, 1))
            if want is None:
                want = sc
            out["%s_t%d" % (name, th)] = round(best, 1)
            out["%s_t%d_equal" % (name, th)] = bool(np.array_equal(sc, want))
    L = C.CDLL(dropin_path)
    r, c = C.c_long(0), C.c_long(0)
    L.np_dropin_combiner_stats(C.byref(r), C.byref(c))
    L.np_dropin_error_count.restype = C.c_long
    out["combiner"] = dict(device_rounds=r.value, calls=c.value, errors=int(L.np_dropin_error_count()), **rounds)
    out["dropin_over_reference_t%d" % cores] = round(out["dropin_t%d" % cores] / out["reference_t%d" % cores], 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
