#!/usr/bin/env python3
"""BASELINE.json configs[3] (variants --consensus screening) on one MI355X: profile_hmm_score calls per second over candidate
haplotypes.  A 10 kb draft, reads of both strands covering it (event alignment on the device first), and for every screened
position a 22-base window with the base haplotype and its single-base substitutions / insertions / deletion
(src/nanopolish_call_variants.cpp:288-361): one forward pass per (haplotype, read), nucleotide model, PRE|POST clipping, the
window's event bounds resolved on the device from the reads' event maps.  The early-out of Variant scoring is not applied
(kernel benchmark).  Prints one JSON line; a sample of the scores is checked against the reference's own
profile_hmm_score (oracle/_ref) -- which is why this tool lives under tests/.

    python tests/bench_variants.py [--reads 250] [--tile 8] [--stride 1] [--indel-bias 0.9] [--steps 3] [--cpu-sample 20000]
(defaults = BASELINE.json configs[3]: 10 kb window x 2 000 reads, every position, hmm_indel_bias_factor 0.9 as
src/nanopolish_call_variants.cpp:1116 sets it for --consensus; also reachable as `python bench.py --workload variants`)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT)
K = 6
FLANK = 10


def haplotypes(ref, i):
    cs, ce = i - FLANK, i + 1 + FLANK
    base = ref[cs:ce + 1]
    o = i - cs
    seqs = [base]
    for b in "ACGT":
        if b != ref[i]:
            seqs.append(base[:o] + b + base[o + 1:])          # substitution
            seqs.append(base[:o + 1] + b + base[o + 1:])      # insertion
    if ref[i - 1] != ref[i]:
        seqs.append(base[:o] + base[o + 1:])                  # deletion
    return cs, ce, seqs


def run(draft=10000, reads=250, tile=8, stride=1, indel_bias=0.9, steps=3, warmup=1, cpu_sample=400000):
    """One variants-screening measurement (BASELINE.json configs[3]); returns the JSON-able dict."""
    import types
    args = types.SimpleNamespace(draft=draft, reads=reads, tile=tile, stride=stride, indel_bias=indel_bias, steps=steps, warmup=warmup,
                                 cpu_sample=cpu_sample)
    import torch
    from oracle import load_models
    from nanopolish_amd import api, lib as _l
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import READ_DT, JOB_DT, HAF
    from nanopolish_amd.synth import synth_read_from_codes, BASES
    models = load_models()
    nuc = models["nucleotide"]
    ctx = Context(0, indel_bias=args.indel_bias)
    m_nuc = ctx.register_model(nuc, "nucleotide")
    L_ = ctx.L
    rng = np.random.default_rng(4)
    ref_codes = rng.integers(0, 4, args.draft)
    ref = BASES[ref_codes].tobytes().decode()
    reads = [synth_read_from_codes(ref_codes, rid, nuc, rc=bool(rid & 1)) for rid in range(args.reads)]
    n = len(reads)
    event_off = np.zeros(n + 1, np.int64); rank_off = np.zeros(n + 1, np.int64)
    event_off[1:] = np.cumsum([len(r["events"]) for r in reads]); rank_off[1:] = np.cumsum([len(r["ranks"]) for r in reads])
    reads_a = np.zeros(n, READ_DT); reads_b = np.zeros(n, READ_DT)
    for i, r in enumerate(reads):
        sh, sc = api.estimate_scalings_using_mom(nuc, r["ranks"], r["events"])
        for arr, (shift, scale, var) in ((reads_a, (sh, sc, 1.0)), (reads_b, (r["shift"], r["scale"], r["var"]))):
            L_.np_fill_read_host(C.cast(arr[i:i + 1].ctypes.data, C.POINTER(_l.ReadDev)), shift, scale, var, int(event_off[i]),
                                 len(r["events"]), int(rank_off[i]), len(r["ranks"]))
    # work items: (position, haplotype) x read; k-mer ranks per strand are shared by all reads of that strand
    Ld = args.draft
    positions = list(range(40, Ld - 40, args.stride))
    ranks_fwd, ranks_rc, off, ends = [], [], [0], []
    seq_list = []
    for i in positions:
        cs, ce, seqs = haplotypes(ref, i)
        for q in seqs:
            ranks_fwd.append(api.sequence_kmer_ranks("nucleotide", q, None, K, False))
            ranks_rc.append(api.sequence_kmer_ranks("nucleotide", q, None, K, True))
            off.append(off[-1] + len(ranks_fwd[-1])); ends.append((cs, ce)); seq_list.append(q)
    n_seq = len(ends)
    off = np.array(off, np.int64); tot = int(off[-1])
    job_ranks = np.concatenate(ranks_fwd + ranks_rc).astype(np.uint16)          # [forward | reverse-complement]
    ends = np.array(ends, np.int64)
    nk = (off[1:] - off[:-1]).astype(np.uint32)
    jobs = np.zeros(n_seq * n, JOB_DT)
    kpos = np.zeros((n_seq * n, 2), np.int32)
    for ri, r in enumerate(reads):
        sl = slice(ri * n_seq, (ri + 1) * n_seq)
        jobs["rank_off"][sl] = off[:-1] + (tot if r["rc"] else 0)
        jobs["n_kmers"][sl] = nk; jobs["read"][sl] = ri; jobs["flags"][sl] = HAF; jobs["stride"][sl] = 1
        # read-strand k-mer positions of the window ends (identity alignment; flip_k_strand for reverse-strand reads)
        kpos[sl, 0] = (Ld - ends[:, 0] - K) if r["rc"] else ends[:, 0]
        kpos[sl, 1] = (Ld - ends[:, 1] - K) if r["rc"] else ends[:, 1]
    # tile the read set
    T = args.tile
    ne, nr = int(event_off[-1]), int(rank_off[-1])
    events = np.tile(np.concatenate([r["events"] for r in reads]).astype(np.float32), T)
    ranks = np.tile(np.concatenate([r["ranks"] for r in reads]).astype(np.uint16), T)
    ra = np.tile(reads_a, T); rb = np.tile(reads_b, T)
    for a in (ra, rb):
        a["event_off"] += np.repeat(np.arange(T, dtype=np.int64) * ne, n); a["rank_off"] += np.repeat(np.arange(T, dtype=np.int64) * nr, n)
    jt = np.tile(jobs, T); jt["read"] += np.repeat(np.arange(T, dtype=np.uint32) * n, len(jobs)).astype(np.uint32)
    kt = np.tile(kpos, (T, 1))
    N, NJ = n * T, len(jt)
    dev = torch.device("cuda:0")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    d_events, d_ranks, d_ra, d_rb, d_jobs, d_kpos, d_jr = up(events), up(ranks), up(ra), up(rb), up(jt), up(kt), up(job_ranks)
    bands = np.tile((event_off[1:] - event_off[:-1]) + (rank_off[1:] - rank_off[:-1]) + 2, T)
    pair_off = np.zeros(N + 1, np.int64); pair_off[1:] = np.cumsum(bands)
    d_pair_off = up(pair_off)
    d_pairs = torch.empty(int(pair_off[-1]) * 8, dtype=torch.uint8, device=dev)
    d_pb = torch.zeros(N, dtype=torch.int32, device=dev); d_np = torch.zeros(N, dtype=torch.int32, device=dev)
    d_map = torch.empty(len(ranks), dtype=torch.int32, device=dev); d_epb = torch.zeros(N, dtype=torch.float64, device=dev)
    d_scores = torch.zeros(NJ, dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    h = ctx.h
    # once: event alignment + event map + window bounds (the reads' AlignmentDB in the reference)
    ctx._chk(L_.np_event_align_dev(h, None, N, p(d_ra), p(d_events), p(d_ranks), m_nuc, int(bands.max()), p(d_pair_off), p(d_pairs), p(d_pb), p(d_np)), "align")
    ctx._chk(L_.np_resolve_jobs_dev(h, None, N, p(d_rb), p(d_pair_off), p(d_pairs), p(d_pb), p(d_np), p(d_map), p(d_epb), NJ, p(d_jobs), p(d_kpos)), "resolve")

    def step():
        ctx._chk(L_.np_hmm_score_dev(h, None, NJ, p(d_jobs), p(d_rb), p(d_events), p(d_jr), m_nuc, p(d_scores)), "score")
    for _ in range(args.warmup):
        step()
    ctx.sync(); torch.cuda.synchronize()
    ctx.kernel_time(1, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sc = d_scores.cpu().numpy()
    jh = d_jobs.cpu().numpy().view(JOB_DT)
    scored = int(np.sum((jh["flags"] & 0x80000000) == 0))
    # HBM roofline of the forward kernel: algorithmic bytes per call 4 e + 2 n + 12 n + 4 (event means, k-mer ranks, the
    # three scaled-Gaussian floats per k-mer, the score), summed over the scored items of one launch; the lattice never leaves the chip
    live = (jh["flags"] & 0x80000000) == 0
    e_len = np.abs(jh["e_stop"].astype(np.int64) - jh["e_start"].astype(np.int64)) + 1
    algo = int((4 * e_len[live] + 14 * jh["n_kmers"][live].astype(np.int64) + 4).sum())
    cells = int((3 * e_len[live] * jh["n_kmers"][live].astype(np.int64)).sum())
    hmm_ms = ctx.kernel_time(1)[0] / args.steps
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_lookup
    roof = dict(bound="hbm", kernel="np_hmm_forward_kernel", achieved=round(algo / (hmm_ms * 1e-3) / 1e9, 2) if hmm_ms > 0 else 0.0, peak=8000.0,
                unit="GB/s", frac=round(algo / (hmm_ms * 1e-3) / 1e9 / 8000.0, 5) if hmm_ms > 0 else 0.0, traffic=pmc_lookup.traffic("hmm_forward_variants", "call", scored), algo_bytes_per_launch=algo,
                issue=pmc_lookup.issue("hmm_forward_variants", "call", hmm_ms * 1e-3 * 2.4e9 * 1024 / max(scored, 1)),
                roofline_issue=pmc_lookup.roofline_issue("hmm_forward_variants", "call", hmm_ms * 1e-3 * 2.4e9 * 1024 / max(scored, 1), "np_hmm_forward_kernel"),
                avg_launch_ms=round(hmm_ms, 3), cell_states_per_s=round(cells / (hmm_ms * 1e-3) / 1e9, 2) if hmm_ms > 0 else 0.0,
                limiter="vector-instruction issue (see issue: the p7_FLogsum look-ups are 6 vector instructions + one LDS gather each, ~8 per cell, bit-exact); nothing but events, ranks and scores touches HBM")
    out = dict(metric="variants screening profile_hmm_score calls/sec", roofline=roof, value=round(scored * args.steps / dt, 1), unit="calls/s", n_gpus=1,
               steps=args.steps, ms_per_step=round(1e3 * dt / args.steps, 3), calls_per_step=scored, items_per_step=NJ,
               hmm_kernel_ms_per_step=round(hmm_ms, 3),
               config=dict(workload="variants --consensus screening shape (BASELINE.json configs[3]): 22-base windows, base + single-base edits",
                           draft=Ld, reads=N, distinct_reads=n, positions=len(positions), haplotypes=n_seq, indel_bias=args.indel_bias))
    if args.cpu_sample > 0:
        try:
            from oracle import RefOracle, Oracle, have_ref
            pick = np.flatnonzero((jh["flags"][:len(jobs)] & 0x80000000) == 0)
            pick = pick[np.linspace(0, len(pick) - 1, min(args.cpu_sample, len(pick))).astype(np.int64)]
            epb = d_epb.cpu().numpy()
            from nanopolish_amd.hostinfo import usable_cores
            threads = usable_cores()[2]            # affinity mask capped by the cgroup CPU quota
            order = np.argsort(jh["read"][pick], kind="stable"); pick = pick[order]
            rd_of = jh["read"][pick].astype(np.int64)
            job_off = np.searchsorted(rd_of, np.arange(n + 1))
            seqs = [seq_list[j % n_seq] for j in pick]
            rcs = [api.reverse_complement("nucleotide", q) for q in seqs]
            if have_ref():
                refo = RefOracle()
                refo.set_indel_bias(args.indel_bias)
                got = refo.score_many_reads("nucleotide", events[:ne], event_off, [r["shift"] for r in reads], [r["scale"] for r in reads],
                                            [r["var"] for r in reads], epb[:n], job_off, seqs, rcs, jh["e_start"][pick], jh["e_stop"][pick],
                                            jh["stride"][pick], [int(reads[i]["rc"]) for i in rd_of], 3, threads)
                t_cpu = refo.last_call_s
                refo.set_indel_bias(1.0)
                out["cpu_baseline"] = dict(value=round(len(pick) / t_cpu, 1), unit="calls/s", cores=threads, kind="reference",
                                           sample="%d of the same work items, OpenMP over reads" % len(pick),
                                           max_abs_diff=float(np.max(np.abs(got.astype(np.float64) - sc[pick].astype(np.float64)))))
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = dict(error=repr(e))
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draft", type=int, default=10000)
    ap.add_argument("--reads", type=int, default=250, help="distinct reads covering the draft")
    ap.add_argument("--tile", type=int, default=8, help="independent copies of the read set in HBM")
    ap.add_argument("--stride", type=int, default=1, help="screen every stride-th draft position")
    ap.add_argument("--indel-bias", type=float, default=0.9, help="hmm_indel_bias_factor (src/nanopolish_call_variants.cpp:1116)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=400000, help="work items for the CPU baseline / parity check (0: skip)")
    a = ap.parse_args()
    print(json.dumps(run(a.draft, a.reads, a.tile, a.stride, a.indel_bias, a.steps, a.warmup, a.cpu_sample)))


if __name__ == "__main__":
    main()
