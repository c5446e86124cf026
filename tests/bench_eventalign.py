#!/usr/bin/env python3
"""BASELINE.json configs[2] (eventalign) on one MI355X: reads/s of the whole realignment of a batch of synthetic R9.4 reads,
raw signal (int16 ADC counts) resident in HBM -> scrappie event detection -> MoM scalings -> adaptive banded event alignment -> event map +
recalibration -> align_read_to_ref's segment chain (np_eventalign_dev), beside the reference's own code on the host cores
(SquiggleRead from raw + align_read_to_ref, oracle/_ref/libnp_ref_full.so, one read per thread).  Prints one JSON line.
This is a measurement tool for DESIGN.md / profiles/, not the driver's bench (bench.py keeps the call-methylation metric); it
lives under tests/ because its CPU leg runs the oracle.

    python tests/bench_eventalign.py [--pool 5000] [--tile 10] [--read-len 5450] [--steps 3] [--cpu-sample 512]
(also reachable as `python bench.py --workload eventalign`)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT)


GENOME = 5_000_000            # BASELINE.json configs[2]: "50k synthetic reads against 5 Mb reference"
CIGAR_MIX = dict(p_sub=0.02, p_ins=0.015, p_del=0.015, max_indel=4, soft_clip=(0, 12))      # synth_cigar_read's defaults, spelled out


def run(pool=5000, tile=10, read_len=5450, steps=3, warmup=1, cpu_sample=-1, ctx=None):
    """One eventalign measurement (BASELINE.json configs[2], literally since round 4): `pool` distinct reads drawn at uniform origins from
    a seeded 5 Mb genome (resident in HBM once), on both strands, each with substitutions, insertions, deletions and soft clips and the
    BAM record an aligner would report for it (nanopolish_amd/synth.py:synth_cigar_read); pool x tile = 50 000 reads per step.  Returns
    the JSON-able dict.  ctx: a Context with the nucleotide and cpg models registered (bench.py's), else one is created."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import load_models
    from nanopolish_amd import api
    from nanopolish_amd.api import Context
    from nanopolish_amd.hostinfo import usable_cores
    from nanopolish_amd.pipeline import build_host_batch_records, tile_host_batch, CallMethylationBatch
    from nanopolish_amd.synth import synth_cigar_read, adc_quantise, BASES
    models = load_models()
    own = ctx is None
    if own:
        ctx = Context(0)
        ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    t_prep = time.perf_counter()
    genome = np.random.default_rng(0x5EED5).integers(0, 4, GENOME)
    contig = BASES[genome].tobytes().decode()

    def make(rid):
        r = synth_cigar_read(rid, genome, models["nucleotide"], span=read_len, **CIGAR_MIX)
        adc, raw = adc_quantise(r["raw"])            # the trace as a sequencer stores it: int16 counts resident, converted on the device
        return dict(seq=r["seq"], raw=raw, adc=adc, rc=int(r["rc"]), pos=int(r["pos"]), cigar=api.cigar_words(r["cigar_ops"]), bam_seq=r["bam_seq"])
    with ThreadPoolExecutor(max(1, usable_cores()[2])) as ex:          # (threads: the HIP runtime is up, a forked pool is not an option)
        recs = list(ex.map(make, range(pool)))
    n_ops = np.array([len(r["cigar"]) for r in recs])
    t_prep = time.perf_counter() - t_prep
    hb = build_host_batch_records(models, recs, contig, with_jobs=False)
    batch = CallMethylationBatch(ctx, tile_host_batch(hb, tile), "cuda:0", calibrate=True, from_raw=True, workload="eventalign")
    for _ in range(warmup):
        batch.step()
    ctx.sync(); torch.cuda.synchronize()
    for w in range(9):
        ctx.kernel_time(w, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.step()
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cells, erows, kmers = (ctx.get_stat("ea_lattice_" + q) for q in ("cells", "rows", "kmers"))
    phase = {q: ctx.get_stat("ea_cycles_" + q) for q in ("geometry", "fill", "backtrack")}
    res = batch.eventalign_results()
    fam = {name: ctx.kernel_time(w)[0] / max(1, steps) for w, name in ((0, "event_align"), (2, "map_calibrate"), (4, "event_detect"), (5, "mom_scalings"), (6, "eventalign_chain"))}
    rows = int(sum(len(r["event_idx"]) for r in res)); calls = int(sum(r["n_calls"] for r in res))
    # HBM roofline of the chain kernel (the leg's dominant kernel).  Algorithmic bytes of one profile_hmm_align as the reference
    # stores it (src/hmm/nanopolish_profile_hmm_r9.cpp:73-204): event means 4 e + k-mer ranks 2 n in, the back-pointer matrix
    # (e + 1) x 3 (n + 2) x 1 B written and walked, 9 B per emitted EventAlignment row out (the fp32 lattice is working storage,
    # not counted -- as the event aligner's band scores are not); summed over the step's segments by the kernel itself
    algo = int(cells + 4 * erows + 2 * kmers + 9 * rows)
    chain_s = fam["eventalign_chain"] * 1e-3
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_lookup
    roof = dict(bound="hbm", kernel="np_eventalign_chain2_kernel" if ctx.get_stat("ea_cycles_fill") > 0 else "np_eventalign_chain_kernel", achieved=round(algo / chain_s / 1e9, 2) if chain_s > 0 else 0.0, peak=8000.0,
                unit="GB/s", frac=round(algo / chain_s / 1e9 / 8000.0, 5) if chain_s > 0 else 0.0, traffic=pmc_lookup.traffic("chain", "segment", calls), algo_bytes_per_launch=algo,
                issue=pmc_lookup.issue("chain", "segment", chain_s * 2.4e9 * 1024 / max(calls, 1)),
                roofline_issue=pmc_lookup.roofline_issue("chain", "segment", chain_s * 2.4e9 * 1024 / max(calls, 1), "np_eventalign_chain2_kernel"),
                avg_launch_ms=round(fam["eventalign_chain"], 3), lattice_cells_per_launch=int(cells), segments_per_launch=calls,
                wave_cycles_by_phase=phase if phase["fill"] > 0 else None,
                limiter="vector-instruction issue of the Viterbi sweep (one wave per read, data-dependent chain of segments)")
    out = dict(metric="eventalign reads/sec", value=round(batch.n_reads * steps / dt, 1), unit="reads/s", n_gpus=1, steps=steps,
               ms_per_step=round(1e3 * dt / steps, 3), reads_per_step=batch.n_reads, rows_per_step=rows, hmm_align_calls_per_step=calls,
               statuses=sorted(set(r["status"] for r in res)), kernel_ms_per_step={k: round(v, 3) for k, v in fam.items()}, roofline=roof,
               config=dict(workload="eventalign from raw signal, 50k synthetic R9.4 reads against a 5 Mb reference (BASELINE.json configs[2])",
                           genome_bases=GENOME, read_span=read_len, distinct_reads=pool, tile=tile, strands="both (odd read ids reverse)",
                           cigar_mix=dict(CIGAR_MIX, soft_clip=list(CIGAR_MIX["soft_clip"])), mean_cigar_ops=round(float(n_ops.mean()), 1),
                           host_prep_s=round(t_prep, 1)))
    # every HBM copy of a read must give the same rows (replication invariance over the whole step: a size-independent property)
    same = True
    for t in range(1, tile):
        for i in range(0, pool, max(1, pool // 64)):
            a, b = res[i], res[t * pool + i]
            same = same and np.array_equal(a["event_idx"], b["event_idx"]) and np.array_equal(a["ref_position"], b["ref_position"])
    out["copies_identical"] = bool(same)
    # CPU: the reference itself, one read per thread (ctypes releases the GIL)
    n_cpu = cpu_sample if cpu_sample >= 0 else 512
    n_cpu = min(n_cpu, pool)
    try:
        from oracle.ref_full import FullRef, have_full
        if n_cpu > 0 and have_full():
            F = FullRef()
            from nanopolish_amd.hostinfo import usable_cores
            threads = usable_cores()[2]            # affinity mask capped by the cgroup CPU quota
            # OpenMP over records inside the reference-backed library; parity: the row count AND a hash of every row (ref_position,
            # event_idx, hmm_state) of every sampled record against the same hash of the device's rows
            from oracle.ref_full import rows_hash
            rows_cpu, hash_cpu, t_cpu = F.many_records(recs[:n_cpu], contig, threads)
            bad = 0
            for i in range(n_cpu):
                g = res[i]
                bad += int(int(rows_cpu[i]) != len(g["event_idx"]) or int(hash_cpu[i]) != rows_hash(g["ref_position"], g["event_idx"], g["hmm_state"]))
            out["cpu_baseline"] = dict(value=round(n_cpu / t_cpu, 2), unit="reads/s", cores=threads, kind="reference",
                                       sample="%d of the same records: SquiggleRead from raw + align_read_to_ref, OpenMP over records" % n_cpu,
                                       rows_match=bool(bad == 0), reads_rows_checked=n_cpu, reads_differing=bad,
                                       rows_checked=int(sum(int(x) for x in rows_cpu)))
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = dict(error=str(e))
    del batch
    torch.cuda.empty_cache()
    if own:
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=5000, help="distinct reads")
    ap.add_argument("--tile", type=int, default=10, help="HBM copies of the pool: pool x tile = 50 000 reads per step (BASELINE.json configs[2])")
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1: 512, 0: skip)")
    args = ap.parse_args()
    print(json.dumps(run(args.pool, args.tile, args.read_len, args.steps, args.warmup, args.cpu_sample)))


if __name__ == "__main__":
    main()
