#!/usr/bin/env python3
"""BASELINE.json configs[2] (eventalign) on one MI355X: reads/s of the whole realignment of a batch of synthetic R9.4 reads,
raw signal resident in HBM -> scrappie event detection -> MoM scalings -> adaptive banded event alignment -> event map +
recalibration -> align_read_to_ref's segment chain (np_eventalign_dev), beside the reference's own code on the host cores
(SquiggleRead from raw + align_read_to_ref, oracle/_ref/libnp_ref_full.so, one read per thread).  Prints one JSON line.
This is a measurement tool for DESIGN.md / profiles/, not the driver's bench (bench.py keeps the call-methylation metric); it
lives under tests/ because its CPU leg runs the oracle.

    python tests/bench_eventalign.py [--pool 1000] [--tile 50] [--read-len 5450] [--steps 3] [--cpu-sample 64]
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


def run(pool=1000, tile=50, read_len=5450, steps=3, warmup=1, cpu_sample=-1, ctx=None):
    """One eventalign measurement (BASELINE.json configs[2]); returns the JSON-able dict.  ctx: a Context with the nucleotide and cpg
    models registered (bench.py's), else one is created."""
    import torch
    from oracle import load_models
    from nanopolish_amd import api
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import build_host_batch_records, tile_host_batch, CallMethylationBatch
    from nanopolish_amd.synth import synth_raw
    models = load_models()
    own = ctx is None
    if own:
        ctx = Context(0)
        ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    recs = []
    for rid in range(pool):
        rd = synth_raw(rid, models["nucleotide"], L=read_len)
        ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
        recs.append(dict(seq=rd["seq"], raw=rd["raw"], rc=rd["rc"], pos=0, cigar=api.cigar_words([("M", len(rd["seq"]))]), contig=ref))
    hb = build_host_batch_records(models, recs, "")
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
    roof = dict(bound="hbm", kernel="np_eventalign_chain2_kernel" if ctx.get_stat("ea_cycles_fill") > 0 else "np_eventalign_chain_kernel", achieved=round(algo / chain_s / 1e9, 2) if chain_s > 0 else 0.0, peak=8000.0,
                unit="GB/s", frac=round(algo / chain_s / 1e9 / 8000.0, 5) if chain_s > 0 else 0.0, traffic=None, algo_bytes_per_launch=algo,
                avg_launch_ms=round(fam["eventalign_chain"], 3), lattice_cells_per_launch=int(cells), segments_per_launch=calls,
                wave_cycles_by_phase=phase if phase["fill"] > 0 else None,
                limiter="vector-instruction issue of the Viterbi sweep (one wave per read, data-dependent chain of segments)")
    out = dict(metric="eventalign reads/sec", value=round(batch.n_reads * steps / dt, 1), unit="reads/s", n_gpus=1, steps=steps,
               ms_per_step=round(1e3 * dt / steps, 3), reads_per_step=batch.n_reads, rows_per_step=rows, hmm_align_calls_per_step=calls,
               statuses=sorted(set(r["status"] for r in res)), kernel_ms_per_step={k: round(v, 3) for k, v in fam.items()}, roofline=roof,
               config=dict(workload="eventalign from raw signal, synthetic R9.4 reads (BASELINE.json configs[2] shape)", read_len=read_len,
                           distinct_reads=pool, tile=tile))
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
            # timing: OpenMP over reads inside the reference-backed library; parity: the row COUNT of every sampled read, and the rows
            # themselves of a spread of them, one by one
            rows_cpu, t_cpu = F.many_identity(0, [r["seq"] for r in recs[:n_cpu]], [r["raw"] for r in recs[:n_cpu]], [r["rc"] for r in recs[:n_cpu]], threads)
            ok = all(int(rows_cpu[i]) == len(res[i]["event_idx"]) for i in range(n_cpu))
            checked = 0
            for i in sorted(set(np.linspace(0, n_cpu - 1, min(16, n_cpu)).astype(int).tolist())):
                r = recs[i]
                fr = F.read("r%d" % i, r["seq"], r["raw"])
                ea = fr.eventalign(r["rc"], 0, r["cigar"], r["contig"], r["contig"]) if fr.n_events else None
                ok = ok and ((ea is None and len(res[i]["event_idx"]) == 0) or
                             (ea is not None and np.array_equal(ea["ref_position"], res[i]["ref_position"]) and
                              np.array_equal(ea["event_idx"], res[i]["event_idx"]) and np.array_equal(ea["hmm_state"], res[i]["hmm_state"])))
                checked += 1
            out["cpu_baseline"] = dict(value=round(n_cpu / t_cpu, 2), unit="reads/s", cores=threads, kind="reference",
                                       sample="%d of the same reads: SquiggleRead from raw + align_read_to_ref, OpenMP over reads" % n_cpu,
                                       rows_match=bool(ok), reads_row_counts_checked=n_cpu, reads_rows_checked=checked)
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = dict(error=str(e))
    del batch
    torch.cuda.empty_cache()
    if own:
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=1000, help="distinct reads")
    ap.add_argument("--tile", type=int, default=50, help="HBM copies of the pool: pool x tile = 50 000 reads per step (BASELINE.json configs[2])")
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1: 512, 0: skip)")
    args = ap.parse_args()
    print(json.dumps(run(args.pool, args.tile, args.read_len, args.steps, args.warmup, args.cpu_sample)))


if __name__ == "__main__":
    main()
