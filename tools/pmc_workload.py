#!/usr/bin/env python3
"""The workload the round-3 counter passes run under `rocprofv3 --pmc` (profiles/collect_r03_pmc.sh): the SHIPPED kernels at a launch
size small enough for a counter pass to finish -- the call-methylation step (kernel A `np_event_align_kernel`, glue, kernel B
`np_hmm_forward_kernel`) over `--reads` reads, then the eventalign step (`np_eventalign_chain_kernel`) over `--ea-reads` reads.
Prints one JSON line with the UNITS every launch processed (reads, bands, events, HMM calls and cell-states, chain segments and lattice
cells), so that profiles/pmc_summary_r03.py can turn counter totals into per-band / per-call / per-segment figures.

    python tools/pmc_workload.py [--reads 2048] [--ea-reads 2048] [--reps 2]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--ea-reads", type=int, default=2048)
    ap.add_argument("--distinct", type=int, default=256)
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--timing-reps", type=int, default=0, help="> 0 (the run WITHOUT counters): one untimed warm-up step, then this many timed ones; `reps` in the output stays the counter passes' count")
    ap.add_argument("--var-tile", type=int, default=0, help="> 0: ONLY the variants screening step (tests/bench_variants.py at this tile; configs[3] shape)")
    args = ap.parse_args()
    if args.var_tile > 0:
        # configs[3]'s forward launches alone (their own process: the kernel names are the call-methylation step's)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_variants
        va = bench_variants.run(tile=args.var_tile, steps=args.timing_reps or args.reps, warmup=1 if args.timing_reps else 0, cpu_sample=0)
        print(json.dumps(dict(reps=args.reps, variants=dict(calls=va["calls_per_step"], algo_bytes=va["roofline"]["algo_bytes_per_launch"],
                                                            unprofiled_ms=va["hmm_kernel_ms_per_step"]))), flush=True)
        return
    import torch
    import bench
    from nanopolish_amd import api
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch, build_host_batch_records, JOB_DT
    from nanopolish_amd.synth import synth_raw
    models = bench.load_models()
    ctx = Context(0)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    out = dict(reps=args.reps)
    if args.reads > 0:
        pool = min(args.distinct, args.reads)
        hb = bench.prep_host_batch(models, 0, pool, args.read_len, False, 8)
        b = CallMethylationBatch(ctx, tile_host_batch(hb, max(1, args.reads // pool)), "cuda:0", calibrate=True, jobs_on_device=True)
        n_t = args.timing_reps or args.reps
        if args.timing_reps:
            b.step(); ctx.sync()
            for w in range(7):
                ctx.kernel_time(w, reset=True)
        for _ in range(n_t):
            b.step()
        ctx.sync(); torch.cuda.synchronize()
        jh = b.jobs_host()
        live = ((jh["flags"] & 0x80000000) == 0) & (jh["n_kmers"] > 0)       # (slots past a read's groups are not written under np_set_job_layout)
        e = np.abs(jh["e_stop"].astype(np.int64) - jh["e_start"].astype(np.int64)) + 1
        n = jh["n_kmers"].astype(np.int64)
        ms = {k: ctx.kernel_time(w)[0] / n_t for k, w in (("event_align", 0), ("hmm_forward", 1), ("glue", 2))}
        out["call_methylation"] = dict(reads=b.n_reads, bands=int(b.band_cells // 100), events=int(b.total_events),
                                       algo_bytes_align=int(b.algo_bytes_align), hmm_calls=int(live.sum()),
                                       hmm_cell_states=int((3 * e[live] * n[live]).sum()), hmm_algo_bytes=int((4 * e[live] + 14 * n[live] + 4).sum()),
                                       unprofiled_ms={k: round(v, 3) for k, v in ms.items()})
        del b
        torch.cuda.empty_cache()
    if args.ea_reads > 0:
        pool = min(args.distinct, args.ea_reads)
        recs = []
        for rid in range(pool):
            rd = synth_raw(rid, models["nucleotide"], L=args.read_len)
            ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
            recs.append(dict(seq=rd["seq"], raw=rd["raw"], rc=rd["rc"], pos=0, cigar=api.cigar_words([("M", len(rd["seq"]))]), contig=ref))
        hbr = build_host_batch_records(models, recs, "")
        b = CallMethylationBatch(ctx, tile_host_batch(hbr, max(1, args.ea_reads // pool)), "cuda:0", calibrate=True, from_raw=True, workload="eventalign")
        n_t = args.timing_reps or args.reps
        if args.timing_reps:
            b.step(); ctx.sync()
        for w in range(7):
            ctx.kernel_time(w, reset=True)
        for _ in range(n_t):
            b.step()
        ctx.sync(); torch.cuda.synchronize()
        cells, erows, kmers = (ctx.get_stat("ea_lattice_" + q) for q in ("cells", "rows", "kmers"))
        res = b.eventalign_results()
        out["eventalign"] = dict(reads=b.n_reads, segments=int(sum(r["n_calls"] for r in res)), lattice_cells=int(cells), lattice_rows=int(erows),
                                 lattice_kmers=int(kmers), rows_out=int(sum(len(r["event_idx"]) for r in res)),
                                 raw_samples=int(b.hb["raw_off"][-1]) if "raw_off" in b.hb else None,
                                 unprofiled_chain_ms=round(ctx.kernel_time(6)[0] / n_t, 3),
                                 unprofiled_ms=dict(event_detect_family=round(ctx.kernel_time(4)[0] / n_t, 3), mom_fill=round(ctx.kernel_time(5)[0] / n_t, 3)))
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
