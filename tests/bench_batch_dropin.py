#!/usr/bin/env python3
"""Reads/s THROUGH the reference-side batched binding (nanopolish_amd/csrc/np_batch_dropin.cpp), i.e. what a nanopolish build with
INTEGRATION.md section 2 applied gets at BamProcessor-like batch sizes:

    python tests/bench_batch_dropin.py [--sizes 512,2048,8192,32768] [--distinct 512] [--read-len 5450]

Per batch size one JSON line: NpBatchPipeline fed int16 ADC counts (`pipelined_adc`, the production configuration) or float pA samples
(`pipelined`), with the reference's own writer behaviour (`pipelined_adc_ref_writer`: the maps cleared on the calling thread), with two
contexts on the one device (`pipelined_adc_2ctx`), and the synchronous np_calculate_methylation_for_batch (`sync`): host wall clock around
the whole loop, from raw signal in host memory to ScoredSite maps handed over, counted and released.  The records are `--distinct` synthetic
R9.4 reads (BASELINE.json configs[1] shape: ~8k events, identity-aligned to a contig made of their own reference strands),
cycled to fill a batch.  Needs oracle/_ref/libnp_ref_full_batch.so (`make -C oracle batch`; it travels to the GPU box prebuilt)
and a GPU."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# OpenMP inside the binding (and inside the reference objects it is linked with) sizes its team from the VISIBLE CPUs; a container
# with a CPU quota below that (the GPU box: 256 visible, 16 granted) would run 256 threads on 16 CPUs.  Must be set before libgomp
# starts its first team.
from nanopolish_amd.hostinfo import usable_cores  # noqa: E402
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores()[2])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="512,2048,8192,32768")
    ap.add_argument("--distinct", type=int, default=512)
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--target-reads", type=int, default=65536, help="timed reads per size (>= 2 batches)")
    ap.add_argument("--skip", default="", help="comma-separated variants to leave out (pipelined, pipelined_adc, pipelined_adc_ref_writer, pipelined_adc_2ctx, sync)")
    args = ap.parse_args()
    from oracle.ref_full import have_batch, bench_batch
    if not have_batch():
        raise SystemExit("oracle/_ref/libnp_ref_full_batch.so is not built")
    import torch  # noqa: F401  (one HIP runtime per process)
    from bench import load_models
    from nanopolish_amd import api
    from nanopolish_amd.synth import synth_raw
    models = load_models()
    t0 = time.perf_counter()
    recs, contig, pos = [], [], 0
    for r in range(args.distinct):
        rd = synth_raw(r, models["nucleotide"], L=args.read_len, k=6, adc=True)       # raw = the pA values the int16 counts convert to
        ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
        # BAM stores the read as it aligns to the forward reference strand: SEQ == the reference segment for an identity alignment
        recs.append(dict(seq=rd["seq"], raw=rd["raw"].astype(np.float32), adc=rd["adc"], rc=int(rd["rc"]), pos=pos,
                         cigar=np.array([(len(ref) << 4) | 0], np.uint32), bam_seq=ref))
        contig.append(ref); pos += len(ref)
    contig = "".join(contig)
    prep = time.perf_counter() - t0
    raw_bytes = float(np.mean([len(r["raw"]) for r in recs])) * 4
    for bs in [int(x) for x in args.sizes.split(",")]:
        nb = max(8, -(-args.target_reads // bs))
        line = dict(metric="call-methylation reads/sec through np_calculate_methylation_for_batch", unit="reads/s", batch_size=bs, batches=nb,
                    distinct_reads=args.distinct, read_len=args.read_len, raw_bytes_per_read=int(raw_bytes))
        from nanopolish_amd.synth import ADC_OFFSET, ADC_UNIT
        # pipelined_adc: the production configuration (int16 samples, maps handed back with recycle()); pipelined: float pA samples;
        # pipelined_adc_ref_writer: the writer stand-in clears the maps on the calling thread, as the reference's batch_func does;
        # pipelined_adc_2ctx: two contexts on this one device (the multi-GPU form; on one GPU the two share its kernels' time)
        for name, pipelined, adc, contexts, consumer in (("pipelined_adc", True, (float(ADC_OFFSET), float(ADC_UNIT)), 0, 1),
                                                          ("pipelined", True, None, 0, 1),
                                                          ("pipelined_adc_ref_writer", True, (float(ADC_OFFSET), float(ADC_UNIT)), 0, 0),
                                                          ("pipelined_adc_2ctx", True, (float(ADC_OFFSET), float(ADC_UNIT)), 2, 1),
                                                          ("pipelined_adc_4ctx", True, (float(ADC_OFFSET), float(ADC_UNIT)), 4, 1),
                                                          ("sync", False, None, 0, 0)):
            if name in args.skip.split(","):
                continue
            sec, sites, bad, hs = bench_batch(recs, contig, bs, nb, warmup=max(7, min(192, 98304 // bs)), pipelined=pipelined, adc=adc, contexts=contexts, consumer=consumer)
            line[name] = dict(value=round(bs * nb / sec, 1), ms_per_batch=round(sec / nb * 1e3, 2), sites_per_read=round(sites / (bs * nb), 2),
                              records_not_ok=bad, h2d_GBps=round(bs * nb * raw_bytes * (0.5 if adc else 1.0) / sec / 1e9, 2))
            if pipelined:
                line[name]["host_ms_per_batch"] = {k: round(v / nb * 1e3, 2) for k, v in hs.items()}
        line["omp_threads"] = int(os.environ["OMP_NUM_THREADS"])
        line["host_prep_s"] = round(prep, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
