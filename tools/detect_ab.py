#!/usr/bin/env python3
"""A/B of the detector's two forms on one box: int16 ADC counts through np_adc_to_pa_checked_dev + np_detect_events_checked_dev against
np_detect_events_adc_dev, the from-raw step's detector family (the library's own HIP events) per 100 000-read step, alternating, and a checksum
of the detected events.  Usage: python tools/detect_ab.py [--pool 2000 --tile 10 --reps 3]"""
import argparse
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=2000); ap.add_argument("--tile", type=int, default=10); ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    models = bench.load_models()
    hb = bench.prep_host_batch(models, 0, a.pool, 5450, True, 8)
    ctx = Context(0)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    th = tile_host_batch(hb, a.tile)
    for rep in range(a.reps):
        for one in (False, True):
            b = CallMethylationBatch(ctx, th, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=True, map_stop=False, adc_one_call=one)
            b.step(); ctx.sync()
            ctx.kernel_time(4, reset=True)
            for _ in range(3):
                b.step()
            ctx.sync()
            ms = ctx.kernel_time(4)[0] / 3
            ne = b.d_n_events.cpu().numpy()
            crc = zlib.crc32(ne.tobytes()) ^ zlib.crc32(b.d_events.cpu().numpy().tobytes()) ^ zlib.crc32(b.d_ev_stdv.cpu().numpy().tobytes())
            print("rep %d  %-9s event_detect %.3f ms per %d reads  (%.2f ms per 100 000)  crc %08x" % (rep, "one call" if one else "two calls", ms, b.n_reads, ms * 1e5 / b.n_reads, crc), flush=True)
            del b
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
