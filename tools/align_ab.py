#!/usr/bin/env python3
"""A/B timing of kernel A builds on the GPU box: for every library given, the event aligner over the same batch of synthetic
reads (kernel time from the library's own HIP events) and a checksum of its output.  Each library runs in its own process
(NP_HIP_LIB).  Usage: python tools/align_ab.py [--pool 2048 --tile 16] lib1.so lib2.so ..."""
import argparse
import json
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import ctypes as C
    import numpy as np
    import torch
    import bench
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    models = bench.load_models()
    lens = bench.ragged_lengths(np.arange(args.pool), args.read_len) if args.ragged else args.read_len
    hb = bench.prep_host_batch(models, 0, args.pool, lens, False, 8)
    ctx = Context(0)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    b = CallMethylationBatch(ctx, tile_host_batch(hb, args.tile), "cuda:0", calibrate=True, jobs_on_device=True)
    L, h = ctx.L, ctx.h
    p = lambda t: C.c_void_p(t.data_ptr())

    def run():
        ctx._chk(L.np_event_align_dev(h, None, b.n_reads, p(b.d_reads_a), p(b.d_events), p(b.d_ranks), b.m_nuc, b.max_bands,
                                      p(b.d_pair_off), p(b.d_pairs), p(b.d_pair_begin), p(b.d_n_pairs)), "align")
    run(); ctx.sync()
    ctx.kernel_time(0, reset=True)
    for _ in range(args.reps):
        run()
    ctx.sync()
    ms, n = ctx.kernel_time(0)
    npairs = b.d_n_pairs.cpu().numpy(); pb = b.d_pair_begin.cpu().numpy()
    crc = zlib.crc32(npairs.tobytes()) ^ zlib.crc32(pb.tobytes())
    for i in range(0, args.pool, max(1, args.pool // 64)):
        crc = zlib.crc32(b.pairs_of(i).tobytes(), crc)
    bands = int(b.band_cells // 100)
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("NP_HIP_LIB", "default")), ms=round(ms / n, 3), reads=b.n_reads,
                          simd_cycles_per_band=round(ms / n * 1e-3 * 2.4e9 * 1024 / bands, 1), ok_reads=int((npairs > 0).sum()), crc="%08x" % crc)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=2048)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--ragged", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for lib in args.libs or [""]:
        env = dict(os.environ)
        if lib:
            env["NP_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--pool", str(args.pool), "--tile", str(args.tile),
                            "--read-len", str(args.read_len), "--ragged", str(args.ragged), "--reps", str(args.reps)],
                           env=env, capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(out[-1] if out else "FAILED %s: %s" % (lib, r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
