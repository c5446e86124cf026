#!/usr/bin/env python3
"""A/B timing of kernel B builds on the GPU box: for every library given, the whole call-methylation step over the same batch of
synthetic reads; prints the forward-kernel family time (the library's own HIP events) and a checksum of the scores.  Each library
runs in its own process (NP_HIP_LIB).  Usage: python tools/hmm_ab.py [--pool 4000 --tile 10] lib1.so lib2.so@NP_HMM_KERNEL=1 ...
(`@NAME=VALUE[,NAME=VALUE]` after a library sets environment variables for that run; an empty library name means the default one)"""
import argparse
import json
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    import bench
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    models = bench.load_models()
    hb = bench.prep_host_batch(models, 0, args.pool, args.read_len, bool(args.from_raw), 8)
    ctx = Context(0)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    b = CallMethylationBatch(ctx, tile_host_batch(hb, args.tile), "cuda:0", calibrate=True, jobs_on_device=True, from_raw=bool(args.from_raw))
    b.step(); ctx.sync()
    for w in (0, 1, 2, 4, 5):
        ctx.kernel_time(w, reset=True)
    for _ in range(args.reps):
        b.step()
    ctx.sync(); torch.cuda.synchronize()
    ms = {n: round(ctx.kernel_time(w)[0] / args.reps, 3) for w, n in ((0, "event_align"), (1, "hmm_forward"), (2, "glue"), (4, "event_detect"), (5, "mom_scalings"))}
    out = dict(lib=os.path.basename(os.environ.get("NP_HIP_LIB", "default")), env=os.environ.get("NP_AB_ENV", ""), reads=b.n_reads, ms=ms,
               scores_crc="%08x" % zlib.crc32(b.scores().tobytes()))
    if args.from_raw:          # the detected events themselves: counts, means, starts
        out["events_crc"] = "%08x" % (zlib.crc32(b.d_n_events.cpu().numpy().tobytes()) ^ zlib.crc32(b.d_events.cpu().numpy().tobytes()) ^ zlib.crc32(b.d_ev_start.cpu().numpy().tobytes()))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=4000)
    ap.add_argument("--tile", type=int, default=10)
    ap.add_argument("--read-len", type=int, default=5450)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--from-raw", type=int, default=0, help="1: the step starts from int16 raw signal (event detection on the device)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for lib in args.libs or [""]:
        env = dict(os.environ)
        lib, _, extra = lib.partition("@")
        for kv in filter(None, extra.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        env["NP_AB_ENV"] = extra
        if lib:
            env["NP_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--pool", str(args.pool), "--tile", str(args.tile),
                            "--read-len", str(args.read_len), "--reps", str(args.reps), "--from-raw", str(args.from_raw)], env=env, capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(out[-1] if out else "FAILED %s: %s" % (lib, r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
