#!/usr/bin/env python3
"""Round-3 experiment: the call-methylation step (BASELINE.json configs[1] shape) three ways --
  fused      : CallMethylationBatch.step, the aligner's fill and back-track in one kernel (round 2's step);
  split      : the same step with np_event_align_split_dev (fill, then back-track, on one stream);
  pipelined  : PipelinedPass -- score(i) on a second stream beside back-track(i+1) + glue(i+1).
Prints one JSON line per configuration with kernel-family times and a checksum of pairs and scores (all three must agree).
    python tools/split_align_bench.py [--pool 20000 --tile 5 --steps 4]"""
import argparse, json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=20000)
    ap.add_argument("--tile", type=int, default=5)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--modes", default="fused,split,pipelined")
    ap.add_argument("--bt-blocks", default="8", help="back-track launch: workgroups per CU (comma list, pipelined mode)")
    ap.add_argument("--prios", default="3:0", help="pipelined mode: comma list of <back-track wave priority>:<forward-kernel wave priority>")
    args = ap.parse_args()
    import bench
    import torch
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch, PipelinedPass
    models = bench.load_models()
    t0 = time.time()
    hb = bench.prep_host_batch(models, 0, args.pool, 5450, False, args.workers)
    thb = tile_host_batch(hb, args.tile)
    print("host prep %.1f s" % (time.time() - t0), file=sys.stderr)
    fam = ((0, "align_or_fill"), (7, "back_track"), (2, "glue"), (1, "hmm"))

    def new_ctx():
        c = Context(0); c.register_model(models["nucleotide"], "nucleotide"); c.register_model(models["cpg"], "cpg")
        return c

    def crc(b):
        b.sync(); torch.cuda.synchronize()
        n = b.d_n_pairs.cpu().numpy(); beg = b.d_pair_begin.cpu().numpy()
        h = zlib.crc32(n.tobytes()); h = zlib.crc32(beg.tobytes(), h)
        for r in range(0, b.n_reads, max(1, b.n_reads // 512)):
            h = zlib.crc32(b.pairs_of(r).tobytes(), h)
        return "%08x" % h, "%08x" % zlib.crc32(b.scores().tobytes())

    for mode in args.modes.split(","):
        if mode in ("fused", "split"):
            c = new_ctx()
            b = CallMethylationBatch(c, thb, "cuda:0", calibrate=True, jobs_on_device=True)
            b.split_align = mode == "split"
            b.step(); b.sync(); torch.cuda.synchronize()
            for w, _ in fam:
                c.kernel_time(w, reset=True)
            t = time.perf_counter()
            for _ in range(args.steps):
                b.step()
            b.sync(); torch.cuda.synchronize()
            dt = time.perf_counter() - t
            out = dict(mode=mode, reads=b.n_reads, ms_per_step=round(1e3 * dt / args.steps, 2), reads_per_s=round(b.n_reads * args.steps / dt, 1),
                       kernel_ms={name: round(c.kernel_time(w)[0] / args.steps, 2) for w, name in fam}, crc=crc(b),
                       scratch_GB=round(c.get_stat("align_scratch_bytes") / 2**30, 2))
            print(json.dumps(out), flush=True)
            del b; c.close(); torch.cuda.empty_cache()
        else:
            for btb, pr in [(int(x), y) for x in args.bt_blocks.split(",") for y in args.prios.split(",")]:
                ctxs = [new_ctx(), new_ctx()]
                for c in ctxs:
                    c.set_option("align_bt_blocks_per_cu", btb)
                    c.set_option("align_bt_prio", int(pr.split(":")[0])); c.set_option("hmm_prio", int(pr.split(":")[1]))
                pp = PipelinedPass(lambda i: CallMethylationBatch(ctxs[i], thb, "cuda:0", calibrate=True, jobs_on_device=True))
                pp.step(); pp.step(); pp.flush(); torch.cuda.synchronize()
                for c in ctxs:
                    for w, _ in fam:
                        c.kernel_time(w, reset=True)
                t = time.perf_counter()
                for _ in range(args.steps):
                    pp.step()
                pp.flush(); torch.cuda.synchronize()
                dt = time.perf_counter() - t
                km = {name: round(sum(c.kernel_time(w)[0] for c in ctxs) / args.steps, 2) for w, name in fam}
                out = dict(mode=mode, bt_blocks_per_cu=btb, prio_bt_hmm=pr, reads=pp.n_reads, ms_per_step=round(1e3 * dt / args.steps, 2),
                           reads_per_s=round(pp.n_reads * args.steps / dt, 1), kernel_ms_overlapping=km, crc=[crc(b) for b in pp.batches])
                print(json.dumps(out), flush=True)
                del pp
                for c in ctxs:
                    c.close()
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
