#!/usr/bin/env python3
"""Timing experiment: the call-methylation step of P part-batches on P contexts / P HIP streams, enqueued without waiting for
each other, against the same P part-batches on one stream.  Kernel A is bound by vector issue and uses no LDS; kernel B is
bound by its LDS lookups; the glue is memory traffic -- so parts in different phases of the step may share the CUs.
    python tools/overlap_ab.py [--pool 10000 --tile 5 --parts 2 --steps 4]
Prints one JSON line per configuration (reads/s over all parts)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=10000)
    ap.add_argument("--tile", type=int, default=5, help="copies of the pool per PART at parts=2 (reads per round are fixed: 2 * pool * tile)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--parts", type=str, default="1,2,4")
    ap.add_argument("--stages", type=int, default=1)
    args = ap.parse_args()
    import bench
    import torch
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    models = bench.load_models()
    t0 = time.time()
    hb = bench.prep_host_batch(models, 0, args.pool, 5450, False, args.workers)
    print("host prep %.1f s" % (time.time() - t0), file=sys.stderr)
    total_tiles = 2 * args.tile
    ref_scores = None
    for P in [int(x) for x in args.parts.split(",")]:
        tiles = total_tiles // P
        ctxs, batches, streams = [], [], []
        for i in range(P):
            c = Context(0); c.register_model(models["nucleotide"], "nucleotide"); c.register_model(models["cpg"], "cpg")
            b = CallMethylationBatch(c, tile_host_batch(hb, tiles), "cuda:0", calibrate=True, from_raw=False, jobs_on_device=True)
            ctxs.append(c); batches.append(b)
        n_reads = sum(b.n_reads for b in batches)

        def sync_all():
            for b in batches:
                b.sync()
            torch.cuda.synchronize()

        def run(mode, prio=False, skew_ms=0.0):
            sts = [torch.cuda.Stream(priority=(-1 if (prio and i % 2) else 0)) for i in range(P)]
            for i, b in enumerate(batches):
                b.stream = sts[0].cuda_stream if mode == "one" else sts[i].cuda_stream
            for b in batches:        # warm-up round (and the initial skew between the parts)
                b.step()
                if skew_ms:
                    time.sleep(skew_ms * 1e-3)
            if not skew_ms:
                sync_all()
            t = time.perf_counter()
            for _ in range(args.steps):
                for b in batches:
                    b.step()
            sync_all()
            dt = time.perf_counter() - t
            return dict(parts=P, mode=mode, prio=prio, skew_ms=skew_ms, reads_per_round=n_reads, ms_per_round=round(1e3 * dt / args.steps, 2),
                        reads_per_s=round(n_reads * args.steps / dt, 1))

        def run_stages(wa, wb):
            """stage 1 (work items + aligner) of every part on one stream, stage 2 (calibration + scoring) on another: one aligner
            and one scorer in flight at any time.  The parts' contexts leave the ordering to the events recorded here: scoring of a
            part after its alignment, the next alignment of a part after its scoring."""
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            for c in ctxs:
                c.set_option("stream_switch_wait", 0)
            e1 = [torch.cuda.Event() for _ in batches]; e2 = [torch.cuda.Event() for _ in batches]
            started = [False] * len(batches)
            def enqueue():
                for i, b in enumerate(batches):
                    if started[i]:
                        sa.wait_event(e2[i])
                    b.stream = sa.cuda_stream; b.step(stage=1); e1[i].record(sa)
                    sb.wait_event(e1[i])
                    b.stream = sb.cuda_stream; b.step(stage=2); e2[i].record(sb)
                    started[i] = True
            enqueue(); sync_all()
            t = time.perf_counter()
            for _ in range(args.steps):
                enqueue()
            sync_all()
            dt = time.perf_counter() - t
            for c in ctxs:
                c.set_option("stream_switch_wait", 1)
            return dict(parts=P, mode="stages", reads_per_round=n_reads, ms_per_round=round(1e3 * dt / args.steps, 2), reads_per_s=round(n_reads * args.steps / dt, 1))

        print(json.dumps(run("one")), flush=True)
        if P > 1 and args.stages:
            print(json.dumps(run_stages(0, 0)), flush=True)
        if P > 1:
            step_ms = 255.0 * n_reads / 100000.0
            print(json.dumps(run("streams")), flush=True)
            print(json.dumps(run("streams", skew_ms=step_ms / P / 2)), flush=True)
            print(json.dumps(run("streams", skew_ms=step_ms / P)), flush=True)
            print(json.dumps(run("streams", prio=True, skew_ms=step_ms / P)), flush=True)
        sc = np.concatenate([b.scores()[: 2 * 1000] for b in batches[:1]])
        if ref_scores is None:
            ref_scores = sc
        else:
            print(json.dumps(dict(parts=P, scores_equal_first_config=bool(np.array_equal(sc, ref_scores, equal_nan=True)))), flush=True)
        for b in batches:
            b.stream = None
        del batches, b
        for c in ctxs:
            c.close()
        del ctxs, c
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
