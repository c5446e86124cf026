#!/usr/bin/env python3
"""Experiment (round 4): can the memory-bound glue of step i - 1 (work items, event map, recalibration, window bounds: ~10 % of a step) run
BESIDE the event aligner of step i?  The aligner is a persistent kernel that fills every wave slot (8 per SIMD at 64 registers), so a
second stream's kernels only start in its tail; with its grid cut to 7 (6) waves per SIMD (`align_blocks_per_cu`) a slot stays free.
Two batch objects on two contexts alternate:   stream A: align(b_i)        stream B: work items + glue + scoring of b_(i-1)
(the scoring kernels need 256 registers per SIMD and wait for the aligner's waves to retire: they run after it).
Prints ms per step for the in-order pass and for the overlapped one at each grid size, and a CRC of the scores of both.
    python tools/overlap_probe.py [--pool 4000 --tile 10 --steps 6]"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pool", type=int, default=4000); ap.add_argument("--tile", type=int, default=10)
    ap.add_argument("--read-len", type=int, default=5450); ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import torch
    import bench
    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    models = bench.load_models()
    hb = tile_host_batch(bench.prep_host_batch(models, 0, args.pool, args.read_len, False, 8), args.tile)
    bs = []
    for _ in range(2):
        ctx = Context(0)
        ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
        ctx.set_option("stream_switch_wait", 0)
        bs.append(CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, jobs_on_device=True))
    p = lambda t: C.c_void_p(t.data_ptr())

    def align_only(b):
        L, h, s = b.ctx.L, b.ctx.h, C.c_void_p(b.stream) if b.stream else None
        b.ctx._chk(L.np_event_align_dev(h, s, b.n_reads, p(b.d_reads_a), p(b.d_events), p(b.d_ranks), b.m_nuc, b.max_bands, p(b.d_pair_off),
                                        p(b.d_pairs), p(b.d_pair_begin), p(b.d_n_pairs)), "np_event_align_dev")

    def rest(b):
        L, h, s = b.ctx.L, b.ctx.h, C.c_void_p(b.stream) if b.stream else None
        b._step_work_items(L, h, p, s, False); b._step_glue(L, h, p, s, False, b.n_jobs); b._step_hmm(L, h, p, s, False)

    def crc():
        return "%08x" % zlib.crc32(bs[0].scores().tobytes() + bs[1].scores().tobytes())

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    out = []
    for bpc in (8, 7, 6):
        for b in bs:
            b.ctx.set_option("align_blocks_per_cu", bpc)
        # in order, one stream
        for b in bs:
            b.stream = 0; b.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            bs[i % 2].step()
        torch.cuda.synchronize()
        t_seq = (time.perf_counter() - t0) / args.steps * 1e3
        c_seq = crc()
        # overlapped
        e_align = [torch.cuda.Event() for _ in bs]; e_rest = [torch.cuda.Event() for _ in bs]
        have_rest = [False, False]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps + 1):
            if i < args.steps:
                j = i % 2
                if have_rest[j]:
                    sa.wait_event(e_rest[j])
                bs[j].stream = sa.cuda_stream; align_only(bs[j]); e_align[j].record(sa)
            if i >= 1:
                j = (i - 1) % 2
                sb.wait_event(e_align[j])
                bs[j].stream = sb.cuda_stream; rest(bs[j]); e_rest[j].record(sb); have_rest[j] = True
        torch.cuda.synchronize()
        t_ovl = (time.perf_counter() - t0) / args.steps * 1e3
        for b in bs:
            b.stream = 0
        out.append(dict(align_waves_per_simd=bpc, ms_per_step_in_order=round(t_seq, 2), ms_per_step_overlapped=round(t_ovl, 2), crc_in_order=c_seq, crc_overlapped=crc()))
        print(json.dumps(out[-1]), flush=True)


main()
