#!/usr/bin/env python3
"""Randomised comparison of np_detect_events_adc_dev (counts in, events out) with np_adc_to_pa_checked_dev + np_detect_events_checked_dev on the
GPU box: batches of reads with random lengths (around the 2 048-sample switch, multiples of 8 and not, tiny and long), random first-sample
alignments, per-read offsets and units (including ones that drive counts negative or to exact zeros), level-shaped and white signals.
Prints one line per batch and a summary; exit code 1 on any difference.   python tools/fuzz_detect_adc.py [--batches 60 --seed 1]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from nanopolish_amd import lib as _l
    from nanopolish_amd.api import Context
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=60); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = Context(0)
    dev = "cuda:0"
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    prm = _l.DetectorParam(); ctx.L.np_event_detection_params(C.byref(prm), 0)
    rng = np.random.default_rng(a.seed)
    bad = 0; reads = 0; samples = 0; serial = 0
    for b in range(a.batches):
        n_reads = int(rng.integers(1, 40))
        lens = []
        for _ in range(n_reads):
            k = rng.integers(0, 8)
            if k == 0: lens.append(int(rng.integers(1, 40)))
            elif k == 1: lens.append(int(2048 + rng.integers(-3, 4)))
            elif k == 2: lens.append(int(8 * rng.integers(256, 4000)))
            elif k == 3: lens.append(int(rng.integers(60000, 140000)))
            else: lens.append(int(rng.integers(2000, 30000)))
        adcs, offs, units = [], [], []
        for n in lens:
            if rng.random() < 0.8:
                level = np.repeat(rng.normal(520, 80, n // 9 + 1), int(rng.integers(4, 14)))[:n]
                if len(level) < n: level = np.resize(level, n)
                x = level + rng.normal(0, rng.uniform(2, 15), n)
            else:
                x = rng.normal(500, 120, n)
            adc = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
            off = float(rng.choice([10.0, 13.0, 0.0, -3.0, -500.0, 7.5]))
            if rng.random() < 0.15:
                adc[rng.integers(0, n, min(n, 4))] = np.int16(-int(off))           # exact zeros
            if rng.random() < 0.1:
                adc[rng.integers(0, n, min(n, 3))] = np.int16(1 - int(off))        # one count above zero: the bound may fail -> serial path
            adcs.append(adc); offs.append(off); units.append(float(rng.choice([1400.0 / 8192.0, 1467.6 / 8192.0, 0.25, 1.0])))
        raw_off = np.zeros(n_reads + 1, np.int64); raw_off[1:] = np.cumsum(lens)
        pad = int(rng.integers(0, 2)) * 2                                           # the batch's count array itself on either 4-byte phase is not
        adc = np.concatenate(adcs)                                                  # possible (4-byte aligned by contract); reads land on odd and even positions
        d_adc, d_off = up(adc), up(raw_off)
        d_o, d_u = up(np.array(offs, np.float32)), up(np.array(units, np.float32))
        ev_off = np.zeros(n_reads + 1, np.int64); ev_off[1:] = np.cumsum([n // 2 + 2 for n in lens])
        d_ev_off = up(ev_off); cap = int(ev_off[-1]); mx = max(lens); mev = max(n // 2 + 2 for n in lens)
        outs = []
        for one in (False, True):
            d_raw = torch.zeros(len(adc), dtype=torch.float32, device=dev)
            d_tstat = torch.zeros(2 * len(adc) + 16, dtype=torch.float32, device=dev)
            st = torch.zeros(cap, dtype=torch.int32, device=dev); ln = torch.zeros(cap, dtype=torch.float32, device=dev)
            mn = torch.zeros(cap, dtype=torch.float32, device=dev); sd = torch.zeros(cap, dtype=torch.float32, device=dev)
            ne = torch.zeros(n_reads, dtype=torch.int32, device=dev)
            if one:
                ctx._chk(ctx.L.np_detect_events_adc_dev(ctx.h, None, n_reads, p(d_adc), p(d_off), mx, p(d_o), p(d_u), p(d_raw), C.byref(prm), p(d_tstat), p(d_ev_off), mev,
                                                        p(st), p(ln), p(mn), p(sd), p(ne)), "adc")
            else:
                v = torch.zeros(n_reads, dtype=torch.int32, device=dev)
                ctx._chk(ctx.L.np_adc_to_pa_checked_dev(ctx.h, None, n_reads, p(d_adc), p(d_off), mx, p(d_o), p(d_u), p(d_raw), p(v)), "conv")
                ctx._chk(ctx.L.np_detect_events_checked_dev(ctx.h, None, n_reads, p(d_raw), p(d_off), mx, C.byref(prm), p(d_tstat), p(d_ev_off), mev,
                                                            p(st), p(ln), p(mn), p(sd), p(ne), p(v)), "detect")
            ctx.sync()
            outs.append((ne.cpu().numpy(), st.cpu().numpy(), ln.cpu().numpy(), mn.cpu().numpy(), sd.cpu().numpy(), ctx.get_stat("ed_serial_reads")))
        two, one = outs
        diff = 0
        if not np.array_equal(two[0], one[0]) or two[5] != one[5]:
            diff += 1
        for i in range(n_reads):
            k = max(int(two[0][i]), 0); lo, hi = int(ev_off[i]), int(ev_off[i]) + k
            for x, y in zip(one[1:5], two[1:5]):
                if not np.array_equal(x[lo:hi], y[lo:hi], equal_nan=True):
                    diff += 1; break
        bad += diff; reads += n_reads; samples += int(raw_off[-1]); serial += int(two[5])
        print("batch %3d: %2d reads, %8d samples, %6d events, serial %d, differing reads %d" % (b, n_reads, int(raw_off[-1]), int(np.clip(two[0], 0, None).sum()), two[5], diff), flush=True)
    print("TOTAL: %d batches, %d reads, %d samples, serial-path reads %d, differing reads %d" % (a.batches, reads, samples, serial, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
