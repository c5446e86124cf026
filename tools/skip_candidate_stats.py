#!/usr/bin/env python3
"""VERDICT r5 item 1b, priced with data before building it: how often can kernel A's band step drop the `left` (k-mer skip) candidate?

The elision the verdict proposes is wave-level: a band may skip sl = (float)(left + lp_skip), its conversion, its two compares and their two
carry adds (8 of the 54 vector instructions) only if NO cell of the band takes FROM_L -- the branch is wave-uniform, the cells are lanes.
lp_skip = log(1e-10) = -23.03, so FROM_L wins wherever the emission is worse than that: (x - mu)^2 / (2 sigma^2) > ~22, i.e. an event 6.7
sigma off the k-mer's level -- rare ON the alignment path, the rule off it (a band's 100 cells pair the event with 100 different k-mers).

This script runs the reference's recurrence (src/nanopolish_raw_loader.cpp:179-195,229-289; fp64 numpy, a statistic, not a parity check) on
synthetic R9.4 reads of the bench's shape and counts, over the bands where every cell of the window exists (the FAST phase):
  * cells that take FROM_L, and bands with at least one such cell (= bands that could NOT drop the candidate);
  * the same for the provable form of the test (left + lp_skip < max(diag, up) + emission - slack for every cell, slack = 2 float ulps).
Usage: python tools/skip_candidate_stats.py [--reads 4] [--len 5450]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BW = 100


def band_stats(events, ranks, mean, stdv, shift, scale):
    E, K = len(events), len(ranks)
    epk = E / K
    lp_skip = np.log(1e-10); lp_stay = np.log(1 - 1 / (epk + 1)); lp_step = np.log(1.0 - np.exp(lp_skip) - np.exp(lp_stay)); lp_trim = np.log(0.01)
    mu = scale * mean[ranks] + shift; sd = stdv[ranks]
    cl = np.log(1.0 / np.sqrt(2 * np.pi)) - np.log(sd)
    NEG = -np.inf
    n_bands = E + K + 2
    # band b holds k-mers llk .. llk + 99; cell (e, k) with e = b - 2 - k; index -1 of both axes is the trim row / column
    llk = -1 - BW // 2
    prev = np.full(BW, NEG); prev_llk = llk            # band b - 1
    prev2 = np.full(BW, NEG); prev2_llk = llk          # band b - 2
    out = dict(fast_bands=0, fast_cells=0, from_l_cells=0, bands_with_from_l=0, bands_not_provable=0, near_path_from_l=0)
    for b in range(n_bands):
        ks = llk + np.arange(BW)
        es = b - 2 - ks
        cur = np.full(BW, NEG)
        valid = (ks >= 0) & (ks < K) & (es >= 0) & (es < E)

        def at(band, band_llk, k):
            o = k - band_llk
            ok = (o >= 0) & (o < BW)
            return np.where(ok, band[np.clip(o, 0, BW - 1)], NEG)
        up = at(prev, prev_llk, ks); left = at(prev, prev_llk, ks - 1); diag = at(prev2, prev2_llk, ks - 1)
        kk = np.clip(ks, 0, K - 1); ee = np.clip(es, 0, E - 1)
        a = (events[ee] - mu[kk]) / sd[kk]
        em = cl[kk] - 0.5 * a * a
        s_d = diag + lp_step + em; s_u = up + lp_stay + em; s_l = left + lp_skip
        m = np.maximum(np.maximum(s_d, s_u), s_l)
        cur = np.where(valid, m, NEG)
        # start cell and trim column
        if b == 0:
            cur[ks == -1] = 0.0
        tr = (ks == -1) & (es >= 0) & (es < E)
        cur[tr] = lp_trim * (es[tr] + 1)
        fast = llk >= 0 and llk + BW - 1 < K - 1 and (b - 2 - llk) <= E - 1 and (b - 2 - (llk + BW - 1)) >= 0
        if fast:
            fl = valid & (m == s_l) & np.isfinite(m)
            out["fast_bands"] += 1; out["fast_cells"] += int(valid.sum()); out["from_l_cells"] += int(fl.sum())
            out["bands_with_from_l"] += int(fl.any())
            slack = 2 * np.spacing(np.abs(m).astype(np.float32)).astype(np.float64)
            best_other = np.maximum(s_d, s_u)
            not_prov = valid & ~(s_l < best_other - slack)
            out["bands_not_provable"] += int(not_prov.any())
            c = int(np.nanargmax(np.where(valid, m, NEG)))                 # the band's best cell ~ the alignment path
            out["near_path_from_l"] += int(fl[max(0, c - 5):c + 6].any())
        # Suzuki's rule for band b + 1
        ll, ur = cur[0], cur[BW - 1]
        if ll == NEG and ur == NEG:
            right = (b & 1) == 0
        else:
            right = ll < ur
        prev2, prev2_llk = prev, prev_llk
        prev, prev_llk = cur, llk
        if b >= 1 or True:
            if right:
                llk += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4)
    ap.add_argument("--len", type=int, default=5450)
    args = ap.parse_args()
    from oracle import load_models
    from nanopolish_amd.synth import synth_read
    from nanopolish_amd import api
    models = load_models(); nuc = models["nucleotide"]
    tot = {}
    for rid in range(args.reads):
        rd = synth_read(rid, nuc, L=args.len)
        sh, sc = api.estimate_scalings_using_mom(nuc, rd["ranks"], rd["events"])
        st = band_stats(rd["events"].astype(np.float64), rd["ranks"].astype(np.int64), nuc["level_mean"], nuc["level_stdv"], sh, sc)
        for k, v in st.items():
            tot[k] = tot.get(k, 0) + v
        print("read %d: %s" % (rid, st), flush=True)
    fb = max(1, tot["fast_bands"])
    print("TOTAL fast bands %d; cells taking FROM_L %.2f %%; bands with >= 1 FROM_L cell %.2f %%; bands where the elision is not provable %.2f %%; "
          "bands with a FROM_L cell within 5 k-mers of the band's best cell %.2f %%"
          % (tot["fast_bands"], 100.0 * tot["from_l_cells"] / max(1, tot["fast_cells"]), 100.0 * tot["bands_with_from_l"] / fb,
             100.0 * tot["bands_not_provable"] / fb, 100.0 * tot["near_path_from_l"] / fb))


if __name__ == "__main__":
    main()
