#!/usr/bin/env python3
"""Soak check (manual, through gpurun; not collected by pytest): N BAM-like records with random substitutions, insertions, deletions and
soft clips on both strands, 300 .. 6000 reference bases each, through the whole device chain from raw signal -- event detection, MoM
scalings, event alignment, recalibration, work items, 2 x profile_hmm_score per CpG group, and the eventalign segment chain -- against
the REFERENCE ITSELF (oracle/_ref/libnp_ref_full.so: SquiggleRead::load_from_raw, calculate_methylation_for_read, align_read_to_ref),
every site and every row.  Prints one JSON line; exit status 1 on any mismatch.
    python tests/gpu_soak.py [--reads 1500] [--seed 1]"""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--adc", type=int, default=0, help="1: the traces as int16 ADC counts (the reference sees what they convert to): the counts-in form of the detector")
    args = ap.parse_args()
    from oracle import load_models
    from oracle.ref_full import FullRef
    from nanopolish_amd import api
    from nanopolish_amd.api import Context
    from nanopolish_amd.hostinfo import usable_cores
    from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch
    from nanopolish_amd.synth import synth_cigar_read
    models = load_models()
    nuc = models["nucleotide"]
    BASES = np.frombuffer(b"ACGT", np.uint8)
    g = np.random.default_rng(977 + args.seed).integers(0, 4, 60000)
    contig = BASES[g].tobytes().decode()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    recs = []
    for rid in range(args.reads):
        rd = synth_cigar_read(50000 + 7919 * args.seed + rid, g, nuc, span=int(rng.integers(300, 6000)), p_sub=float(rng.choice([0.0, 0.02, 0.06])),
                              p_ins=float(rng.choice([0.0, 0.02, 0.05])), p_del=float(rng.choice([0.0, 0.02, 0.05])),
                              max_indel=int(rng.integers(1, 12)), soft_clip=(0, int(rng.integers(0, 40))))
        rec = dict(seq=rd["seq"], raw=rd["raw"], rc=rd["rc"], pos=rd["pos"], cigar=api.cigar_words(rd["cigar_ops"]), bam_seq=rd["bam_seq"])
        if args.adc:
            from nanopolish_amd.synth import adc_quantise
            rec["adc"], rec["raw"] = adc_quantise(rd["raw"])
        recs.append(rec)
    t_synth = time.time() - t0
    ctx = Context(0); ctx.register_model(nuc, "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    hb = build_host_batch_records(models, recs, contig)
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=True)
    t0 = time.time()
    batch.step()
    ea = batch.eventalign()
    groups = [batch.groups_of(i) for i in range(len(recs))]
    t_gpu = time.time() - t0
    F = FullRef()

    def check(i):
        r = recs[i]
        fr = F.read("r%d" % i, r["seq"], r["raw"])
        bad_s = bad_r = 0; n_sites = n_rows = 0
        try:
            want = fr.call_methylation(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig, modbam=False, cap=8192) if fr.n_events else None
            first, nm, u, m = groups[i]
            got = {int(f) + r["pos"]: (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
            exp = {} if want is None else {int(s): (float(a), float(b)) for s, a, b in zip(want["start"], want["ll_unmeth"], want["ll_meth"])}
            # (the reference holds the two log-likelihoods as doubles converted from the float scores)
            bad_s = int(got != exp); n_sites = len(exp)
            rows = fr.eventalign(r["rc"], r["pos"], r["cigar"], r["bam_seq"], contig) if fr.n_events else None
            e = ea[i]
            if rows is None:
                bad_r = int(len(e["event_idx"]) != 0)
            else:
                bad_r = int(not (e["status"] == 0 and np.array_equal(rows["ref_position"], e["ref_position"]) and
                                 np.array_equal(rows["event_idx"], e["event_idx"]) and np.array_equal(rows["hmm_state"], e["hmm_state"])))
                n_rows = len(rows["event_idx"])
        finally:
            fr.close()
        return bad_s, bad_r, n_sites, n_rows

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=usable_cores()[2]) as ex:
        res = list(ex.map(check, range(len(recs))))
    t_cpu = time.time() - t0
    bs, br = sum(r[0] for r in res), sum(r[1] for r in res)
    out = dict(reads=len(recs), seed=args.seed, sites_checked=sum(r[2] for r in res), rows_checked=sum(r[3] for r in res),
               reads_with_site_mismatch=bs, reads_with_row_mismatch=br, reads_without_events=sum(1 for r in res if r[2] == 0 and r[3] == 0),
               serial_path_reads=int(ctx.get_stat("ed_serial_reads")), seconds=dict(synth=round(t_synth, 1), gpu=round(t_gpu, 2), reference=round(t_cpu, 1)))
    print(json.dumps(out), flush=True)
    sys.exit(1 if (bs or br) else 0)


if __name__ == "__main__":
    main()
