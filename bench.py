#!/usr/bin/env python3
"""bench.py -- call-methylation reads/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 100 000 synthetic R9.4 reads (BASELINE.json configs[1]: ~8k events
each, r9.4_450bps CpG model; 20 000 distinct reads x 5 copies in HBM):
    [--from-raw 1: scrappie event detection -> MoM scalings ->] work-item generation (motif scan, grouping, window k-mers)
    -> adaptive_banded_simple_event_align -> event map / recalibrate_model / window event bounds -> 2 x profile_hmm_score per
    CpG group
with every stage on the device inside the timed region.  Three timings of the same pass go into the ONE JSON line rank 0
prints:
    value           inputs resident in HBM when the timed region starts (the contract's headline)
    value_streamed  the same batch fed from pinned host memory every step: events, k-mer ranks, read records and reference
                    strands host->device, scores and site metadata device->host, double-buffered against the compute
                    (what BamProcessor's 512-record batches would deliver, src/common/nanopolish_bam_processor.cpp:90-119)
    value_ragged    a batch with log-normal read lengths of the same mean (resident)
Reads shard across ranks with no data-path collective (weak scaling); the only exchange is one all-reduce of the per-site
table at the end of the timed region (N > 1).

One GPU: the line also carries value_from_raw (the step from int16 raw signal), value_eventalign / value_variants (BASELINE.json configs[2] /
configs[3]) and value_binding_512 / value_binding_8192 (reads/s through the reference-side batched binding, host memory to ScoredSite maps), each
with its parity field.  N > 1 GPUs: BASELINE.json configs[4] -- 250 000 reads per rank and step by default (2 M over 8 GPUs).

`--gpus N` without a torch.distributed launcher (no WORLD_SIZE in the environment) starts the N ranks itself.
`--workload eventalign|variants` runs BASELINE.json configs[2] / configs[3] (tests/bench_eventalign.py, tests/bench_variants.py);
`--workload cpu-t1` is configs[0]'s plumbing line: the reference's own code on ONE host thread, no GPU.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The streamed variant runs three HIP streams (upload, compute, read-back) beside torch's and the library's own.  HIP maps
# streams onto GPU_MAX_HW_QUEUES (default 4) in-order hardware queues; when the upload stream shares one with the compute
# stream, the upload of step k+1 queues behind the kernels of step k and nothing overlaps (seen in the copy/kernel trace of
# profiles/r02).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from nanopolish_amd.hostinfo import usable_cores  # noqa: E402

CLOCK_HZ = 2.4e9          # MI355X engine clock (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs
N_SIMD = 1024


def load_models():
    z = np.load(os.path.join(ROOT, "tests", "golden", "models_r9.4_450bps.npz"))
    return {a: dict(k=6, level_mean=z[a + "_level_mean"], level_stdv=z[a + "_level_stdv"],
                    level_log_stdv=z[a + "_level_log_stdv"]) for a in ("nucleotide", "cpg")}


# ---- host preparation (numpy, before the GPU is touched; a pool of forked workers) ---------------------------------------
def ragged_lengths(read_ids, mean_len, sigma=0.5, lo=600, hi=40000):
    """Log-normal read lengths with mean `mean_len` (what nanopore read-length distributions look like), deterministic
    in the read id."""
    z = np.array([np.random.default_rng(0x5EED + int(r)).standard_normal() for r in read_ids])
    L = mean_len * np.exp(sigma * z - 0.5 * sigma * sigma)
    return np.clip(np.rint(L), lo, hi).astype(np.int64)


def _prep_chunk(a):
    from nanopolish_amd.pipeline import build_host_batch
    models, ids, lens, raw = a
    return build_host_batch(models, ids, L=lens, raw=raw, with_jobs=False, adc=raw)       # raw traces are int16 ADC counts


def prep_host_batch(models, lo, hi, lens, raw, workers):
    from nanopolish_amd.pipeline import concat_host_batches
    ids = np.arange(lo, hi)
    lens = np.broadcast_to(np.asarray(lens, np.int64), ids.shape)
    chunk = 1024
    parts = [(models, ids[i:i + chunk], lens[i:i + chunk], raw) for i in range(0, len(ids), chunk)]
    if workers > 1 and len(parts) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(parts))) as pool:
            out = pool.map(_prep_chunk, parts)
    else:
        out = [_prep_chunk(p) for p in parts]
    return concat_host_batches(out)


GENOME_LEN = 5_000_000        # the seeded reference of BASELINE.json configs[2] / the eventalign leg; configs[4]'s reads are placed on it too
_GENOME = {}                  # (codes uint8, contig str) of the forked workers


def bench_genome(n=GENOME_LEN):
    from nanopolish_amd.synth import BASES
    codes = np.random.default_rng(0x5EED5).integers(0, 4, n).astype(np.uint8)
    return codes, BASES[codes].tobytes().decode()


def _prep_record_chunk(a):
    from nanopolish_amd import api
    from nanopolish_amd.pipeline import build_host_batch_records
    from nanopolish_amd.synth import synth_cigar_read_fast
    models, ids, read_len = a
    codes, contig = _GENOME["g"]
    recs = []
    for rid in ids:
        r = synth_cigar_read_fast(int(rid), codes, models["nucleotide"], span=read_len)
        recs.append(dict(seq=r["seq"], events=r["events"], shift=r["shift"], scale=r["scale"], var=r["var"], rc=int(r["rc"]), pos=int(r["pos"]),
                         cigar=api.cigar_words(r["cigar_ops"])))
    hb = build_host_batch_records(models, recs, contig, with_jobs=False)
    for r, q in zip(hb["reads"], recs):
        r["contig"] = None                      # (one contig for all: not pickled once per read)
    hb["genome"] = None
    return hb


def prep_record_batch(models, lo, hi, read_len, workers):
    """The N > 1 batch (BASELINE.json configs[4]): reads drawn at uniform origins from the seeded 5 Mb genome, both strands,
    with substitutions / indels / soft clips and the BAM record an aligner would report (nanopolish_amd/synth.py:synth_cigar_read_fast), from
    pre-detected events like configs[1]'s.  Work items follow the CIGARs on the device (np_cm_build_jobs_cigar_dev) and the per-site table is
    keyed by GENOME position: ranks hold overlapping reads, the all-reduce adds their counts site by site."""
    from nanopolish_amd.pipeline import concat_record_batches
    _GENOME["g"] = bench_genome()
    ids = np.arange(lo, hi) + (1 << 26)                       # its own id range
    chunk = 512
    parts = [(models, ids[i:i + chunk], read_len) for i in range(0, len(ids), chunk)]
    if workers > 1 and len(parts) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(parts))) as pool:
            out = pool.map(_prep_record_chunk, parts)
    else:
        out = [_prep_record_chunk(p) for p in parts]
    g = np.frombuffer(_GENOME["g"][1].encode(), np.uint8).copy()
    for o in out:
        o["genome"] = g
    hb = concat_record_batches(out)
    hb["contig"] = _GENOME["g"][1]
    return hb


def record_sample_parity(models, hb, batch, rank, n_sample=12):
    """Genome mode: a sample of this rank's reads through the oracle's restatement of the reference's per-read pass ON THE SAME RECORD
    (oracle/workloads.py:call_methylation_record -- CIGAR walk, event alignment, recalibration, calculate_methylation_for_read's work items and
    scores) against the GPU batch: scored sites (genome position, n_motif) identical, LLRs compared."""
    from oracle import Oracle
    from oracle.workloads import call_methylation_record
    orc = Oracle()
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    n = len(hb["reads"])
    pick = sorted(set(np.linspace(0, n - 1, min(n, n_sample)).astype(int).tolist()))
    bad_sites, groups, missing, d = 0, 0, 0, []
    for i in pick:
        r = hb["reads"][i]
        want = call_methylation_record(orc, mn, mc, r["seq"], None, r["rc"], r["pos"], r["cigar"], hb["contig"], events=r["events"])
        first, nm, u, m = batch.groups_of(i)
        keep = np.isfinite(u)
        got = {(int(f) + r["pos"], int(k)): (float(uu), float(mm)) for f, k, uu, mm in zip(first[keep], nm[keep], u[keep], m[keep])}
        exp = {(s_["start"], s_["n_motif"]): (s_["ll_unmeth"], s_["ll_meth"]) for s_ in want["sites"]}
        groups += len(exp)
        if set(got) != set(exp):
            bad_sites += 1
        for key, (eu, em) in exp.items():
            if key not in got:
                missing += 1
            else:
                d.append((got[key][1] - got[key][0]) - (em - eu))
    return dict(rank=rank, reads=len(pick), reads_pairs_differ=bad_sites, groups=groups, groups_missing_on_gpu=missing,
                max_abs_dLLR=float(np.max(np.abs(d))) if d else None,
                what="reads_pairs_differ counts reads whose SET of scored sites (genome start, n_motif) differs from the oracle's")


# ---- CPU baseline ------------------------------------------------------------------------------------------------------
def cpu_pass(models, hb, idx, thread_list, calibrate, from_raw, repeats=2):
    """One pass of the hot path on the host over reads hb["reads"][idx]: align + 2 x score per group, OpenMP over reads like
    src/common/nanopolish_bam_processor.cpp:99, timed once per entry of thread_list.  The reference's own code when
    oracle/_ref/libnp_ref.so travelled with the repo (kind="reference"), else the oracle port (kind="port").  Returns the
    timings per thread count and the results (the parity check of the GPU numbers)."""
    from oracle import Oracle, RefOracle, have_ref
    from oracle.workloads import methylation_jobs, K
    orc = Oracle()
    ref = RefOracle() if have_ref() else None
    if ref is not None and hasattr(ref.L, "npref_tune_malloc"):
        ref.L.npref_tune_malloc()          # allocator settings of the harness, see oracle/ref_harness.cpp
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    rds = [hb["reads"][i] for i in idx]
    n = len(rds)
    T = {th: dict(detect=0.0, align=1e30, score=1e30, calib=0.0) for th in thread_list}
    mom = np.array([hb["mom"][i] for i in idx]).reshape(-1, 2).copy()
    if from_raw:
        raw = np.concatenate([r["raw"] for r in rds]).astype(np.float32)
        wo = np.zeros(n + 1, np.int64); wo[1:] = np.cumsum([len(r["raw"]) for r in rds])
        for th in thread_list:
            T[th]["detect"] = 1e30
            for _ in range(repeats):
                evm, evo, evn = (ref or orc).detect_events_many(raw, wo, th)
                T[th]["detect"] = min(T[th]["detect"], (ref or orc).last_call_s)
        rds = [dict(r, events=evm[evo[i]:evo[i] + evn[i]].copy()) for i, r in enumerate(rds)]
        for i, r in enumerate(rds):
            mom[i] = orc.estimate_scalings_mom(mn, r["ranks"], r["events"])
    eo = np.zeros(n + 1, np.int64); eo[1:] = np.cumsum([len(r["events"]) for r in rds])
    ro = np.zeros(n + 1, np.int64); ro[1:] = np.cumsum([len(r["ranks"]) for r in rds])
    ev = np.concatenate([r["events"] for r in rds]).astype(np.float32)
    rk = np.concatenate([r["ranks"] for r in rds]).astype(np.uint32)
    # each leg runs `repeats` times per thread count (the first call also pays thread start-up and page faults); the faster
    # run counts, and only the C call itself is timed (oracle_py.last_call_s), not the ctypes marshalling around it
    for th in thread_list:
        for _ in range(repeats):
            if ref:
                pairs, pair_off, n_pairs = ref.align_many([r["seq"] for r in rds], ev, eo, mom[:, 0], mom[:, 1], th)
                T[th]["align"] = min(T[th]["align"], ref.last_call_s)
            else:
                pairs, pair_off, n_pairs = orc.align_many(mn, ev, eo, rk, ro, mom[:, 0], mom[:, 1], th)
                T[th]["align"] = min(T[th]["align"], orc.last_call_s)
    # event map + window bounds through the oracle's glue (untimed host bookkeeping)
    job_read, e1, e2, stride, rcs, jr, jr_off, epb = [], [], [], [], [], [], [0], np.zeros(n)
    seqs, rc_seqs, first, job_off = [], [], [], [0]
    sh = [r["shift"] for r in rds]; sc_ = [r["scale"] for r in rds]; vr = [r["var"] for r in rds]
    cals = [None] * n
    t_calib = {th: 0.0 for th in thread_list}
    if calibrate:
        # recalibrate_model on the event map (oracle restatement, one C call per read): run on `th` host threads for real -- ctypes
        # releases the GIL -- and timed by the wall clock, like the other legs (round 2 divided a serial time by the thread count)
        from concurrent.futures import ThreadPoolExecutor

        def _cal(i):
            p = pairs[pair_off[i]:pair_off[i] + n_pairs[i]]
            if not len(p):
                return None
            start, stop, _ = orc.build_base_to_event_map(p, len(rds[i]["ranks"]))
            return orc.recalibrate(mn, rds[i]["events"], rds[i]["ranks"], start, stop)
        for th in thread_list:
            tc = time.perf_counter()
            if th > 1:
                with ThreadPoolExecutor(th) as ex:
                    cals = list(ex.map(_cal, range(n)))
            else:
                cals = [_cal(i) for i in range(n)]
            t_calib[th] = time.perf_counter() - tc
    for i in range(n):
        p = pairs[pair_off[i]:pair_off[i] + n_pairs[i]]
        if len(p):
            epb[i], jobs = methylation_jobs(orc, rds[i], p)
            if calibrate:
                cal = cals[i]
                if cal is None or cal[2] > 2.5:
                    jobs = []
                else:
                    sh[i], sc_[i], vr[i] = cal
            for j in (jobs if epb[i] <= 5.0 else []):
                for s, r in ((j["subseq"], j["rc_subseq"]), (j["m_subseq"], j["rc_m_subseq"])):
                    job_read.append(i); e1.append(j["e1"]); e2.append(j["e2"]); stride.append(j["stride"]); rcs.append(j["rc"])
                    seqs.append(s); rc_seqs.append(r)
                    if not ref:
                        q = orc.sequence_kmer_ranks("cpg", s, r, K, j["rc"])
                        jr.append(q); jr_off.append(jr_off[-1] + len(q))
                first.append((i, j["first"]))
        job_off.append(len(seqs))
    for th in thread_list:
        T[th]["calib"] = t_calib[th]
        for _ in range(repeats):
            if ref:
                sc = ref.score_many_reads("cpg", ev, eo, sh, sc_, vr, epb, job_off, seqs, rc_seqs, e1, e2, stride, rcs, 3, th)
                T[th]["score"] = min(T[th]["score"], ref.last_call_s)
            else:
                sc = orc.score_many(mc, job_read, ev, eo, sh, sc_, vr, epb, np.concatenate(jr), jr_off, e1, e2, stride, 1.0, 3, th)
                T[th]["score"] = min(T[th]["score"], orc.last_call_s)
    for th in thread_list:
        T[th]["seconds"] = T[th]["detect"] + T[th]["align"] + T[th]["calib"] + T[th]["score"]
    return dict(n=n, timings=T, n_events=[len(r["events"]) for r in rds], pairs=(pairs, pair_off, n_pairs),
                first=first, scores=sc, kind="reference" if ref else "port")


def cpu_baseline(models, hb, calibrate, from_raw, budget_reads):
    """The CPU line: one thread first (t1), then the same reads-per-thread load (>= 32 reads per thread) on every available
    hardware thread and on half of them (SMT siblings share a core's FP units); the best multi-thread rate is `value`."""
    visible, quota, cores = usable_cores()
    pool = len(hb["reads"])
    n1 = min(pool, 24)
    c1 = cpu_pass(models, hb, list(range(n1)), [1], calibrate, from_raw, repeats=1)
    t1 = c1["timings"][1]
    out = dict(t1_value=round(n1 / t1["seconds"], 2), t1_reads=n1, unit="reads/s", kind=c1["kind"])
    out.update(visible_cores=visible, cgroup_cpu_quota=quota)
    cands = sorted({c for c in (cores, cores // 2, cores // 4) if c >= 2}, reverse=True)
    if not cands:
        out.update(value=out["t1_value"], cores=cores, threads=1, per_core=out["t1_value"], sample="%d reads on one thread" % n1)
        return out, c1
    n = min(pool, max(32 * cands[0], 64), budget_reads)
    cb = cpu_pass(models, hb, list(range(n)), cands, calibrate, from_raw, repeats=2)
    runs = [dict(threads=th, reads=n, value=round(n / cb["timings"][th]["seconds"], 2), align_s=round(cb["timings"][th]["align"], 2),
                 score_s=round(cb["timings"][th]["score"], 2)) for th in cands]
    th = max(cands, key=lambda c: n / cb["timings"][c]["seconds"])
    t = cb["timings"][th]
    rate = n / t["seconds"]
    out.update(value=round(rate, 2), cores=cores, threads=th, per_core=round(rate / th, 2), runs=runs,
               sample="%d of the same synthetic reads, OpenMP over reads, %d threads (detect %.2fs + align %.2fs + calibrate %.2fs + "
                      "score %.2fs); t1 = %d reads on one thread" % (n, th, t["detect"], t["align"], t["calib"], t["score"], n1))
    cb["threads"] = th
    return out, cb


def compare_with_cpu(batch, cb, idx, from_raw):
    """GPU results of reads `idx` of `batch` against the CPU pass `cb` over the same reads (cpu_pass keeps idx's order): detected
    event counts (from raw), aligned pairs bit for bit, and the LLR of every group the CPU scored."""
    pairs, pair_off, n_pairs = cb["pairs"]
    ok = True
    if from_raw:
        ok &= np.array_equal(batch.d_n_events.cpu().numpy()[idx], np.array(cb["n_events"]))
    bad_reads = 0
    for q, i in enumerate(idx):
        same = np.array_equal(batch.pairs_of(i), pairs[pair_off[q]:pair_off[q] + n_pairs[q]])
        bad_reads += 0 if same else 1
    ok &= bad_reads == 0
    gmap = {}
    if list(idx) == list(range(len(idx))):
        per_read = batch.groups_bulk(len(idx))
    else:
        per_read = (batch.groups_of(i) for i in idx)
    for q, (f, nm, su, sm) in enumerate(per_read):
        for j in range(len(f)):
            gmap[(q, int(f[j]))] = (float(su[j]), float(sm[j]))
    want = cb["scores"]
    d, missing = [], 0
    for j, key in enumerate(cb["first"]):
        g = gmap.get(key)
        if g is None or not np.isfinite(g[0]):
            missing += 1
            continue
        d.append((g[1] - g[0]) - (float(want[2 * j + 1]) - float(want[2 * j])))
    return dict(reads=len(idx), reads_pairs_compared=len(idx), reads_pairs_differ=bad_reads, groups=len(cb["first"]),
                groups_scored_on_gpu=sum(1 for v in gmap.values() if np.isfinite(v[0])), groups_missing_on_gpu=missing,
                pairs_bit_exact=bool(ok), max_abs_dLLR=float(np.max(np.abs(d))) if d else None)


def ragged_parity(models, hb, batch, calibrate, from_raw, n_sample=48):
    """Pairs and LLRs of a sample of the RAGGED batch against the CPU pass (the ragged leg was only checked for
    `reads_aligned_ok`): the longest and the shortest reads of the batch and a spread between them."""
    lens = np.array([len(q) for q in hb["ref_seqs"]])
    order = np.argsort(lens)
    pick = sorted(set(order[-4:].tolist() + order[:4].tolist() + order[np.linspace(0, len(order) - 1, n_sample).astype(int)].tolist()))
    _, _, cores = usable_cores()
    cb = cpu_pass(models, hb, pick, [max(1, min(cores, len(pick)))], calibrate, from_raw, repeats=1)
    out = compare_with_cpu(batch, cb, pick, from_raw)
    out.update(read_len_min=int(lens[pick].min()), read_len_max=int(lens[pick].max()))
    return out


def rank_sample_parity(models, hb, batch, calibrate, from_raw, rank, threads, n_sample=32):
    """N > 1: every rank checks a sample of ITS OWN shard against the CPU pass (a rank that computed garbage must not hide behind
    rank 0's check): n_sample reads spread over the shard, on this rank's share of the host cores."""
    n = len(hb["reads"])
    pick = sorted(set(np.linspace(0, n - 1, min(n, n_sample)).astype(int).tolist()))
    cb = cpu_pass(models, hb, pick, [max(1, threads)], calibrate, from_raw, repeats=1)
    out = compare_with_cpu(batch, cb, pick, from_raw)
    out["rank"] = rank
    return out


# ---- the legs VERDICT r4 item 3 asked to see in the driver-timed line --------------------------------------------------------
def from_raw_leg(ctx, torch, models, hb_raw, tile, steps, warmup, calibrate, cpu_check):
    """The step from RAW SIGNAL (SURVEY 8 f2 in front: int16 ADC counts -> pA -> scrappie event detection -> MoM scalings -> the resident
    step), inputs resident, `steps` timed steps.  Parity: the sites per read of a sample against the reference's WHOLE per-read function
    (SquiggleRead::load_from_raw + calculate_methylation_for_read, compiled in place)."""
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch
    b = CallMethylationBatch(ctx, tile_host_batch(hb_raw, tile), "cuda:%d" % torch.cuda.current_device(), calibrate=calibrate, from_raw=True,
                             jobs_on_device=True, map_stop=False)
    for _ in range(warmup):
        b.step()
    ctx.sync(); torch.cuda.synchronize()
    for w in (0, 1, 2, 4, 5):
        ctx.kernel_time(w, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        b.step()
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fam = {n: round(ctx.kernel_time(w)[0] / max(1, steps), 3) for w, n in ((4, "event_detect"), (5, "mom_scalings"), (0, "event_align"), (2, "glue_and_work_items"), (1, "hmm_score"))}
    nev = b.d_n_events.clamp(min=0).to(torch.int64)
    out = dict(value=round(b.n_reads * steps / dt, 2), unit="reads/s", ms_per_step=round(dt / steps * 1e3, 3), reads_per_step=b.n_reads,
               distinct_reads=hb_raw["n"], mean_events_detected=round(float(nev.sum().item()) / b.n_reads, 1), kernel_ms_per_step=fam,
               reads_aligned_ok=int((b.d_n_pairs > 0).sum().item()), input="int16 ADC counts resident in HBM")
    if cpu_check:
        try:
            from oracle.ref_full import FullRef, have_full
            if have_full():
                n = min(64, hb_raw["n"])
                rds = hb_raw["reads"][:n]
                _, _, cores = usable_cores()
                sites, t_full = FullRef().many_identity(1, [r["seq"] for r in rds], [r["raw"] for r in rds], [r["rc"] for r in rds], max(1, cores))
                n_gpu = [int(np.isfinite(b.groups_of(i)[2]).sum()) for i in range(n)]
                out["check"] = dict(reads=n, what="sites per read against load_from_raw + calculate_methylation_for_read, reference code",
                                    sites_per_read_match_reference=bool(np.array_equal(sites, np.array(n_gpu))), sites=int(np.sum(n_gpu)),
                                    cpu_whole_function_reads_per_s=round(n / t_full, 2))
        except Exception as e:  # noqa: BLE001
            out["check"] = dict(error=repr(e))
    del b
    torch.cuda.empty_cache()
    return out


def _binding_records(models, distinct, read_len):
    from nanopolish_amd import api
    from nanopolish_amd.synth import synth_raw
    recs, contig, pos = [], [], 0
    for r in range(distinct):
        rd = synth_raw(r, models["nucleotide"], L=read_len, k=6, adc=True)
        ref = api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]
        recs.append(dict(seq=rd["seq"], raw=rd["raw"].astype(np.float32), adc=rd["adc"], rc=int(rd["rc"]), pos=pos,
                         cigar=np.array([(len(ref) << 4) | 0], np.uint32), bam_seq=ref))
        contig.append(ref); pos += len(ref)
    return recs, "".join(contig)


def binding_child(bs, distinct, read_len, target_reads):
    """One batch size through the binding in a process of its own (`python bench.py --binding-child ...`): the caller of NpBatchPipeline is a
    nanopolish process, not one that also holds torch's allocator, three other workloads' buffers and their thread pools -- inside bench.py's own
    process the binding's host phases (packing, map building: all 16 CPUs) ran 25-40 % slower than in a clean one (profiles/r06_batch_binding.md)."""
    from oracle.ref_full import bench_batch
    from nanopolish_amd.synth import ADC_OFFSET, ADC_UNIT
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores()[2])))
    recs, contig = _binding_records(load_models(), distinct, read_len)
    nb = max(8, -(-target_reads // bs))
    # (warm-up: every slot of the pipeline used twice and the passes' buffers grown to their steady size)
    sec, sites, bad, hs = bench_batch(recs, contig, bs, nb, warmup=max(7, min(192, 98304 // bs)), pipelined=True, adc=(float(ADC_OFFSET), float(ADC_UNIT)), contexts=0, consumer=1)
    return dict(value=round(bs * nb / sec, 1), unit="reads/s", records_per_batch=bs, batches=nb, ms_per_batch=round(sec / nb * 1e3, 2),
                records_not_ok=bad, sites_written=sites, host_ms_per_batch={k: round(v / nb * 1e3, 2) for k, v in hs.items()})


def binding_legs(models, sizes=(512, 8192), distinct=512, read_len=5450, target_reads=262144, cpu_check=True):
    """Reads/s THROUGH the reference-side batched binding (nanopolish_amd/csrc/np_batch_dropin.cpp: NpBatchPipeline linked into the
    reference's read-level build in place of call-methylation's per-record loop, src/common/nanopolish_bam_processor.cpp:90-119,
    INTEGRATION.md section 2): BAM records + int16 raw signal in HOST memory in, the reference's ScoredSite maps out, host wall clock
    around the whole loop, at BamProcessor's default batch size (512 records) and at 8 192.  The harness that plays the caller
    (oracle/ref_full_harness.cpp) and the reference objects live under oracle/_ref; what is timed is the product's binding + library.
    Every size runs in a process of its own (binding_child).
    Parity: the sites written must equal, in number, what the reference's whole per-read function finds for the same records."""
    from oracle.ref_full import have_batch
    if not have_batch():
        return dict(error="oracle/_ref/libnp_ref_full_batch.so did not travel with the repository")
    want_sites = None
    if cpu_check:
        try:
            from oracle.ref_full import FullRef, have_full
            if have_full():
                recs, _ = _binding_records(models, distinct, read_len)
                sites, _ = FullRef().many_identity(1, [r["seq"] for r in recs], [r["raw"] for r in recs], [r["rc"] for r in recs], max(1, usable_cores()[2]))
                want_sites = int(np.sum(sites))
        except Exception:  # noqa: BLE001
            want_sites = None
    out = dict(distinct_reads=distinct, read_len=read_len, what="NpBatchPipeline (default construction: two contexts on the device), int16 samples from host memory, "
                                                              "ScoredSite maps handed over and recycled; host wall clock; each batch size in a process of its own")
    for bs in sizes:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--binding-child", "%d,%d,%d,%d" % (bs, distinct, read_len, target_reads)],
                           capture_output=True, text=True, timeout=900)
        ls = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not ls:
            out["records_%d" % bs] = dict(error="binding child failed (rc %d): %s" % (r.returncode, r.stderr[-400:]))
            continue
        d = json.loads(ls[-1])
        if want_sites is not None and bs % distinct == 0:
            d["sites_match_reference"] = bool(d["sites_written"] == want_sites * (bs // distinct) * d["batches"])
        out["records_%d" % bs] = d
    return out


# ---- N > 1 without a launcher ---------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Start the n ranks of `bench.py --gpus n` ourselves (the driver's N = 1 command shape with --gpus > 1): one process per
    GPU, rendezvous on 127.0.0.1.  Rank 0's JSON line is this process's output."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def timed_steps(batch_step, barrier, steps, warmup, before_stop=None):
    for _ in range(warmup):
        batch_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch_step()
    if before_stop:
        before_stop()
    barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="call-methylation", choices=["call-methylation", "eventalign", "variants", "cpu-t1"])
    ap.add_argument("--pool", type=int, default=-1,
                    help="distinct synthetic reads per rank (-1: 20 000 at --gpus 1 -- BASELINE.json configs[1], 100 000 reads per step -- and "
                         "50 000 at --gpus N > 1 -- configs[4], 250 000 reads per rank per step, 2 M over 8 GPUs)")
    ap.add_argument("--tile", type=int, default=5, help="independent HBM copies of the pool per batch (pool x tile reads per rank and step)")
    ap.add_argument("--read-len", type=int, default=5450, help="bases per read (5450 -> ~8k events)")
    ap.add_argument("--calibrate", type=int, default=1,
                    help="1: recalibrate each read on the device between the two kernels, as load_from_raw does (SURVEY 8 f1); "
                         "0: score with the scalings the synthetic reads were made with")
    ap.add_argument("--from-raw", type=int, default=0,
                    help="1: a step starts from raw current samples (scrappie event detection + MoM scalings on the device, "
                         "SURVEY 8 f2); 0: from pre-detected events")
    ap.add_argument("--jobs-on-device", type=int, default=1, help="1: work items are generated on the device inside the step (SURVEY 8 f3)")
    ap.add_argument("--streamed", type=int, default=-1,
                    help="1: also time the host-fed (pinned, double-buffered) variant (-1: on at --gpus 1; off at N > 1, where every rank "
                         "would pin 12 GB of host memory for its 250 000 reads)")
    ap.add_argument("--ragged", type=int, default=1, help="1: also time a batch with log-normal read lengths of the same mean")
    ap.add_argument("--ragged-pool", type=int, default=-1, help="distinct reads of the ragged batch (-1: pool / 2, tile x 2)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="cap on the reads of the CPU baseline (-1: 32 per thread, 0: skip)")
    ap.add_argument("--workers", type=int, default=-1, help="host-preparation worker processes (-1: min(32, cores))")
    ap.add_argument("--genome", type=int, default=-1,
                    help="1: the reads have a place on the seeded 5 Mb genome (BAM-style records; work items by CIGAR on the device) and the per-site "
                         "table is keyed by genome position -- what the ranks of an N > 1 run all-reduce; 0: reads identity-aligned to themselves "
                         "(configs[1]); -1: 1 at --gpus N > 1, 0 at --gpus 1")
    ap.add_argument("--site-rows", choices=("site", "base"), default="site",
                    help="genome mode: rows of the all-reduced table -- one per motif site of the genome (np_genome_site_index_dev's ordinals; 24 bytes x "
                         "sites) or one per base (24 bytes x bases)")
    ap.add_argument("--parity-reads", type=int, default=12, help="genome mode: reads of every rank's shard checked against the oracle on the same BAM record")
    ap.add_argument("--binding-child", default="", help=argparse.SUPPRESS)       # "bs,distinct,read_len,target_reads": binding_child in this process
    ap.add_argument("--legs", type=int, default=1,
                    help="1: rank 0 of a one-GPU run also measures BASELINE.json configs[2] (eventalign, 50 000 reads per step) and configs[3] "
                         "(variants, 10 kb x 2 000 reads) and folds them into the line as value_eventalign / value_variants, each with its "
                         "parity fields and its own roofline")
    args, extra = ap.parse_known_args()

    if args.binding_child:
        import torch  # noqa: F401  (one HIP runtime per process, loaded before the binding's library)
        print(json.dumps(binding_child(*[int(x) for x in args.binding_child.split(",")])), flush=True)
        return
    if args.workload in ("eventalign", "variants"):
        # BASELINE.json configs[2] / configs[3]: their own tools (one JSON line each), same --steps / --warmup
        import runpy
        tool = os.path.join(ROOT, "tests", "bench_%s.py" % args.workload)
        sys.argv = [tool, "--steps", str(args.steps), "--warmup", str(args.warmup)] + extra
        runpy.run_path(tool, run_name="__main__")
        return
    if extra:
        ap.error("unknown arguments: %s" % " ".join(extra))
    if args.pool < 0:
        args.pool = 20000 if args.gpus == 1 else 50000
    if args.streamed < 0:
        args.streamed = 1 if args.gpus == 1 else 0
    if args.genome < 0:
        args.genome = 1 if args.gpus > 1 else 0
    if args.genome:
        if args.from_raw:
            ap.error("--genome 1 starts from events (configs[4] as configs[1]: pre-detected events); the from-raw chain on records is the eventalign leg")
        args.streamed = 0; args.ragged = 0; args.legs = 0           # (the host-fed / ragged variants and the other workloads are legs of the identity line)
    models = load_models()
    if args.workload == "cpu-t1":
        # BASELINE.json configs[0]: the CPU plumbing line (-t 1), the reference's own code on one host thread, no GPU
        n = min(args.pool, 1000)
        n = args.cpu_sample if args.cpu_sample > 0 else n
        hb = prep_host_batch(models, 0, n, args.read_len, False, 1)
        cb = cpu_pass(models, hb, list(range(n)), [1], bool(args.calibrate), False, repeats=1)
        t = cb["timings"][1]
        print(json.dumps(dict(metric="call-methylation reads/sec", value=round(cb["n"] / t["seconds"], 2), unit="reads/s", n_gpus=0,
                              higher_is_better=True, dtype="f32", data="synthetic",
                              config=dict(workload="call-methylation hot path on ONE host thread (-t 1), %d synthetic R9.4 reads "
                                                   "(BASELINE.json configs[0] shape; the bundled E. coli subset is not in this image)" % cb["n"],
                                          kind=cb["kind"], align_s=round(t["align"], 2), score_s=round(t["score"], 2),
                                          calibrate_s=round(t["calib"], 3)))), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    from nanopolish_amd.hostinfo import usable_cores
    cores = usable_cores()[2]                  # affinity mask capped by the cgroup CPU quota (16 on the pool's boxes, whatever nproc says)
    workers = args.workers if args.workers > 0 else max(1, min(32, (2 * cores) // max(1, world)))

    # ---- host preparation (before the HIP runtime is initialised: the pool forks) ----
    from nanopolish_amd.shard import shard_read_ids
    t_prep = time.perf_counter()
    lo, hi = shard_read_ids(world * args.pool, rank, world)          # reads shard by contiguous id range
    hb = prep_record_batch(models, lo, hi, args.read_len, workers) if args.genome else prep_host_batch(models, lo, hi, args.read_len, bool(args.from_raw), workers)
    hb_rag = None
    if args.ragged:
        rp = args.ragged_pool if args.ragged_pool > 0 else max(1, args.pool // 2)
        rt = max(1, (args.pool * args.tile) // rp)
        rlo, rhi = shard_read_ids(world * rp, rank, world)
        ids = np.arange(rlo, rhi) + (1 << 24)                          # its own id range
        hb_rag = prep_host_batch(models, int(ids[0]), int(ids[-1]) + 1, ragged_lengths(ids, args.read_len), bool(args.from_raw), workers)
    hb_raw = None
    if world == 1 and args.legs and not args.from_raw and not args.genome:
        # the from-raw leg's pool (int16 traces): 4 000 distinct reads, 25 copies in HBM = the same 100 000 reads per step
        hb_raw = prep_host_batch(models, (1 << 25), (1 << 25) + min(4000, args.pool), args.read_len, True, workers)
    t_prep = time.perf_counter() - t_prep

    import torch
    import torch.distributed as dist
    # one process per GPU.  NP_BENCH_BACKEND=gloo + fewer devices than ranks is a single-GPU rehearsal of the N>1 code path
    backend = os.environ.get("NP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    if ndev < world and backend == "nccl":
        raise SystemExit("--gpus %d but only %d device(s) visible (NP_BENCH_BACKEND=gloo rehearses the N>1 path on fewer)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    class stdout_to_stderr:
        """gloo announces every new process group on STDOUT ("[Gloo] Rank 0 is connected to ..."); the line rank 0 prints must stay the
        only thing there: file descriptor 1 points at stderr while a group is being created"""
        def __enter__(self):
            sys.stdout.flush(); self.saved = os.dup(1); os.dup2(2, 1)
        def __exit__(self, *a):
            sys.stdout.flush(); os.dup2(self.saved, 1); os.close(self.saved)

    if world > 1:
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
                dist.barrier()                     # (gloo connects lazily: make it talk now)

    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import tile_host_batch, CallMethylationBatch, StreamedFeed
    from nanopolish_amd.sites import site_table_dev

    ctx = Context(local)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    dev = "cuda:%d" % local

    def barrier(stream=None):
        ctx.sync(stream); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctx.sync(stream); torch.cuda.synchronize()

    def make_batch(h, tile):
        # (map_stop=False: call-methylation never reads base_to_event_map[].stop -- get_closest_event_to and the recalibration walk .start,
        #  squiggle_read.cpp:161-186,339-389; the eventalign leg, whose binding hands the map back to the reference, builds both)
        return CallMethylationBatch(ctx, tile_host_batch(h, tile), dev, calibrate=bool(args.calibrate), from_raw=bool(args.from_raw),
                                    jobs_on_device=bool(args.jobs_on_device), map_stop=False)

    def all_reduce(t, op):
        """all-reduce of a device tensor; the gloo rehearsal backend reduces a host copy"""
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(t, op=op)
            else:
                h = t.cpu(); dist.all_reduce(h, op=op); t.copy_(h)
        return t

    def device_table(b):
        """per-site table of this rank's reads (the all-reduce payload) from the device-resident scores and group metadata
        (np_site_table_dev); skipped groups and unused group slots carry NaN scores"""
        if args.genome:
            # keyed (contig, start, end) on the resident genome: int32 [rows, 6], one row per motif site of the genome (or per base:
            # --site-rows) -- np_site_table_genome_indexed_dev / np_site_table_genome_dev; groups cut on both sides are counted beside it
            t, ovf = b.genome_site_table(per_site=args.site_rows == "site")
            ctx.sync()
            b.site_overflow = ovf
            return t
        t = site_table_dev(ctx, torch, b.d_scores, b.d_first, b.d_n_motif, b.max_len)     # on the library's stream, after the steps
        ctx.sync()
        return t

    # ---------------- resident (headline) ----------------
    batch = make_batch(hb, args.tile)
    batch.max_len = int(max(len(q) for q in hb["ref_seqs"]))
    if not batch.jobs_on_device:
        raise SystemExit("--jobs-on-device 0 is no longer a bench configuration (work items are part of the step)")
    n_reads = batch.n_reads
    for _ in range(args.warmup):
        batch.step()
    if world > 1:
        ctx.sync()
        all_reduce(device_table(batch), dist.ReduceOp.SUM)          # warm the collective path (RCCL communicator set-up is not a step)
    barrier()
    for w in range(9):
        ctx.kernel_time(w, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.step()
    ctx.sync()
    t_steps = time.perf_counter() - t0          # this rank's K steps alone (N > 1: what the per-rank diagnostics report)
    table = None
    t_reduce = 0.0
    if world > 1:
        t1 = time.perf_counter()
        table = device_table(batch)
        all_reduce(table, dist.ReduceOp.SUM)        # RCCL all-reduce(sum): the job's only collective (final site-level reduction)
        torch.cuda.synchronize()
        t_reduce = time.perf_counter() - t1         # table kernel + all-reduce, incl. waiting for the slowest rank
    barrier()
    dt = time.perf_counter() - t0
    dt_rank = dt

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        return float(all_reduce(t, dist.ReduceOp.MAX).item())

    dt = max_over_ranks(dt)
    if table is None:
        t1 = time.perf_counter()
        table = device_table(batch)                 # one rank: the same table, outside the timed region
        torch.cuda.synchronize()
        t_reduce = time.perf_counter() - t1         # (reported per rank as table_and_allreduce_ms: the table kernel alone here)
    batch_overflow = getattr(batch, "site_overflow", None)
    ovf_total = int(all_reduce(batch_overflow.clone(), dist.ReduceOp.SUM).item()) if batch_overflow is not None else None      # (every rank takes part)
    k_ms = {}
    for name, w in (("event_align", 0), ("hmm_score", 1), ("glue_and_work_items", 2), ("event_detect", 4), ("mom_scalings", 5)):
        k_ms[name] = ctx.kernel_time(w)

    # ---------------- streamed: the same batch, host-fed ----------------
    streamed = None
    if args.streamed:
        feed = StreamedFeed(batch)
        for k in range(args.warmup + 1):
            feed.submit()
        feed.drain(); barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            feed.submit()
        feed.drain(); barrier()
        dts_rank = time.perf_counter() - t0
        dts = max_over_ranks(dts_rank)
        steady = feed.steady_ms_per_step(args.steps)
        same = feed.check_against(batch) if rank == 0 else True
        streamed = dict(value=round(world * n_reads * args.steps / dts, 2), value_rank=round(n_reads * args.steps / dts_rank, 2), ms_per_step=round(dts / args.steps * 1e3, 3),
                        h2d_bytes_per_step=feed.h2d_bytes, d2h_bytes_per_step=feed.d2h_bytes,
                        pcie_GBps=round((feed.h2d_bytes + feed.d2h_bytes) * args.steps / dts / 1e9, 2),
                        steady_state_ms_per_step=round(steady, 3) if steady else None,
                        note="value_streamed times K steps from a cold pipeline: the first step's upload is not overlapped; "
                             "steady_state_ms_per_step is the interval between step completions after it",
                        results_equal_resident=bool(same), host_enqueue_ms_last_steps=feed.host_ms[-args.steps:])
        feed.close()
        batch.stream = None
        del feed                       # (it holds the second buffer set and clones of the outputs)

    # results of the resident batch on the host (rank 0), before it is released
    res = None
    if rank == 0:
        scores = batch.scores()
        llr = scores[1::2].astype(np.float64) - scores[0::2]
        n_groups = int(np.isfinite(llr).sum())
        n_ok = int((batch.d_n_pairs > 0).sum().item())
        res = dict(n_groups=n_groups, n_ok=n_ok)
        if args.from_raw:
            # the event counts only exist after the detector has run: algorithmic bytes from the detected counts
            nev = batch.d_n_events.clamp(min=0).to(torch.int64).cpu().numpy()
            nk = (batch.hb["rank_off"][1:] - batch.hb["rank_off"][:-1])
            bands = nev + nk + 2
            batch.algo_bytes_align = int((4 * nev + 2 * nk + 100 * bands + 8 * nev).sum())
            batch.band_cells = int((100 * bands).sum()); batch.total_events = int(nev.sum())
        res.update(algo=batch.algo_bytes_align, band_cells=batch.band_cells, total_events=batch.total_events)

    # ---------------- parity + CPU baseline (outside every timed region) ----------------
    # N = 1: rank 0 times the CPU pass and compares its reads with the GPU's.  N > 1: EVERY rank first checks a
    # 32-read sample of its own shard against the CPU pass on its share of the host cores; the mismatch counts are summed over the
    # ranks and each rank's verdict lands in per_rank.  Then rank 0 runs the same cpu_baseline + pairs / LLR check as at N = 1 while
    # the other ranks wait on a SOCKET barrier (a gloo group: an RCCL barrier would keep N - 1 host threads spinning on the very
    # cores the baseline is timed on).
    cpu = None
    max_dllr = None
    rank_check = None
    host_group = None
    if world > 1:
        with stdout_to_stderr():
            host_group = dist.new_group(backend="gloo") if backend == "nccl" else dist.group.WORLD
            dist.barrier(group=host_group)

    def host_barrier():
        if world > 1:
            dist.barrier(group=host_group)

    if args.genome and args.cpu_sample != 0:
        rank_check = record_sample_parity(models, hb, batch, rank, n_sample=args.parity_reads)          # every rank, its own shard, the oracle on the same records
        host_barrier()
    elif world > 1 and args.cpu_sample != 0:
        rank_check = rank_sample_parity(models, hb, batch, bool(args.calibrate), bool(args.from_raw), rank, max(1, cores // world))
        host_barrier()
    if rank == 0 and args.cpu_sample != 0 and not args.genome:
        budget = args.cpu_sample if args.cpu_sample > 0 else 1 << 30
        cpu, cb = cpu_baseline(models, hb, bool(args.calibrate), bool(args.from_raw), budget)
        # parity of the GPU results with the CPU pass on that sample: pairs bit-exact, LLR within 1e-4
        cpu["check"] = compare_with_cpu(batch, cb, list(range(cb["n"])), bool(args.from_raw))
        max_dllr = cpu["check"]["max_abs_dLLR"]
        if args.from_raw and args.calibrate:
            # the same sample through the reference's WHOLE per-read function: SquiggleRead(sequence, Fast5Data) -> load_from_raw
            # -> calculate_methylation_for_read, compiled in place (oracle/_ref/libnp_ref_full.so), OpenMP over reads
            try:
                from oracle.ref_full import FullRef, have_full
                if have_full():
                    rds = hb["reads"][:cb["n"]]
                    sites, t_full = FullRef().many_identity(1, [r["seq"] for r in rds], [r["raw"] for r in rds], [r["rc"] for r in rds], cb["threads"])
                    n_gpu = [int(np.isfinite(batch.groups_of(i)[2]).sum()) for i in range(cb["n"])]
                    cpu["whole_function"] = dict(value=round(cb["n"] / t_full, 2), unit="reads/s",
                                                 what="load_from_raw + calculate_methylation_for_read per read, reference code",
                                                 sites_per_read_match_gpu=bool(np.array_equal(sites, np.array(n_gpu))))
            except Exception as e:  # noqa: BLE001
                cpu["whole_function"] = dict(error=repr(e))
    host_barrier()

    # ---------------- ragged read lengths (resident) ----------------
    ragged = None
    mean_events = batch.total_events / n_reads
    del batch
    torch.cuda.empty_cache()
    if hb_rag is not None:
        rt = max(1, (args.pool * args.tile) // hb_rag["n"])
        rb = make_batch(hb_rag, rt)
        ctx.kernel_time(0, reset=True)
        dtr = max_over_ranks(timed_steps(rb.step, barrier, args.steps, args.warmup, ctx.sync))
        a_ms, a_n = ctx.kernel_time(0)
        lens = np.array([len(q) for q in hb_rag["ref_seqs"]])
        nev_r = rb.total_events if not args.from_raw else int(rb.d_n_events.clamp(min=0).sum().item())
        rag_check = None
        if rank == 0 and args.cpu_sample != 0:
            rag_check = ragged_parity(models, hb_rag, rb, bool(args.calibrate), bool(args.from_raw))
        host_barrier()
        ragged = dict(value=round(world * rb.n_reads * args.steps / dtr, 2), ms_per_step=round(dtr / args.steps * 1e3, 3), check=rag_check,
                      reads_per_step_per_gpu=rb.n_reads, distinct_reads_per_gpu=hb_rag["n"],
                      read_len=dict(mean=round(float(lens.mean()), 1), p50=int(np.median(lens)), min=int(lens.min()), max=int(lens.max())),
                      mean_events=round(nev_r / rb.n_reads, 1),
                      events_per_s=round(world * nev_r * args.steps / dtr, 1),
                      event_align_ms_per_step=round(a_ms / a_n, 3) if a_n else None,
                      reads_aligned_ok=int((rb.d_n_pairs > 0).sum().item()))
        del rb
        torch.cuda.empty_cache()

    # per-rank diagnostics: with N > 1 a scaling efficiency below 0.9 must be attributable -- this rank's own
    # K steps, its host-fed rate, its host preparation time and what the final table + all-reduce cost it
    mine = [float(rank), n_reads * args.steps / t_steps, (streamed or {}).get("value_rank", 0.0), t_prep, t_reduce * 1e3, dt_rank * 1e3,
            k_ms["event_align"][0] / max(args.steps, 1), k_ms["hmm_score"][0] / max(args.steps, 1)]
    rc_ = rank_check or {}
    dl = rc_.get("max_abs_dLLR")
    mine += [float(rc_.get("reads", 0)), float(rc_.get("reads_pairs_differ", 0)), float(rc_.get("groups", 0)),
             float(rc_.get("groups_missing_on_gpu", 0)), float(dl) if dl is not None else -1.0]
    gathered = [mine]
    if world > 1:
        t = torch.tensor(mine, dtype=torch.float64, device="cuda")
        lst = [torch.zeros_like(t) for _ in range(world)]
        if backend == "nccl":
            dist.all_gather(lst, t)
        else:
            hl = [x.cpu() for x in lst]; dist.all_gather(hl, t.cpu()); lst = hl
        gathered = [x.cpu().tolist() for x in lst]
    per_rank = [dict(rank=int(g[0]), value=round(g[1], 1), value_streamed=round(g[2], 1) if g[2] else None, host_prep_s=round(g[3], 1),
                     table_and_allreduce_ms=round(g[4], 3), timed_region_ms=round(g[5], 3), event_align_ms_per_step=round(g[6], 3),
                     hmm_score_ms_per_step=round(g[7], 3),
                     check=dict(reads=int(g[8]), reads_pairs_differ=int(g[9]), groups=int(g[10]), groups_missing_on_gpu=int(g[11]),
                                max_abs_dLLR=(g[12] if g[12] >= 0 else None)) if g[8] else None) for g in gathered]
    shard_check = None
    if (world > 1 or args.genome) and any(pr["check"] for pr in per_rank):
        cks = [pr["check"] for pr in per_rank if pr["check"]]
        dls = [c["max_abs_dLLR"] for c in cks if c["max_abs_dLLR"] is not None]
        shard_check = dict(ranks_checked=len(cks), reads=sum(c["reads"] for c in cks), reads_pairs_differ=sum(c["reads_pairs_differ"] for c in cks),
                           groups=sum(c["groups"] for c in cks), groups_missing_on_gpu=sum(c["groups_missing_on_gpu"] for c in cks),
                           max_abs_dLLR=max(dls) if dls else None,
                           what=("every rank: a sample of its own shard (--parity-reads) through the oracle's restatement of the reference's per-read pass on the same BAM "
                                 "record (reads_pairs_differ: reads whose set of scored genome sites differs), LLRs compared" if args.genome else
                                 "every rank: 32 reads of its own shard, GPU pairs and LLRs against the CPU pass"))
        if args.genome and max_dllr is None:
            max_dllr = shard_check["max_abs_dLLR"]

    # ---------------- BASELINE.json configs[2] and configs[3], folded into the line (one GPU, rank 0) ----------------
    legs = None
    if rank == 0 and world == 1 and args.legs:
        legs = {}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        if hb_raw is not None:
            try:
                fr = from_raw_leg(ctx, torch, models, hb_raw, max(1, n_reads // hb_raw["n"]), args.steps, args.warmup, bool(args.calibrate), args.cpu_sample != 0)
                legs["value_from_raw"] = fr["value"]; legs["from_raw"] = fr
            except Exception as e:  # noqa: BLE001
                legs["from_raw"] = dict(error=repr(e))
            hb_raw = None
            torch.cuda.empty_cache()
        try:
            import bench_eventalign
            ea = bench_eventalign.run(steps=args.steps, warmup=args.warmup, cpu_sample=0 if args.cpu_sample == 0 else -1, ctx=ctx)
            legs["value_eventalign"] = ea["value"]; legs["eventalign"] = ea
        except Exception as e:  # noqa: BLE001
            legs["eventalign"] = dict(error=repr(e))
        torch.cuda.empty_cache()
        try:
            import bench_variants
            va = bench_variants.run(steps=args.steps, warmup=args.warmup, cpu_sample=0 if args.cpu_sample == 0 else 400000)
            legs["value_variants"] = va["value"]; legs["variants"] = va
        except Exception as e:  # noqa: BLE001
            legs["variants"] = dict(error=repr(e))
        torch.cuda.empty_cache()
        try:
            bl = binding_legs(models, cpu_check=args.cpu_sample != 0)
            legs["binding"] = bl
            for bs in (512, 8192):
                if "value" in (bl.get("records_%d" % bs) or {}):
                    legs["value_binding_%d" % bs] = bl["records_%d" % bs]["value"]
        except Exception as e:  # noqa: BLE001
            legs["binding"] = dict(error=repr(e))

    if rank == 0:
        # dominant kernel + HBM roofline (algorithmic bytes)
        dom = max(k_ms, key=lambda k: k_ms[k][0])
        a_ms, a_n = k_ms["event_align"]
        a_avg_s = a_ms / max(a_n, 1) * 1e-3
        algo = res["algo"]
        achieved = algo / a_avg_s / 1e9 if a_avg_s > 0 else 0.0
        # HBM traffic of THIS run's launch: bytes per BAND from the counter passes over the shipped kernel (profiles/r04_pmc.json:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE at a launch size the passes finish at, profiles/collect_r04_pmc.sh) x the bands this
        # run's reads have -- a from-raw run (more events per read) reports its own figure.  `issue`: vector instructions per band from
        # the same passes, priced at the calibrated issue cycles per class (profiles/r04_valu_calibration.json), over this run's
        # SIMD-cycles per band: floor = every instruction at the fastest class's cost, priced = by the class mix of the band loop.
        n_bands = res["band_cells"] // 100
        cyc_per_band = a_avg_s * CLOCK_HZ * N_SIMD / max(n_bands, 1)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_lookup
        traffic = pmc_lookup.traffic("event_align", "band", n_bands)
        issue = pmc_lookup.issue("event_align", "band", cyc_per_band) or dict(kind="unavailable", simd_cycles_per_band=round(cyc_per_band, 1))
        # the kernel's own `limiter` says vector issue, so THAT is the line's roofline (bound = "valu_issue": the counters'
        # vector instructions per band x the guide's 2 cycles per wave64 instruction over the SIMD-cycles this launch spent per band); the HBM
        # roofline on the algorithmic bytes stays beside it as roofline.hbm, with the counter traffic
        hbm = dict(bound="hbm", achieved=round(achieved, 2), peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 5), traffic=traffic,
                   algo_bytes_per_launch=algo)
        ri = pmc_lookup.roofline_issue("event_align", "band", cyc_per_band, "np_event_align_kernel")
        roof = dict(ri) if ri else dict(hbm)
        roof.update(kernel="np_event_align_kernel", traffic=traffic, hbm=hbm,
                    algo_bytes_per_launch=algo, avg_launch_ms=round(a_ms / max(a_n, 1), 3),
                    band_cells_per_s=round(res["band_cells"] / a_avg_s / 1e9, 3) if a_avg_s > 0 else 0.0,
                    limiter="vector-instruction issue (one wave per read, ~13.5k dependent band steps; HBM traffic ~0.9 x the algorithmic bytes): see issue",
                    issue=issue, dominant_kernel_by_time=dom,
                    kernel_ms_per_step={k: round(v[0] / max(args.steps, 1), 3) for k, v in k_ms.items()})

        value = world * n_reads * args.steps / dt
        # kernel B of the same step: HBM roofline on the algorithmic bytes (4 e + 2 n + 12 n + 4 per call, from the counter
        # passes' calls) and the vector-issue roofline it actually sits under
        h_ms, h_n = k_ms["hmm_score"]
        calls = 2 * res["n_groups"]
        roof_b = None
        if h_n and calls:
            h_s = h_ms / h_n * 1e-3
            fb = pmc_lookup.family("hmm_forward")
            ab = fb.get("algo_bytes_per_call")
            roof_b = dict(kernel="np_hmm_forward_kernel", avg_launch_family_ms=round(h_ms / h_n, 3), calls_per_launch=calls,
                          hbm=dict(bound="hbm", achieved=round(ab * calls / h_s / 1e9, 2), peak=8000.0, unit="GB/s", frac=round(ab * calls / h_s / 1e9 / 8000.0, 5),
                                   algo_bytes_per_call=ab, traffic=pmc_lookup.traffic("hmm_forward", "call", calls)) if ab else None,
                          roofline_issue=pmc_lookup.roofline_issue("hmm_forward", "call", h_s * CLOCK_HZ * N_SIMD / calls, "np_hmm_forward_kernel"))
        # the configuration this line is quoted on (BASELINE.json): one GPU = configs[1] (100 000 reads per step); N GPUs = configs[4]
        # (250 000 reads per rank and step: "2M synthetic R9.4 reads sharded across 8 x MI355X", one all-reduce of the per-site table)
        if args.genome:
            workload_name = ("call-methylation, %s synthetic R9.4 reads placed on a seeded %.0f Mb genome (BAM-style records, both strands, indels / clips; "
                             "%d per rank and step) sharded across %dxMI355X, one RCCL all-reduce of the per-site table keyed by genome position "
                             "(BASELINE.json configs[4]%s)" % ("2M" if world * n_reads == 2000000 else "%dk" % (world * n_reads // 1000), GENOME_LEN / 1e6, n_reads, world,
                                                               "" if world == 8 and n_reads == 250000 else ": its shape on %d GPU(s), %d reads per rank" % (world, n_reads)))
        elif world == 1 and n_reads == 100000:
            workload_name = "call-methylation, 100k synthetic R9.4 reads (~8k events each), r9.4_450bps CpG model (BASELINE.json configs[1])"
        elif world > 1 and n_reads == 250000:
            workload_name = ("call-methylation, %s synthetic R9.4 reads sharded across %dxMI355X (250 000 per rank and step), RCCL reduction of the "
                             "per-site table (BASELINE.json configs[4]%s)" % ("2M" if world == 8 else "%dk" % (world * 250), world,
                                                                              "" if world == 8 else ": its per-GPU shape on %d GPUs" % world))
        else:
            workload_name = ("call-methylation, %d synthetic R9.4 reads per rank and step on %d GPU(s), r9.4_450bps CpG model (a non-default --pool / --tile: "
                             "neither BASELINE.json configs[1] nor configs[4])" % (n_reads, world))
        out = dict(metric="call-methylation reads/sec", value=round(value, 2), unit="reads/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=workload_name,
                               reads_per_step_per_gpu=n_reads, reads_total=world * n_reads, distinct_reads_per_gpu=args.pool, tile=args.tile,
                               read_len=args.read_len, mean_events=round(mean_events, 1), jobs_on_device=bool(args.jobs_on_device),
                               groups_per_step_per_gpu=res["n_groups"], reads_aligned_ok=res["n_ok"],
                               calibrate_on_device=bool(args.calibrate), from_raw_signal=bool(args.from_raw),
                               genome_records=bool(args.genome),
                               map_stop=False,   # base_to_event_map[].stop is not built in this step: call-methylation never reads it (the CPU baseline does build it)
                               parallelism="reads sharded over %d GPU(s), 1 process/GPU" % world),
                   cpg_site_groups_per_s=round(world * res["n_groups"] * args.steps / dt, 1),
                   value_streamed=streamed["value"] if streamed else None, streamed=streamed,
                   value_ragged=ragged["value"] if ragged else None, ragged=ragged,
                   max_abs_dLLR_vs_cpu=max_dllr, roofline=roof, roofline_hmm_forward=roof_b, cpu_baseline=cpu, host_prep_s=round(t_prep, 1))
        out["per_rank"] = per_rank
        if shard_check:
            out["shard_check"] = shard_check
        if legs:
            out.update(legs)
        if table is not None and args.genome:
            # keys (contig, start, end): columns 0-2 by start (the group ends where its genome cluster ends), 3-5 by end (a read that stops inside a
            # cluster).  The genome's own cluster count beside it: every cluster some read spans with its flanks shows up as a key.
            from nanopolish_amd.sites import motif_sites
            hit = np.flatnonzero(motif_sites(_GENOME["g"][1].encode(), [0, GENOME_LEN]))
            n_clusters = int(1 + (np.diff(hit) > 10).sum()) if len(hit) else 0
            out["site_table"] = dict(keyed_by="(contig, start, end) on the genome", sites=int((table[:, 0] > 0).sum().item() + (table[:, 3] > 0).sum().item()),
                                     sites_keyed_by_end=int((table[:, 3] > 0).sum().item()), genome_cpg_sites=int(len(hit)), genome_cpg_groups=n_clusters,
                                     num_reads=int(table[:, 0].sum().item() + table[:, 3].sum().item()),
                                     called_sites=int(table[:, 1].sum().item() + table[:, 4].sum().item()),
                                     called_sites_methylated=int(table[:, 2].sum().item() + table[:, 5].sum().item()),
                                     groups_cut_on_both_sides=ovf_total, table_bytes_per_rank=int(table.numel() * 4),
                                     rows="one per motif site of the genome" if args.site_rows == "site" else "one per base", table_rows=int(table.shape[0]),
                                     max_reads_on_one_site=int(table[:, 0].max().item()))
        elif table is not None:
            out["site_table"] = dict(sites=int((table[:, 0] > 0).sum().item()), num_reads=int(table[:, 0].sum().item()),
                                     called_sites=int(table[:, 1].sum().item()), called_sites_methylated=int(table[:, 2].sum().item()))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
