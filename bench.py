#!/usr/bin/env python3
"""bench.py -- call-methylation reads/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic R9.4 reads already resident in HBM:
[scrappie event detection -> MoM scalings ->] adaptive_banded_simple_event_align -> event map / recalibration / window
bounds -> 2 x profile_hmm_score per CpG group
(workload = BASELINE.json configs[1]: ~8k-event reads, r9.4_450bps CpG model).  Reads shard across ranks
with no data-path collective (weak scaling); the only exchange is one all-reduce of the per-site table
at the end of the timed region (N > 1).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_models():
    z = np.load(os.path.join(ROOT, "tests", "golden", "models_r9.4_450bps.npz"))
    return {a: dict(k=6, level_mean=z[a + "_level_mean"], level_stdv=z[a + "_level_stdv"],
                    level_log_stdv=z[a + "_level_log_stdv"]) for a in ("nucleotide", "cpg")}


def cpu_baseline(models, hb, n_sample, threads, calibrate, from_raw=False):
    """CPU baseline on this box's host cores over a bounded sample of the same reads: align + 2 x score per group,
    OpenMP over reads like src/common/nanopolish_bam_processor.cpp:99.  Uses the reference's own code when
    oracle/_ref/libnp_ref.so travelled with the repo (kind="reference"), else the oracle port (kind="port").
    Also returns the results, which double as the parity check of the GPU numbers."""
    from oracle import Oracle, RefOracle, have_ref
    from oracle.workloads import methylation_jobs, K
    orc = Oracle()
    ref = RefOracle() if have_ref() else None
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    n = min(n_sample, len(hb["reads"]))
    rds = hb["reads"][:n]
    t_detect = 0.0
    mom = hb["mom"][:n].copy()
    if from_raw:
        # event detection with the reference's own scrappie objects (kind="reference") or the port, then MoM on the host
        wo = hb["raw_off"][:n + 1]
        t_detect = 1e30
        for _ in range(2):
            evm, evo, evn = (ref or orc).detect_events_many(hb["raw"][:wo[-1]], wo, threads)
            t_detect = min(t_detect, (ref or orc).last_call_s)
        rds = [dict(r, events=evm[evo[i]:evo[i] + evn[i]].copy()) for i, r in enumerate(rds)]
        for i, r in enumerate(rds):
            mom[i] = orc.estimate_scalings_mom(mn, r["ranks"], r["events"])
        eo = np.zeros(n + 1, np.int64); eo[1:] = np.cumsum(evn)
        ev = np.concatenate([r["events"] for r in rds]).astype(np.float32)
        ro = hb["rank_off"][:n + 1]; rk = hb["ranks"][:ro[-1]].astype(np.uint32)
    else:
        eo = hb["event_off"][:n + 1]; ro = hb["rank_off"][:n + 1]
        ev = hb["events"][:eo[-1]]; rk = hb["ranks"][:ro[-1]].astype(np.uint32)
    # each leg runs twice (the first call also pays thread start-up and page faults); the faster run counts, and only
    # the C call itself is timed (oracle_py.last_call_s), not the ctypes marshalling around it
    t_align = 1e30
    for _ in range(2):
        if ref:
            pairs, pair_off, n_pairs = ref.align_many([r["seq"] for r in rds], ev, eo, mom[:, 0], mom[:, 1], threads)
            t_align = min(t_align, ref.last_call_s)
        else:
            pairs, pair_off, n_pairs = orc.align_many(mn, ev, eo, rk, ro, mom[:, 0], mom[:, 1], threads)
            t_align = min(t_align, orc.last_call_s)
    # event map + window bounds through the oracle's glue (untimed host bookkeeping, tiny)
    job_read, e1, e2, stride, rcs, jr, jr_off, epb = [], [], [], [], [], [], [0], np.zeros(n)
    seqs, rc_seqs, first, job_off = [], [], [], [0]
    sh = [r["shift"] for r in rds]; sc_ = [r["scale"] for r in rds]; vr = [r["var"] for r in rds]
    t_calib = 0.0
    for i in range(n):
        p = pairs[pair_off[i]:pair_off[i] + n_pairs[i]]
        if len(p):
            epb[i], jobs = methylation_jobs(orc, rds[i], p)
            if calibrate:      # recalibrate_model on the event map (oracle restatement; serial, its time is added below)
                tc = time.perf_counter()
                start, stop, _ = orc.build_base_to_event_map(p, len(rds[i]["ranks"]))
                cal = orc.recalibrate(mn, rds[i]["events"], rds[i]["ranks"], start, stop)
                t_calib += time.perf_counter() - tc
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
    t_calib /= max(1, threads)          # as if spread over the cores like the other legs
    t_score = 1e30
    for _ in range(2):
        if ref:
            sc = ref.score_many_reads("cpg", ev, eo, sh, sc_, vr, epb, job_off, seqs, rc_seqs, e1, e2, stride, rcs, 3, threads)
            t_score = min(t_score, ref.last_call_s)
        else:
            sc = orc.score_many(mc, job_read, ev, eo, sh, sc_, vr, epb, np.concatenate(jr), jr_off, e1, e2, stride, 1.0, 3, threads)
            t_score = min(t_score, orc.last_call_s)
    return dict(n=n, seconds=t_detect + t_align + t_calib + t_score, t_align=t_align, t_score=t_score, t_calib=t_calib, t_detect=t_detect,
                n_events=[len(r["events"]) for r in rds],
                pairs=(pairs, pair_off, n_pairs),
                first=first, scores=sc, kind="reference" if ref else "port")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pool", type=int, default=1024, help="distinct synthetic reads per rank")
    ap.add_argument("--tile", type=int, default=32, help="independent HBM copies of the pool per batch (32768 reads/step by default:\n"
                    "                    per-read kernel time keeps falling up to ~100k reads per launch, tools/sweep_batch.sh)")
    ap.add_argument("--read-len", type=int, default=5450, help="bases per read (5450 -> ~8k events)")
    ap.add_argument("--calibrate", type=int, default=1,
                    help="1: recalibrate each read on the device between the two kernels, as load_from_raw does (SURVEY 8 f1); "
                         "0: score with the scalings the synthetic reads were made with")
    ap.add_argument("--from-raw", type=int, default=0,
                    help="1: a step starts from raw current samples (scrappie event detection + MoM scalings on the device, "
                         "SURVEY 8 f2); 0: from pre-detected events")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1: ~8 per core, 0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one process per GPU.  NP_BENCH_BACKEND=gloo + fewer devices than ranks is a single-GPU rehearsal of the N>1 code path
    backend = os.environ.get("NP_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from nanopolish_amd.api import Context
    from nanopolish_amd.pipeline import build_host_batch, tile_host_batch, CallMethylationBatch
    from nanopolish_amd.sites import site_table
    from nanopolish_amd.shard import shard_read_ids, reduce_site_table

    models = load_models()
    ctx = Context(local)
    ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg")
    t_prep = time.perf_counter()
    lo, hi = shard_read_ids(world * args.pool, rank, world)          # reads shard by contiguous id range
    hb = build_host_batch(models, np.arange(lo, hi), L=args.read_len, raw=bool(args.from_raw))
    hbt = tile_host_batch(hb, args.tile)
    batch = CallMethylationBatch(ctx, hbt, "cuda:%d" % local, calibrate=bool(args.calibrate), from_raw=bool(args.from_raw))
    t_prep = time.perf_counter() - t_prep
    n_reads = batch.n_reads

    # per-group metadata for the site table (device)
    first = torch.from_numpy(np.tile(np.concatenate([m["first"] for m in hb["meta"]]), args.tile).astype(np.int64)).cuda()
    n_motif = torch.from_numpy(np.tile(np.concatenate([m["n_motif"] for m in hb["meta"]]), args.tile).astype(np.int64)).cuda()

    def barrier():
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ctx.sync(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.step()
    if world > 1:
        # warm the collective path too (RCCL communicator set-up is not part of a step)
        ctx.sync()
        sc = batch.d_scores[:batch.n_jobs].to(torch.float64)
        reduce_site_table(site_table(torch, first, n_motif, sc[1::2] - sc[0::2], args.read_len))
    barrier()
    for w in range(6):
        ctx.kernel_time(w, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.step()
    ctx.sync()
    table = None
    if world > 1:
        sc = batch.d_scores[:batch.n_jobs].to(torch.float64)
        table = site_table(torch, first, n_motif, sc[1::2] - sc[0::2], args.read_len)
        reduce_site_table(table)        # RCCL all-reduce(sum): the job's only collective (final site-level reduction)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        k_ms = {}
        for name, w in (("event_align", 0), ("hmm_score", 1), ("resolve", 2), ("event_detect", 4), ("mom_scalings", 5)):
            ms, n = ctx.kernel_time(w)
            k_ms[name] = (ms, n)
        scores = batch.scores()
        llr = scores[1::2].astype(np.float64) - scores[0::2]
        n_groups = int(np.isfinite(llr).sum())
        n_ok = int((batch.d_n_pairs > 0).sum().item())

        # dominant kernel + HBM roofline (algorithmic bytes)
        dom = max(k_ms, key=lambda k: k_ms[k][0])
        a_ms, a_n = k_ms["event_align"]
        a_avg_s = a_ms / max(a_n, 1) * 1e-3
        if args.from_raw:
            # the event counts only exist after the detector has run: algorithmic bytes from the detected counts
            nev = batch.d_n_events.clamp(min=0).to(torch.int64).cpu().numpy()
            nk = (hbt["rank_off"][1:] - hbt["rank_off"][:-1])
            bands = nev + nk + 2
            batch.algo_bytes_align = int((4 * nev + 2 * nk + 100 * bands + 8 * nev).sum())
            batch.band_cells = int((100 * bands).sum()); batch.total_events = int(nev.sum())
        algo = batch.algo_bytes_align
        achieved = algo / a_avg_s / 1e9 if a_avg_s > 0 else 0.0
        # HBM traffic per launch from the PMC passes (profiles/collect_pmc.sh + profiles/pmc_summary.py -> profiles/r01_pmc.json): FETCH_SIZE +
        # WRITE_SIZE per read of this kernel, measured on this workload shape in separate rocprofv3 --pmc runs
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))["event_align"]
            traffic = int((pm["fetch_bytes_per_read"] + pm["write_bytes_per_read"]) * n_reads)
        except Exception:
            pass
        roof = dict(bound="hbm", kernel="np_event_align_kernel", achieved=round(achieved, 2), peak=8000.0, unit="GB/s",
                    frac=round(achieved / 8000.0, 5), traffic=traffic,
                    algo_bytes_per_launch=algo, avg_launch_ms=round(a_ms / max(a_n, 1), 3),
                    band_cells_per_s=round(batch.band_cells / a_avg_s / 1e9, 3) if a_avg_s > 0 else 0.0,
                    dominant_kernel_by_time=dom,
                    kernel_ms_per_step={k: round(v[0] / max(v[1], 1), 3) for k, v in k_ms.items()})

        cpu = None
        max_dllr = None
        cores = len(os.sched_getaffinity(0))
        n_sample = args.cpu_sample if args.cpu_sample >= 0 else max(8, 8 * cores)
        if n_sample > 0 and world == 1:
            cb = cpu_baseline(models, hb, n_sample, cores, bool(args.calibrate), bool(args.from_raw))
            cpu = dict(value=round(cb["n"] / cb["seconds"], 2), unit="reads/s", cores=cores, kind=cb["kind"],
                       sample="%d of the same synthetic reads, OpenMP over reads (detect %.2fs + align %.1fs + calibrate %.2fs + score %.1fs)"
                              % (cb["n"], cb["t_detect"], cb["t_align"], cb["t_calib"], cb["t_score"]))
            if args.from_raw and args.calibrate:
                # the same sample through the reference's WHOLE per-read function: SquiggleRead(sequence, Fast5Data) -> load_from_raw
                # -> calculate_methylation_for_read, compiled in place (oracle/_ref/libnp_ref_full.so), OpenMP over reads
                try:
                    from oracle.ref_full import FullRef, have_full
                    if have_full():
                        rds = hb["reads"][:cb["n"]]
                        sites, t_full = FullRef().many_identity(1, [r["seq"] for r in rds], [r["raw"] for r in rds], [r["rc"] for r in rds], cores)
                        n_gpu = [int(np.isfinite(llr[int(hb["job_off"][i]) // 2:int(hb["job_off"][i + 1]) // 2]).sum()) for i in range(cb["n"])]
                        cpu["whole_function"] = dict(value=round(cb["n"] / t_full, 2), unit="reads/s",
                                                     what="load_from_raw + calculate_methylation_for_read per read, reference code",
                                                     sites_per_read_match_gpu=bool(np.array_equal(sites, np.array(n_gpu))))
                except Exception as e:  # noqa: BLE001
                    cpu["whole_function"] = dict(error=repr(e))
            # parity of the GPU results with the oracle on that sample: pairs bit-exact, LLR within 1e-4
            pairs, pair_off, n_pairs = cb["pairs"]
            ok = True
            if args.from_raw:
                ok &= np.array_equal(batch.d_n_events[:cb["n"]].cpu().numpy(), np.array(cb["n_events"]))
            for i in range(cb["n"]):
                g = batch.pairs_of(i)
                ok &= np.array_equal(g, pairs[pair_off[i]:pair_off[i] + n_pairs[i]])
            jh = hb["job_off"]
            firsts = np.concatenate([m["first"] for m in hb["meta"]])
            gmap = {}
            gi = 0
            for i, m in enumerate(hb["meta"]):
                for f in m["first"]:
                    gmap[(i, int(f))] = gi; gi += 1
            want = cb["scores"]
            d = []
            for q, key in enumerate(cb["first"]):
                g = gmap[key]
                d.append((float(scores[2 * g + 1]) - float(scores[2 * g])) - (float(want[2 * q + 1]) - float(want[2 * q])))
            max_dllr = float(np.max(np.abs(d))) if d else 0.0
            n_gpu_groups = int(np.isfinite(llr[:sum(len(m["first"]) for m in hb["meta"][:cb["n"]])]).sum())
            cpu["check"] = dict(reads=cb["n"], groups=len(d), groups_scored_on_gpu=n_gpu_groups, pairs_bit_exact=bool(ok),
                                max_abs_dLLR=max_dllr)

        value = world * n_reads * args.steps / dt
        out = dict(metric="call-methylation reads/sec", value=round(value, 2), unit="reads/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="call-methylation, synthetic R9.4 reads (~8k events each), r9.4_450bps CpG model "
                                        "(BASELINE.json configs[1] shape)",
                               reads_per_step_per_gpu=n_reads, distinct_reads_per_gpu=args.pool, tile=args.tile,
                               read_len=args.read_len, mean_events=round(batch.total_events / n_reads, 1),
                               groups_per_step_per_gpu=n_groups, reads_aligned_ok=n_ok, calibrate_on_device=bool(args.calibrate), from_raw_signal=bool(args.from_raw),
                               parallelism="reads sharded over %d GPU(s), 1 process/GPU" % world),
                   cpg_site_groups_per_s=round(world * n_groups * args.steps / dt, 1),
                   max_abs_dLLR_vs_cpu=max_dllr, roofline=roof, cpu_baseline=cpu, host_prep_s=round(t_prep, 1))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
