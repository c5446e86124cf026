"""Device-resident call-methylation pass over a batch of reads (the north-star path):

    [from_raw=True: detect_events (scrappie) -> estimate_scalings_using_mom + aligner constants, all on the device
     (SURVEY 8 f2): the batch then starts from raw current samples instead of events]
    adaptive_banded_simple_event_align  (kernel A, one wave per read)
      -> base_to_event_map / events_per_base / transitions / window event bounds  (glue kernels)
         [calibrate=True: + recalibrate_model on the event map, src/nanopolish_methyltrain.cpp:204-306, SURVEY 8 f1]
      -> 2 x profile_hmm_score per CpG group (kernel B)

mirroring SquiggleRead::load_from_raw (src/nanopolish_squiggle_read.cpp:270-301) followed by
calculate_methylation_for_read (src/basemods/nanopolish_basemods.cpp:238-457) for reads laid out as in
SURVEY.md section 8d (identity alignment to their own sequence; odd read ids are reverse-strand).
Inputs are uploaded once; `step()` only enqueues kernels through the C ABI's *_dev entry points.
torch is used for device memory only.
"""
import ctypes as C
import os
import numpy as np

from . import lib as _l
from . import api
from .synth import synth_read, synth_raw

READ_DT = np.dtype([("scale", "<f8"), ("shift", "<f8"), ("var", "<f8"), ("log_var", "<f8"),
                    ("lp_skip", "<f8"), ("lp_stay", "<f8"), ("lp_step", "<f8"), ("lp_trim", "<f8"),
                    ("event_off", "<i8"), ("rank_off", "<i8"), ("n_events", "<u4"), ("n_kmers", "<u4"),
                    ("trans", "<f4", (10,)), ("flags", "<u4"), ("reserved", "<u4")])
JOB_DT = np.dtype([("rank_off", "<i8"), ("n_kmers", "<u4"), ("read", "<u4"), ("e_start", "<u4"), ("e_stop", "<u4"),
                   ("stride", "<i4"), ("flags", "<u4")])
assert READ_DT.itemsize == C.sizeof(_l.ReadDev) and JOB_DT.itemsize == C.sizeof(_l.HmmJobDev)

HAF = api.HAF_ALLOW_PRE_CLIP | api.HAF_ALLOW_POST_CLIP


def build_host_batch(models, read_ids, L=5450, k=6, raw=False, with_jobs=True, adc=False):
    """Host-side preparation of the distinct reads of a batch (numpy only).
    L: bases per read, one number or one per read (ragged batches).
    raw=True: reads carry synthetic raw signal (synth_raw); the event arrays are then sized as CAPACITY for the device
    detector (n_samples/2 + 2 per read) and hold no data, and the per-read records only carry offsets and n_kmers.
    with_jobs=False: no host-built work items (the device builds them, jobs_on_device=True): hb["jobs"] stays empty.
    adc=True (with raw): the traces are int16 ADC counts (hb["adc"], per-read offset / unit) and hb["raw"] holds the pA values
    they convert to; a batch built this way uploads the counts and converts on the device (np_adc_to_pa_dev)."""
    L_ = _l.load_library()
    nuc = models["nucleotide"]
    Ls = np.broadcast_to(np.asarray(L, np.int64), (len(read_ids),))
    reads = [(synth_raw(r, nuc, L=int(l), k=k, adc=adc) if raw else synth_read(r, nuc, L=int(l), k=k)) for r, l in zip(read_ids, Ls)]
    n = len(reads)
    if raw:
        for r in reads:
            r["events"] = np.zeros(len(r["raw"]) // 2 + 2, np.float32)          # capacity only
    event_off = np.zeros(n + 1, np.int64); rank_off = np.zeros(n + 1, np.int64)
    event_off[1:] = np.cumsum([len(r["events"]) for r in reads]); rank_off[1:] = np.cumsum([len(r["ranks"]) for r in reads])
    events = np.concatenate([r["events"] for r in reads]).astype(np.float32)
    ranks = np.concatenate([r["ranks"] for r in reads]).astype(np.uint16)
    reads_a = np.zeros(n, READ_DT); reads_b = np.zeros(n, READ_DT)
    mom = np.zeros((n, 2))
    jobs, kpos, jranks, meta, ref_seqs = [], [], [], [], []
    jr_off = 0
    for i, r in enumerate(reads):
        sh, sc = (0.0, 1.0) if raw else api.estimate_scalings_using_mom(nuc, r["ranks"], r["events"])
        mom[i] = (sh, sc)
        ne, nk = len(r["events"]), len(r["ranks"])
        for arr, (shift, scale, var) in ((reads_a, (sh, sc, 1.0)), (reads_b, (r["shift"], r["scale"], r["var"]))):
            L_.np_fill_read_host(C.cast(arr[i:i + 1].ctypes.data, C.POINTER(_l.ReadDev)), shift, scale, var,
                                 int(event_off[i]), ne, int(rank_off[i]), nk)
        ref_seq = api.reverse_complement("nucleotide", r["seq"]) if r["rc"] else r["seq"]
        ref_seqs.append(ref_seq)
        if not with_jobs:
            continue
        jb = api.cm_build_jobs_identity(ref_seq, r["rc"], k)
        ng = len(jb["first"])
        j = np.zeros(2 * ng, JOB_DT)
        nk_j = jb["n_kmers"].astype(np.uint32)
        ro = jb["rank_off"][:-1]
        tot = int(jb["rank_off"][-1])
        j["n_kmers"] = np.repeat(nk_j, 2)
        j["read"] = i
        j["flags"] = HAF
        j["stride"] = 1
        j["rank_off"][0::2] = jr_off + ro                 # unmethylated copy
        j["rank_off"][1::2] = jr_off + tot + ro           # methylated copy
        jobs.append(j)
        kpos.append(np.repeat(jb["kpos"], 2, axis=0))
        jranks.append(jb["ranks_unmeth"]); jranks.append(jb["ranks_meth"])
        jr_off += 2 * tot
        meta.append(dict(first=jb["first"], last=jb["last"], n_motif=jb["n_motif"]))
    extra = {}
    if raw:
        raw_off = np.zeros(n + 1, np.int64); raw_off[1:] = np.cumsum([len(r["raw"]) for r in reads])
        extra = dict(raw=np.concatenate([r["raw"] for r in reads]).astype(np.float32), raw_off=raw_off)
        if adc:
            from .synth import ADC_OFFSET, ADC_UNIT
            extra.update(adc=np.concatenate([r["adc"] for r in reads]), adc_offset=np.full(n, ADC_OFFSET, np.float32),
                         adc_unit=np.full(n, ADC_UNIT, np.float32))
    return dict(reads=reads, n=n, events=events, ranks=ranks, event_off=event_off, rank_off=rank_off,
                reads_a=reads_a, reads_b=reads_b, mom=mom, ref_seqs=ref_seqs, k=k, **extra,
                jobs=np.concatenate(jobs) if jobs else np.zeros(0, JOB_DT),
                kpos=np.concatenate(kpos).astype(np.int32) if kpos else np.zeros((0, 2), np.int32),
                job_ranks=np.concatenate(jranks).astype(np.uint16) if jranks else np.zeros(0, np.uint16),
                job_off=np.concatenate([[0], np.cumsum([len(j) for j in jobs])]).astype(np.int64), meta=meta)


def concat_host_batches(parts):
    """Join host batches built for consecutive read-id ranges (identity layout, e.g. by a pool of worker processes)."""
    if len(parts) == 1:
        return parts[0]
    out = dict(parts[0])
    ne = np.cumsum([0] + [len(p["events"]) for p in parts]); nr = np.cumsum([0] + [len(p["ranks"]) for p in parts])
    nj = np.cumsum([0] + [len(p["jobs"]) for p in parts]); njr = np.cumsum([0] + [len(p["job_ranks"]) for p in parts])
    out["n"] = sum(p["n"] for p in parts)
    out["reads"] = [r for p in parts for r in p["reads"]]
    out["ref_seqs"] = [r for p in parts for r in p["ref_seqs"]]
    out["meta"] = [m for p in parts for m in p["meta"]]
    for key in ("events", "ranks", "job_ranks", "kpos", "mom"):
        out[key] = np.concatenate([p[key] for p in parts])
    for key in ("reads_a", "reads_b"):
        a = [p[key].copy() for p in parts]
        for i, x in enumerate(a):
            x["event_off"] += ne[i]; x["rank_off"] += nr[i]
        out[key] = np.concatenate(a)
    jb = [p["jobs"].copy() for p in parts]
    nread = np.cumsum([0] + [p["n"] for p in parts])
    for i, x in enumerate(jb):
        x["read"] += np.uint32(nread[i]); x["rank_off"] += njr[i]
    out["jobs"] = np.concatenate(jb)
    out["event_off"] = np.concatenate([p["event_off"][:-1] + ne[i] for i, p in enumerate(parts)] + [[ne[-1]]]).astype(np.int64)
    out["rank_off"] = np.concatenate([p["rank_off"][:-1] + nr[i] for i, p in enumerate(parts)] + [[nr[-1]]]).astype(np.int64)
    out["job_off"] = np.concatenate([p["job_off"][:-1] + nj[i] for i, p in enumerate(parts)] + [[nj[-1]]]).astype(np.int64)
    if "raw" in out:
        ns = np.cumsum([0] + [len(p["raw"]) for p in parts])
        out["raw"] = np.concatenate([p["raw"] for p in parts])
        out["raw_off"] = np.concatenate([p["raw_off"][:-1] + ns[i] for i, p in enumerate(parts)] + [[ns[-1]]]).astype(np.int64)
    if "adc" in out:
        for key in ("adc", "adc_offset", "adc_unit"):
            out[key] = np.concatenate([p[key] for p in parts])
    return out


def concat_record_batches(parts):
    """Join record batches (build_host_batch_records) built against the SAME contig(s), e.g. by a pool of worker processes: the genome stays
    one copy, every per-read array is concatenated and the offsets shifted."""
    if len(parts) == 1:
        return parts[0]
    out = concat_host_batches([{k_: v for k_, v in p.items()} for p in parts])
    nc = np.cumsum([0] + [len(p["cigar"]) for p in parts])
    out["cigar"] = np.concatenate([p["cigar"] for p in parts])
    out["cigar_off"] = np.concatenate([p["cigar_off"][:-1] + nc[i] for i, p in enumerate(parts)] + [[nc[-1]]]).astype(np.int64)
    for key in ("ref_begin", "ref_len", "read_len", "deg_kpos"):
        out[key] = np.concatenate([p[key] for p in parts])
    assert all(len(p["genome"]) == len(parts[0]["genome"]) for p in parts), "record batches of different contigs"
    return out


def build_host_batch_records(models, records, contig, k=6, alphabet="cpg", with_jobs=True):
    """Host-side preparation of a batch of reads given EXPLICITLY, each with its raw signal and the BAM record of its
    base-to-reference alignment: records = dicts(seq: the read's own sequence, raw: float32 samples, rc: bam_is_rev,
    pos: 0-based leftmost reference position, cigar: uint32 BAM words[, contig: this record's own reference][, adc: the int16 counts `raw`
    is the conversion of (synth.adc_quantise): when every record has them the batch uploads counts]).  contig: the reference the records align
    to (those without their own).
    The batch starts from raw signal (from_raw) and its work items follow the CIGARs: the host builder's
    items are in hb["jobs"] / hb["kpos"]; the device builder needs hb["cigar"], hb["ref_begin"], ... (same numbering).
    with_jobs=False: no methylation work items (an eventalign-only batch, e.g. direct-RNA reads: k = 5, base model u_to_t_rna).
    Round 6: a record may carry pre-detected `events` (float32 means, with its `shift` / `scale` / `var`) INSTEAD of `raw`: the batch then starts
    from events (from_raw=False), MoM scalings taken here on the host as build_host_batch does -- the call-methylation step of the bench on reads
    that have a place on a genome."""
    L_ = _l.load_library()
    from .synth import nucleotide_kmer_ranks
    lut = np.zeros(256, np.int64); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
    reads = []
    contigs, contig_base = [], {}
    for r in records:
        cs = r.get("contig", contig)
        if cs not in contig_base:
            contig_base[cs] = sum(len(x) for x in contigs); contigs.append(cs)
    from_events = len(records) > 0 and "events" in records[0]
    for r in records:
        codes = lut[np.frombuffer(r["seq"].encode(), np.uint8)]
        if from_events:
            reads.append(dict(seq=r["seq"], rc=bool(r["rc"]), ranks=nucleotide_kmer_ranks(codes, k), events=np.ascontiguousarray(r["events"], np.float32),
                              shift=float(r.get("shift", 0.0)), scale=float(r.get("scale", 1.0)), var=float(r.get("var", 1.0)),
                              pos=int(r["pos"]), cigar=np.ascontiguousarray(r["cigar"], np.uint32), contig=r.get("contig", contig)))
            continue
        reads.append(dict(seq=r["seq"], rc=bool(r["rc"]), raw=np.ascontiguousarray(r["raw"], np.float32), ranks=nucleotide_kmer_ranks(codes, k),
                          events=np.zeros(len(r["raw"]) // 2 + 2, np.float32), shift=0.0, scale=1.0, var=1.0,
                          pos=int(r["pos"]), cigar=np.ascontiguousarray(r["cigar"], np.uint32), contig=r.get("contig", contig)))
    n = len(reads)
    event_off = np.zeros(n + 1, np.int64); rank_off = np.zeros(n + 1, np.int64)
    event_off[1:] = np.cumsum([len(r["events"]) for r in reads]); rank_off[1:] = np.cumsum([len(r["ranks"]) for r in reads])
    reads_a = np.zeros(n, READ_DT); reads_b = np.zeros(n, READ_DT)
    jobs, kpos, jranks, meta, ref_seqs = [], [], [], [], []
    deg = np.zeros((n, 2), np.int32)
    ref_begin = np.zeros(n, np.int64); ref_len = np.zeros(n, np.int32)
    jr_off = 0
    mom = np.zeros((n, 2))
    for i, r in enumerate(reads):
        sh, sc = api.estimate_scalings_using_mom(models["nucleotide"], r["ranks"], r["events"]) if from_events else (0.0, 1.0)
        mom[i] = (sh, sc)
        for arr, (shift, scale, var) in ((reads_a, (sh, sc, 1.0)), (reads_b, (r["shift"], r["scale"], r["var"]))):
            L_.np_fill_read_host(C.cast(arr[i:i + 1].ctypes.data, C.POINTER(_l.ReadDev)), shift, scale, var,
                                 int(event_off[i]), len(r["events"]), int(rank_off[i]), len(r["ranks"]))
        # the segment calculate_methylation_for_read fetches: contig[pos .. bam_endpos] inclusive, clipped (basemods.cpp:259-270)
        span = int(sum(int(w) >> 4 for w in r["cigar"] if (int(w) & 0xf) in (0, 2, 3, 7, 8)))
        endpos = r["pos"] + (span if span > 0 else 1)
        seg = r["contig"][r["pos"]:min(endpos + 1, len(r["contig"]))]
        ref_seqs.append(seg); ref_begin[i] = contig_base[r["contig"]] + r["pos"]; ref_len[i] = len(seg)
        try:
            if not with_jobs:
                raise ValueError("no work items wanted")
            jb = api.cm_build_jobs_cigar(seg, r["cigar"], len(r["seq"]), r["rc"], k, alphabet)
        except ValueError:            # a record the reference refuses (spliced / padded CIGAR): no work items, as on the device
            z = np.zeros(0, np.int32)
            jb = dict(first=z, last=z, n_motif=z, kpos=np.zeros((0, 2), np.int32), n_kmers=z, ranks_unmeth=np.zeros(0, np.uint16),
                      ranks_meth=np.zeros(0, np.uint16), rank_off=np.zeros(1, np.int64), deg_kpos=np.array([-1, -1], np.int32))
        deg[i] = jb["deg_kpos"]
        ng = len(jb["first"])
        j = np.zeros(2 * ng, JOB_DT)
        ro = jb["rank_off"][:-1]
        tot = int(jb["rank_off"][-1])
        j["n_kmers"] = np.repeat(jb["n_kmers"].astype(np.uint32), 2)
        j["read"] = i; j["flags"] = HAF; j["stride"] = 1
        j["rank_off"][0::2] = jr_off + ro
        j["rank_off"][1::2] = jr_off + tot + ro
        jobs.append(j)
        kpos.append(np.repeat(jb["kpos"], 2, axis=0))
        jranks.append(jb["ranks_unmeth"]); jranks.append(jb["ranks_meth"])
        jr_off += 2 * tot
        meta.append(dict(first=jb["first"], last=jb["last"], n_motif=jb["n_motif"]))
    cigar_off = np.zeros(n + 1, np.int64); cigar_off[1:] = np.cumsum([len(r["cigar"]) for r in reads])
    extra = {}
    if not from_events:
        raw_off = np.zeros(n + 1, np.int64); raw_off[1:] = np.cumsum([len(r["raw"]) for r in reads])
        extra = dict(raw=np.concatenate([r["raw"] for r in reads]), raw_off=raw_off)
        if n > 0 and all("adc" in r for r in records):
            # the traces as a sequencer stores them (int16 counts, synth.adc_quantise: "raw" is what they convert to): the batch uploads the counts
            from .synth import ADC_OFFSET, ADC_UNIT
            extra.update(adc=np.concatenate([np.ascontiguousarray(r["adc"], np.int16) for r in records]), adc_offset=np.full(n, ADC_OFFSET, np.float32),
                         adc_unit=np.full(n, ADC_UNIT, np.float32))
    contig_off = np.concatenate([[0], np.cumsum([len(x) for x in contigs])]).astype(np.int64)
    return dict(reads=reads, n=n, events=np.concatenate([r["events"] for r in reads]), ranks=np.concatenate([r["ranks"] for r in reads]).astype(np.uint16),
                event_off=event_off, rank_off=rank_off, reads_a=reads_a, reads_b=reads_b, mom=mom, ref_seqs=ref_seqs, k=k, contig_off=contig_off, **extra,
                jobs=np.concatenate(jobs) if jobs else np.zeros(0, JOB_DT),
                kpos=np.concatenate(kpos).astype(np.int32) if kpos else np.zeros((0, 2), np.int32),
                job_ranks=np.concatenate(jranks).astype(np.uint16) if jranks else np.zeros(0, np.uint16),
                job_off=np.concatenate([[0], np.cumsum([len(j) for j in jobs])]).astype(np.int64), meta=meta,
                genome=np.frombuffer("".join(contigs).encode(), np.uint8).copy(), ref_begin=ref_begin, ref_len=ref_len,
                cigar=np.concatenate([r["cigar"] for r in reads]).astype(np.uint32), cigar_off=cigar_off,
                read_len=np.array([len(r["seq"]) for r in reads], np.int32), deg_kpos=deg, alphabet=alphabet)


def tile_host_batch(hb, tile):
    """Replicate the distinct reads `tile` times (independent copies in HBM, shifted offsets)."""
    if tile == 1:
        return hb
    n, ne, nr, nj, njr = hb["n"], len(hb["events"]), len(hb["ranks"]), len(hb["jobs"]), len(hb["job_ranks"])
    out = dict(hb)
    out["n"] = n * tile
    out["events"] = np.tile(hb["events"], tile); out["ranks"] = np.tile(hb["ranks"], tile)
    out["job_ranks"] = np.tile(hb["job_ranks"], tile); out["kpos"] = np.tile(hb["kpos"], (tile, 1))
    for key in ("reads_a", "reads_b"):
        a = np.tile(hb[key], tile)
        a["event_off"] += np.repeat(np.arange(tile, dtype=np.int64) * ne, n)
        a["rank_off"] += np.repeat(np.arange(tile, dtype=np.int64) * nr, n)
        out[key] = a
    j = np.tile(hb["jobs"], tile)
    j["read"] += np.repeat(np.arange(tile, dtype=np.uint32) * n, nj).astype(np.uint32)
    j["rank_off"] += np.repeat(np.arange(tile, dtype=np.int64) * njr, nj)
    out["jobs"] = j
    out["ref_seqs"] = hb["ref_seqs"] * tile
    if "raw" in hb:
        ns = len(hb["raw"])
        out["raw"] = np.tile(hb["raw"], tile)
        out["raw_off"] = np.concatenate([hb["raw_off"][:-1] + t * ns for t in range(tile)] + [[ns * tile]]).astype(np.int64)
        if "adc" in hb:
            out["adc"] = np.tile(hb["adc"], tile); out["adc_offset"] = np.tile(hb["adc_offset"], tile); out["adc_unit"] = np.tile(hb["adc_unit"], tile)
    if "cigar" in hb:
        nc = len(hb["cigar"])
        out["cigar"] = np.tile(hb["cigar"], tile)
        out["cigar_off"] = np.concatenate([hb["cigar_off"][:-1] + t * nc for t in range(tile)] + [[nc * tile]]).astype(np.int64)
        for key in ("ref_begin", "ref_len", "read_len"):          # every copy points at the same resident contig(s)
            out[key] = np.tile(hb[key], tile)
        out["deg_kpos"] = np.tile(hb["deg_kpos"], (tile, 1))
    out["event_off"] = np.concatenate([hb["event_off"][:-1] + t * ne for t in range(tile)] + [[ne * tile]]).astype(np.int64)
    out["rank_off"] = np.concatenate([hb["rank_off"][:-1] + t * nr for t in range(tile)] + [[nr * tile]]).astype(np.int64)
    return out


class CallMethylationBatch:
    def __init__(self, ctx, hb, device="cuda:0", calibrate=False, from_raw=False, jobs_on_device=False, workload="call-methylation", rna=False,
                 base_model="nucleotide", map_stop=True, adc_one_call=True):
        """calibrate=False: kernel B scores with the scalings the caller put in hb["reads_b"] (a read whose
        calibration was done elsewhere).  calibrate=True: the pass recalibrates every read on the device from its
        own event alignment, as load_from_raw does (squiggle_read.cpp:304-323), and reads_b is overwritten.
        rna=True (from_raw, workload "eventalign"): direct-RNA reads as load_from_raw treats them (squiggle_read.cpp:206-213,260-263):
        the RNA detector parameters, the events reversed after the MoM scalings, the base model registered as `base_model`
        (r9.4_70bps / u_to_t_rna / 5-mers; hb built with k = 5).
        map_stop=False (calibrate=True): base_to_event_map[].stop is not built -- recalibration and the window bounds read .start only
        (squiggle_read.cpp:161-186,339-389); the eventalign workload, which hands the map back to the reference, always builds it.
        adc_one_call (a batch of ADC counts): np_detect_events_adc_dev; False: np_adc_to_pa_checked_dev + np_detect_events_checked_dev (same events)."""
        import torch
        self.torch = torch
        self.calibrate = bool(calibrate)
        self.from_raw = bool(from_raw)
        self.rna = bool(rna)
        assert not self.rna or (self.from_raw and workload == "eventalign"), "rna=True: from raw signal, eventalign workload"
        self.from_adc = False
        self.adc_one_call = bool(adc_one_call)
        self.jobs_on_device = bool(jobs_on_device)
        self.workload = workload       # "eventalign": a step ends with the segment chain instead of the methylation scoring
        self.by_cigar = "cigar" in hb          # work items follow BAM CIGARs (build_host_batch_records)
        self.ctx = ctx
        self.hb = hb
        self.stream = None         # raw hipStream_t the step is enqueued on (None: the context's own stream)
        self.use_job_layout = os.environ.get("NP_JOB_LAYOUT", "1") != "0"     # tests: the same pass with and without np_set_job_layout
        self.n_reads = hb["n"]
        self.n_jobs = len(hb["jobs"])
        dev = torch.device(device)

        def up(a):
            a = np.ascontiguousarray(a)
            return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)

        self.d_events = up(hb["events"]); self.d_ranks = up(hb["ranks"])
        if self.from_raw:
            # raw signal in, events out: the detector writes event means into d_events at the capacity offsets
            self.from_adc = "adc" in hb          # int16 ADC counts in, converted to pA on the device at the head of the step
            if self.from_adc:
                self.d_adc = up(hb["adc"]); self.d_adc_offset = up(hb["adc_offset"]); self.d_adc_unit = up(hb["adc_unit"])
                self.d_raw = torch.empty(len(hb["raw"]) * 4, dtype=torch.uint8, device=dev)
            else:
                self.d_raw = up(hb["raw"])
            self.d_raw_off = up(hb["raw_off"]); self.d_event_off = up(hb["event_off"])
            ns = hb["raw_off"][1:] - hb["raw_off"][:-1]
            self.max_samples = int(ns.max()); self.total_samples = int(ns.sum())
            ecap = hb["event_off"][1:] - hb["event_off"][:-1]
            self.max_events = int(ecap.max())
            nev = int(hb["event_off"][-1])
            self.d_tstat = torch.empty(self.total_samples * 2 + 16, dtype=torch.float32, device=dev)
            self.d_ev_start = torch.empty(nev, dtype=torch.int32, device=dev)
            self.d_ev_len = torch.empty(nev, dtype=torch.float32, device=dev)
            self.d_ev_stdv = torch.empty(nev, dtype=torch.float32, device=dev)
            self.d_n_events = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
            # the conversion's by-product, handed to the detector explicitly (np_adc_to_pa_checked_dev -> np_detect_events_checked_dev):
            # this batch owns both calls and nothing edits the samples in between
            self.d_ed_verdict = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
            self.prm = _l.DetectorParam(); ctx.L.np_event_detection_params(C.byref(self.prm), 1 if self.rna else 0)
        self.d_reads_a = up(hb["reads_a"]); self.d_reads_b = up(hb["reads_b"])
        if self.jobs_on_device:
            # work items are generated on the device from the reads' reference strands: group slots at
            # per-read capacity offsets (groups are > min_separation apart), two work items per slot
            MINSEP, FLANK = 10, 10
            seqs = hb["ref_seqs"]
            ln = np.asarray(hb["ref_len"], np.int64) if self.by_cigar else np.array([len(q) for q in seqs], np.int64)

            self.seq_off = np.zeros(self.n_reads + 1, np.int64); self.seq_off[1:] = np.cumsum(ln)
            gcap = ln // (MINSEP + 1) + 2
            self.group_off = np.zeros(self.n_reads + 1, np.int64); self.group_off[1:] = np.cumsum(gcap)
            rcap = 2 * (ln + (2 * FLANK + 1) * gcap)
            jr_off = np.zeros(self.n_reads + 1, np.int64); jr_off[1:] = np.cumsum(rcap)
            self.n_slots = int(self.group_off[-1])
            self.n_jobs = 2 * self.n_slots
            if not self.by_cigar:          # (the CIGAR builder reads the resident contigs through ref_begin / ref_len)
                self.d_seq = up(np.frombuffer("".join(seqs).encode(), np.uint8).copy()); self.d_seq_off = up(self.seq_off)
            self.d_rc = up(np.array([r["rc"] for r in hb["reads"]] * (self.n_reads // len(hb["reads"])), np.uint8))
            self.d_group_off = up(self.group_off); self.d_jr_off = up(jr_off)
            self.d_jobs = torch.zeros(self.n_jobs * JOB_DT.itemsize, dtype=torch.uint8, device=dev)
            self.d_kpos = torch.zeros(2 * self.n_jobs, dtype=torch.int32, device=dev)
            self.d_job_ranks = torch.zeros(int(jr_off[-1]), dtype=torch.int16, device=dev)
            self.d_first = torch.zeros(self.n_slots, dtype=torch.int32, device=dev)
            self.d_last = torch.zeros(self.n_slots, dtype=torch.int32, device=dev)
            self.d_n_motif = torch.zeros(self.n_slots, dtype=torch.int32, device=dev)
            self.d_n_groups = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
            self.cm = (MINSEP, FLANK, int(hb.get("k", 6)))
        else:
            self.d_jobs = up(hb["jobs"]); self.d_kpos = up(hb["kpos"]); self.d_job_ranks = up(hb["job_ranks"])
        if self.by_cigar:
            # the contig(s) stay resident; reads carry (offset, length) of their reference segment and their CIGAR
            self.d_genome = up(hb["genome"]); self.d_ref_begin = up(hb["ref_begin"]); self.d_ref_len = up(hb["ref_len"])
            self.d_cigar = up(hb["cigar"]); self.d_cigar_off = up(hb["cigar_off"]); self.d_read_len = up(hb["read_len"])
            self.n_cigar_ops = int(hb["cigar_off"][-1])
            if not self.jobs_on_device:
                self.d_rc = up(np.array([r["rc"] for r in hb["reads"]] * (self.n_reads // len(hb["reads"])), np.uint8))
            self.d_deg = up(hb["deg_kpos"]) if not self.jobs_on_device else torch.zeros(2 * self.n_reads, dtype=torch.int32, device=dev)
        ne = (hb["event_off"][1:] - hb["event_off"][:-1]); nk = (hb["rank_off"][1:] - hb["rank_off"][:-1])
        bands = ne + nk + 2
        self.max_bands = int(bands.max())
        pair_off = np.zeros(self.n_reads + 1, np.int64); pair_off[1:] = np.cumsum(bands)
        self.pair_off = pair_off
        self.d_pair_off = up(pair_off)
        self.d_pairs = torch.empty(int(pair_off[-1]) * 8, dtype=torch.uint8, device=dev)
        self.d_pair_begin = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
        self.d_n_pairs = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
        self.d_map = torch.empty(len(hb["ranks"]), dtype=torch.int32, device=dev)
        self.d_epb = torch.zeros(self.n_reads, dtype=torch.float64, device=dev)
        self.want_map_stop = self.calibrate and (bool(map_stop) or workload == "eventalign")
        self.d_map_stop = torch.empty(len(hb["ranks"]) if self.want_map_stop else 1, dtype=torch.int32, device=dev)
        self.d_calibrated = torch.ones(self.n_reads, dtype=torch.int32, device=dev)
        self.d_scores = torch.zeros(max(self.n_jobs, 1), dtype=torch.float32, device=dev)
        self.alphabet = hb.get("alphabet", "cpg")      # the methylation alphabet whose model scores the work items
        self.m_nuc = ctx.models[base_model]; self.m_cpg = ctx.models.get(self.alphabet, -1) if workload == "eventalign" else ctx.models[self.alphabet]
        torch.cuda.synchronize()
        # algorithmic bytes of one pass: kernel A 4E + 2K + 100(E+K+2) + 8E per read,
        # kernel B 4e + 2n + 12n + 4 per call (e, n of every work item are only known after the pass; use the
        # window sizes: n exact, e ~ events_per_kmer * n)
        self.algo_bytes_align = int((4 * ne + 2 * nk + 100 * bands + 8 * ne).sum())
        self.band_cells = int((100 * bands).sum())
        self.total_events = int(ne.sum())

    def step(self, stage=0):
        """One pass.  stage 0: everything.  Stages for callers that pipeline batches over streams (set self.stream before each):
        1: work items + (event detection +) event alignment;  2: calibration, window bounds and scoring (the aligner is bound by
        vector issue and uses no LDS, the scorer by vector issue too, its LDS look-ups the largest part)."""
        L, h = self.ctx.L, self.ctx.h
        p = lambda t: C.c_void_p(t.data_ptr())
        s = C.c_void_p(self.stream) if self.stream else None      # raw hipStream_t (0 / None: the context's own stream)
        ea = self.workload == "eventalign"
        n_jobs = 0 if ea else self.n_jobs
        # the slot layout of this batch's work items (np_set_job_layout): the kernels between the builder and the scorer visit live items
        # only.  Declared per call -- a context serves several batches -- and cleared on the way out.
        slots = self.jobs_on_device and not ea and self.use_job_layout
        if slots:
            self.ctx._chk(L.np_set_job_layout(h, self.n_reads, p(self.d_group_off), p(self.d_n_groups), self.n_slots), "np_set_job_layout")
        try:
            return self._step(L, h, p, s, ea, n_jobs, stage)
        finally:
            if slots:
                L.np_set_job_layout(h, 0, None, None, 0)

    def _step(self, L, h, p, s, ea, n_jobs, stage):
        if stage == 2:
            self._step_glue(L, h, p, s, ea, n_jobs)
            return self._step_hmm(L, h, p, s, ea)
        self._step_work_items(L, h, p, s, ea)
        if self.from_raw:
            if self.from_adc and self.adc_one_call:
                # counts in, events out: the long reads are never written as pA values (np_detect_events_adc_dev)
                rc = L.np_detect_events_adc_dev(h, s, self.n_reads, p(self.d_adc), p(self.d_raw_off), self.max_samples, p(self.d_adc_offset),
                                                p(self.d_adc_unit), p(self.d_raw), C.byref(self.prm), p(self.d_tstat), p(self.d_event_off),
                                                self.max_events, p(self.d_ev_start), p(self.d_ev_len), p(self.d_events), p(self.d_ev_stdv),
                                                p(self.d_n_events))
                self.ctx._chk(rc, "np_detect_events_adc_dev")
            else:
                if self.from_adc:
                    rc = L.np_adc_to_pa_checked_dev(h, s, self.n_reads, p(self.d_adc), p(self.d_raw_off), self.max_samples, p(self.d_adc_offset),
                                                    p(self.d_adc_unit), p(self.d_raw), p(self.d_ed_verdict))
                    self.ctx._chk(rc, "np_adc_to_pa_checked_dev")
                rc = L.np_detect_events_checked_dev(h, s, self.n_reads, p(self.d_raw), p(self.d_raw_off), self.max_samples, C.byref(self.prm),
                                                    p(self.d_tstat), p(self.d_event_off), self.max_events, p(self.d_ev_start), p(self.d_ev_len),
                                                    p(self.d_events), p(self.d_ev_stdv), p(self.d_n_events),
                                                    p(self.d_ed_verdict) if self.from_adc else None)
                self.ctx._chk(rc, "np_detect_events_checked_dev")
            rc = L.np_mom_fill_dev(h, s, self.n_reads, p(self.d_reads_a), p(self.d_reads_b), p(self.d_events), p(self.d_n_events),
                                   p(self.d_ranks), self.m_nuc)
            self.ctx._chk(rc, "np_mom_fill_dev")
            if self.rna:          # 3' -> 5' signal: the events run along the sequence from here on (squiggle_read.cpp:260-263)
                rc = L.np_reverse_events_dev(h, s, self.n_reads, p(self.d_event_off), p(self.d_n_events), p(self.d_ev_start), p(self.d_ev_len),
                                             p(self.d_events), p(self.d_ev_stdv))
                self.ctx._chk(rc, "np_reverse_events_dev")
        rc = L.np_event_align_dev(h, s, self.n_reads, p(self.d_reads_a), p(self.d_events), p(self.d_ranks), self.m_nuc,
                                  self.max_bands, p(self.d_pair_off), p(self.d_pairs), p(self.d_pair_begin), p(self.d_n_pairs))
        self.ctx._chk(rc, "np_event_align_dev")
        if stage == 1:
            return
        self._step_glue(L, h, p, s, ea, n_jobs)
        return self._step_hmm(L, h, p, s, ea)

    def _step_work_items(self, L, h, p, s, ea):
        if ea:
            pass
        elif self.jobs_on_device and self.by_cigar:
            rc = L.np_cm_build_jobs_cigar_dev(h, s, self.n_reads, p(self.d_genome), p(self.d_ref_begin), p(self.d_ref_len), p(self.d_cigar),
                                              p(self.d_cigar_off), self.n_cigar_ops, p(self.d_read_len), p(self.d_rc), api.alphabet_id(self.alphabet),
                                              self.cm[2], self.cm[0], self.cm[1], p(self.d_group_off), self.n_slots, p(self.d_jr_off),
                                              p(self.d_jobs), p(self.d_kpos), p(self.d_job_ranks), p(self.d_first), p(self.d_last),
                                              p(self.d_n_motif), p(self.d_n_groups), p(self.d_deg))
            self.ctx._chk(rc, "np_cm_build_jobs_cigar_dev")
        elif self.jobs_on_device:
            rc = L.np_cm_build_jobs_identity_dev(h, s, self.n_reads, p(self.d_seq), p(self.d_seq_off), p(self.d_rc), api.alphabet_id("cpg"),
                                                 self.cm[2], self.cm[0], self.cm[1], p(self.d_group_off), self.n_slots, p(self.d_jr_off),
                                                 p(self.d_jobs), p(self.d_kpos), p(self.d_job_ranks), p(self.d_first), p(self.d_last),
                                                 p(self.d_n_motif), p(self.d_n_groups))
            self.ctx._chk(rc, "np_cm_build_jobs_identity_dev")

    def _step_glue(self, L, h, p, s, ea, n_jobs):
        if self.calibrate:
            rc = L.np_calibrate_resolve_dev(h, s, self.n_reads, p(self.d_reads_b), p(self.d_events), p(self.d_ranks),
                                            self.m_nuc, p(self.d_pair_off), p(self.d_pairs), p(self.d_pair_begin),
                                            p(self.d_n_pairs), p(self.d_map), p(self.d_map_stop) if self.want_map_stop else None, p(self.d_epb),
                                            p(self.d_calibrated), n_jobs, p(self.d_jobs), p(self.d_kpos))
            self.ctx._chk(rc, "np_calibrate_resolve_dev")
        else:
            rc = L.np_resolve_jobs_dev(h, s, self.n_reads, p(self.d_reads_b), p(self.d_pair_off), p(self.d_pairs),
                                       p(self.d_pair_begin), p(self.d_n_pairs), p(self.d_map), p(self.d_epb), n_jobs,
                                       p(self.d_jobs), p(self.d_kpos))
            self.ctx._chk(rc, "np_resolve_jobs_dev")
        if not ea and self.by_cigar:
            rc = L.np_cm_discard_degenerate_dev(h, s, p(self.d_reads_b), p(self.d_map), p(self.d_deg), self.n_jobs, p(self.d_jobs))
            self.ctx._chk(rc, "np_cm_discard_degenerate_dev")

    def _step_hmm(self, L, h, p, s, ea):
        if ea:
            self.eventalign_enqueue()
            return
        rc = L.np_hmm_score_dev(h, s, self.n_jobs, p(self.d_jobs), p(self.d_reads_b), p(self.d_events), p(self.d_job_ranks),
                                self.m_cpg, p(self.d_scores))
        self.ctx._chk(rc, "np_hmm_score_dev")

    def eventalign(self):
        """align_read_to_ref for every read of the batch (np_eventalign_dev), after step(): the segment chain of
        profile_hmm_align calls under the base model.  Returns a list of dicts(ref_position (absolute, record pos added),
        event_idx, hmm_state, status, n_calls) per read.  Needs a record-based batch (build_host_batch_records)."""
        self.eventalign_enqueue()
        return self.eventalign_results()

    def eventalign_enqueue(self):
        assert self.by_cigar, "eventalign needs BAM records (build_host_batch_records)"
        torch = self.torch
        L, h = self.ctx.L, self.ctx.h
        p = lambda t: C.c_void_p(t.data_ptr())
        s = C.c_void_p(self.stream) if self.stream else None
        dev = self.d_events.device
        if not hasattr(self, "d_ea_off"):
            ecap = (self.hb["event_off"][1:] - self.hb["event_off"][:-1]) + 1
            self.ea_off = np.zeros(self.n_reads + 1, np.int64); self.ea_off[1:] = np.cumsum(ecap)
            tot = int(self.ea_off[-1])
            self.d_ea_off = torch.from_numpy(self.ea_off).to(dev)
            self.d_ea_ref = torch.zeros(tot, dtype=torch.int32, device=dev); self.d_ea_event = torch.zeros(tot, dtype=torch.int32, device=dev)
            self.d_ea_state = torch.zeros(tot, dtype=torch.uint8, device=dev)
            self.d_ea_n = torch.zeros(self.n_reads, dtype=torch.int32, device=dev); self.d_ea_status = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
            self.d_ea_calls = torch.zeros(self.n_reads, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()       # the zero-fills ran on torch's stream; the kernels below run on the library's
        rc = L.np_eventalign_dev(h, s, self.n_reads, p(self.d_reads_b), p(self.d_events), p(self.d_map), p(self.d_n_pairs), p(self.d_epb),
                                 p(self.d_calibrated) if self.calibrate else None, self.m_nuc, p(self.d_genome), p(self.d_ref_begin),
                                 p(self.d_ref_len), p(self.d_cigar), p(self.d_cigar_off), self.n_cigar_ops, p(self.d_read_len), p(self.d_rc),
                                 int(self.hb.get("k", 6)), p(self.d_ea_off), p(self.d_ea_ref), p(self.d_ea_event), p(self.d_ea_state),
                                 p(self.d_ea_n), p(self.d_ea_status), p(self.d_ea_calls))
        self.ctx._chk(rc, "np_eventalign_dev")

    def eventalign_results(self):
        self.sync()
        n = self.d_ea_n.cpu().numpy(); st = self.d_ea_status.cpu().numpy(); nc = self.d_ea_calls.cpu().numpy()
        ref, ev, hs = self.d_ea_ref.cpu().numpy(), self.d_ea_event.cpu().numpy(), self.d_ea_state.cpu().numpy()
        out = []
        for i in range(self.n_reads):
            lo = int(self.ea_off[i]); m = int(n[i])
            out.append(dict(ref_position=ref[lo:lo + m] + self.hb["reads"][i % len(self.hb["reads"])]["pos"], event_idx=ev[lo:lo + m].copy(),
                            hmm_state=hs[lo:lo + m].copy(), status=int(st[i]), n_calls=int(nc[i])))
        return out

    def sync(self):
        self.ctx.sync(C.c_void_p(self.stream) if self.stream else None)

    # ---- results on the host (for checking) -------------------------------------------------------------
    def scores(self):
        self.sync()
        return self.d_scores[:self.n_jobs].cpu().numpy()

    def pairs_of(self, r):
        self.sync()
        nb = int(self.d_pair_begin[r]); n = int(self.d_n_pairs[r])
        lo = int(self.pair_off[r]) + nb
        return self.d_pairs[lo * 8:(lo + n) * 8].cpu().numpy().view(np.int32).reshape(-1, 2)

    def jobs_host(self):
        self.sync()
        return self.d_jobs.cpu().numpy().view(JOB_DT)

    def epb(self):
        self.sync()
        return self.d_epb.cpu().numpy()

    def reads_scored(self):
        """np_read_dev records kernel B used (after device calibration when calibrate=True)."""
        self.sync()
        return self.d_reads_b.cpu().numpy().view(READ_DT)

    def detected(self, r):
        """(n_events, start, length, mean, stdv) of read r as the device detector left them (from_raw=True)."""
        self.sync()
        n = int(self.d_n_events[r]); lo = int(self.hb["event_off"][r])
        f = lambda t, dt: t[lo:lo + max(n, 0)].cpu().numpy().view(dt)
        return n, f(self.d_ev_start, np.uint32), f(self.d_ev_len, np.float32), f(self.d_events.view(self.torch.float32), np.float32), \
            f(self.d_ev_stdv, np.float32)

    def reads_aligned(self):
        """np_read_dev records kernel A used (MoM scalings + aligner constants; filled on the device when from_raw=True)."""
        self.sync()
        return self.d_reads_a.cpu().numpy().view(READ_DT)

    def groups_of(self, r):
        """(first_site positions, n_motif, unmethylated scores, methylated scores) of read r's groups (jobs_on_device=True)."""
        self.sync()
        ng = int(self.d_n_groups[r]); g0 = int(self.group_off[r])
        sc = self.d_scores[2 * g0:2 * (g0 + ng)].cpu().numpy()
        return self.d_first[g0:g0 + ng].cpu().numpy(), self.d_n_motif[g0:g0 + ng].cpu().numpy(), sc[0::2], sc[1::2]

    def calibrated(self):
        self.sync()
        return self.d_calibrated.cpu().numpy()

    def genome_site_table(self, out=None, overflow=None, call_threshold=2.0, per_site=False):
        """The batch's per-site table keyed (contig, start, end) on the resident contigs (np_site_table_genome_dev; sites.site_table_genome_dev):
        needs a record batch with work items built on the device.  per_site: one row per motif SITE of the genome (the rank structure of
        np_genome_site_index_dev, built on first use) instead of one per base.  Returns (table int32 [genome length or sites, 6], overflow [1])."""
        assert self.by_cigar and self.jobs_on_device, "genome-keyed table: a record batch (build_host_batch_records) with jobs_on_device=True"
        from .sites import site_table_genome_dev, genome_site_index_dev
        if not hasattr(self, "d_contig_off"):
            self.d_contig_off = self.torch.from_numpy(np.ascontiguousarray(self.hb["contig_off"], np.int64)).to(self.d_scores.device)
        if per_site and getattr(self, "site_index", None) is None:
            self.site_index = genome_site_index_dev(self.ctx, self.torch, self.d_genome, self.d_contig_off, alphabet=self.alphabet, stream=self.stream)
        return site_table_genome_dev(self.ctx, self.torch, self.d_scores, self.d_first, self.d_last, self.d_n_motif, self.d_jobs,
                                     self.d_ref_begin.view(self.torch.int64), self.d_genome, self.d_contig_off, alphabet=self.alphabet,
                                     min_separation=self.cm[0], call_threshold=call_threshold, stream=self.stream, out=out, overflow=overflow,
                                     index=self.site_index if per_site else None)

    def event_map(self):
        self.sync()
        return self.d_map.cpu().numpy(), (self.d_map_stop.cpu().numpy() if self.want_map_stop else None)

    def groups_bulk(self, n):
        """groups_of(r) for reads 0..n-1 with one transfer per array: list of (first, n_motif, unmeth, meth)."""
        self.sync()
        g1 = int(self.group_off[n])
        ng = self.d_n_groups[:n].cpu().numpy()
        first = self.d_first[:g1].cpu().numpy(); nm = self.d_n_motif[:g1].cpu().numpy()
        sc = self.d_scores[:2 * g1].cpu().numpy()
        out = []
        for r in range(n):
            g0 = int(self.group_off[r]); k = max(int(ng[r]), 0)
            out.append((first[g0:g0 + k], nm[g0:g0 + k], sc[2 * g0:2 * (g0 + k):2], sc[2 * g0 + 1:2 * (g0 + k):2]))
        return out


class StreamedFeed:
    """Host-fed operation of a CallMethylationBatch: every step's inputs arrive in pinned host memory, as BamProcessor's record
    batches would deliver them (src/common/nanopolish_bam_processor.cpp:90-119), and are copied to the device while the
    previous step computes; the results go back to pinned host memory the same way.  Two sets of input and output buffers
    (double buffering), three HIP streams: host->device, compute (every *_dev entry point of the C ABI takes the stream),
    device->host.  Scratch (alignments, event maps, work items) stays single: the compute stream runs one step at a time.

    host -> device per step: event means (or raw samples), nucleotide k-mer ranks, the per-read records, and -- when work items
    are generated on the device -- the reads' reference strands;  device -> host: the scores and the per-group site metadata.
    """

    def __init__(self, batch):
        torch = batch.torch
        self.b = b = batch
        ins = ["d_ranks", "d_reads_a", "d_reads_b"] + ((["d_adc"] if b.from_adc else ["d_raw"]) if b.from_raw else ["d_events"])
        if b.jobs_on_device and not b.by_cigar:
            ins += ["d_seq", "d_rc"]
        outs = ["d_scores"] + (["d_first", "d_n_motif", "d_n_groups"] if b.jobs_on_device else [])
        b.sync(); torch.cuda.synchronize()
        self.ins, self.outs = ins, outs
        self.host_in = {n: getattr(b, n).cpu().pin_memory() for n in ins}
        self.host_out = [{n: torch.empty_like(getattr(b, n), device="cpu").pin_memory() for n in outs} for _ in range(2)]
        self.ref_out = {n: getattr(b, n).clone() for n in outs}            # the resident pass's results, for check_against
        self.sets = [{n: getattr(b, n) for n in ins + outs}, {n: torch.zeros_like(getattr(b, n)) for n in ins + outs}]
        self.h2d_bytes = sum(t.numel() * t.element_size() for t in self.host_in.values())
        self.d2h_bytes = sum(t.numel() * t.element_size() for t in self.host_out[0].values())
        self.s_h2d, self.s_cmp, self.s_d2h = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        self.ev_h2d = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_cmp = [None, None]
        self.ev_d2h = [None, None]
        self.k = 0
        self.read_back = 0           # steps whose read-back has been enqueued
        self.done_events = []        # one timed event per step, recorded when the step's kernels have finished
        self.host_ms = []            # host time of each submit: (enqueue H2D, enqueue compute, enqueue D2H) in ms
        torch.cuda.synchronize()

    def submit(self):
        import time
        torch = self.b.torch
        i = self.k & 1
        st = self.sets[i]
        t0 = time.perf_counter()
        with torch.cuda.stream(self.s_h2d):
            if self.ev_cmp[i] is not None:
                self.s_h2d.wait_event(self.ev_cmp[i])          # the step that last read this input set has finished
            for n in self.ins:
                st[n].copy_(self.host_in[n], non_blocking=True)
            self.ev_h2d[i].record(self.s_h2d)
        t1 = time.perf_counter()
        self.s_cmp.wait_event(self.ev_h2d[i])
        if self.ev_d2h[i] is not None:
            self.s_cmp.wait_event(self.ev_d2h[i])              # this output set has been read back
        for n in self.ins + self.outs:
            setattr(self.b, n, st[n])
        self.b.stream = self.s_cmp.cuda_stream
        self.b.step()
        t2 = time.perf_counter()
        self.ev_cmp[i] = torch.cuda.Event(enable_timing=True); self.ev_cmp[i].record(self.s_cmp)
        self.done_events.append(self.ev_cmp[i])
        # The read-back of step k is enqueued one submit LATER, behind the host->device copies of step k+1: HIP multiplexes
        # streams onto a few in-order hardware queues, and when the two copy streams share one, a device->host copy that waits
        # for the compute of step k would hold back the upload of step k+1 queued behind it -- the upload that is supposed to
        # run WHILE step k computes (seen as a 20 ms bubble per step in profiles/r02: the upload started only after the
        # previous step's kernels).
        if self.read_back < self.k:
            self._read_back(i ^ 1)                              # step k - 1
            self.read_back = self.k
        self.k += 1
        self.host_ms.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((time.perf_counter() - t2) * 1e3, 2)))

    def _read_back(self, i):
        torch = self.b.torch
        with torch.cuda.stream(self.s_d2h):
            self.s_d2h.wait_event(self.ev_cmp[i])
            for n in self.outs:
                self.host_out[i][n].copy_(self.sets[i][n], non_blocking=True)
            self.ev_d2h[i] = torch.cuda.Event(); self.ev_d2h[i].record(self.s_d2h)

    def drain(self):
        if self.read_back < self.k:
            self._read_back((self.k - 1) & 1)              # the last step's results
            self.read_back = self.k
        for s in (self.s_h2d, self.s_cmp, self.s_d2h):
            s.synchronize()
        self.b.sync()

    def steady_ms_per_step(self, last):
        """mean device time between the completions of consecutive steps over the last `last` steps (the first step of a
        sequence also pays its own upload, which nothing overlaps)"""
        self.drain()
        ev = self.done_events[-last:]
        return ev[0].elapsed_time(ev[-1]) / (len(ev) - 1) if len(ev) >= 2 else None

    def check_against(self, batch=None):
        """the results of the last two streamed steps, as they arrived on the host, against the resident pass (bit for bit)"""
        self.drain()
        ok = True
        for i in range(min(self.k, 2)):
            for n in self.outs:
                a = self.host_out[i][n].numpy().view(np.uint8); r = self.ref_out[n].cpu().numpy().view(np.uint8)
                ok = ok and bool(np.array_equal(a, r))
        return ok

    def close(self):
        self.drain()
        for n in self.ins + self.outs:
            setattr(self.b, n, self.sets[0][n])
        self.b.stream = None
