"""GPU: the device pass from raw signal and BAM records against vectors the REFERENCE ITSELF produced
(tests/golden/golden_reflevel.npz = calculate_methylation_for_read compiled in place, see tests/gen_golden_reflevel.py):
event detection -> MoM -> event alignment -> event map -> recalibration -> CIGAR-driven work items -> 2 x profile_hmm_score,
work items built on the host (np_cm_build_jobs_cigar) and on the device (np_cm_build_jobs_cigar_dev)."""
import os
import numpy as np
import pytest

from nanopolish_amd import api
from nanopolish_amd.pipeline import build_host_batch_records, CallMethylationBatch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_reflevel.npz")


def _s(a):
    return bytes(a).decode()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _records(gold):
    recs, want = [], []
    for i in range(int(gold["n_reads"])):
        p = "r%d_" % i
        rc, pos = (int(v) for v in gold[p + "rc_pos"])
        recs.append(dict(seq=_s(gold[p + "seq"]), raw=gold[p + "raw"], rc=rc, pos=pos, cigar=gold[p + "cigar"]))
        want.append({int(s): (float(u), float(m)) for s, u, m in zip(gold[p + "site_start"], gold[p + "site_ll_unmeth"], gold[p + "site_ll_meth"])})
    return recs, want


def test_device_pass_with_host_work_items_matches_reference(ctx, models, gold):
    recs, want = _records(gold)
    hb = build_host_batch_records(models, recs, _s(gold["contig"]))
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True)
    batch.step(); batch.step()
    sc = batch.scores()
    jobs = batch.jobs_host()
    n = 0
    for i, rec in enumerate(recs):
        p = "r%d_" % i
        # read-level state the reference left: detected events, calibrated scalings, events per base
        ne, _, _, mean, _ = batch.detected(i)
        if int(gold[p + "n_events"]):
            assert np.array_equal(mean, gold[p + "events"])
            r = batch.reads_scored()[i]
            assert (r["shift"], r["scale"], r["var"]) == tuple(gold[p + "scalings"][:3]) and batch.epb()[i] == gold[p + "scalings"][3]
        lo, hi = int(hb["job_off"][i]), int(hb["job_off"][i + 1])
        got = {}
        for g, f in enumerate(hb["meta"][i]["first"]):
            j = lo + 2 * g
            if not (jobs[j]["flags"] & 0x80000000):
                got[int(f) + rec["pos"]] = (float(sc[j]), float(sc[j + 1]))
        assert got == want[i], i
        n += len(got)
    assert n > 150


def test_whole_chain_on_device_from_bam_records_matches_reference(ctx, models, gold):
    """only raw samples, read k-mer ranks, the contig and the CIGARs go to the device"""
    recs, want = _records(gold)
    hb = build_host_batch_records(models, recs, _s(gold["contig"]))
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=True)
    batch.step(); batch.step()
    n = 0
    for i, rec in enumerate(recs):
        first, nm, u, m = batch.groups_of(i)
        got = {int(f) + rec["pos"]: (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
        assert got == want[i], i
        # the device builder writes the same items as the host builder (groups that survive the CIGAR bounds)
        assert np.array_equal(first, hb["meta"][i]["first"]) and np.array_equal(nm, hb["meta"][i]["n_motif"])
        n += len(got)
    assert n > 150


def test_device_cigar_items_equal_host_builder_on_adversarial_cigars(ctx, models):
    """ragged CIGARs (clips, long indels, zero-length ops, alignments shorter than k, no aligned base at all) and a spliced
    record: kpos, ranks, sites and the degenerate-record positions, device vs host, item by item"""
    import ctypes as C
    import torch
    from nanopolish_amd.pipeline import JOB_DT
    from nanopolish_amd.synth import BASES
    rng = np.random.default_rng(9)
    g = rng.integers(0, 4, 3000)
    contig = BASES[g].tobytes().decode()
    cases = []
    for t in range(24):
        ops = []
        if rng.random() < 0.5:
            ops.append(("S", int(rng.integers(1, 40))))
        for _ in range(int(rng.integers(1, 60))):
            ops.append((str(rng.choice(list("MMMM=XIDID"))), int(rng.integers(0 if rng.random() < 0.1 else 1, 60))))
        if rng.random() < 0.5:
            ops.append(("S", int(rng.integers(1, 40))))
        cases.append(ops)
    cases += [[("M", 5)], [("S", 30)], [("M", 2000)], [("M", 300), ("N", 50), ("M", 300)], [("H", 5), ("M", 11), ("D", 400), ("M", 11), ("H", 2)],
              [("I", 20), ("M", 700), ("I", 20)]]
    recs = []
    for t, ops in enumerate(cases):
        span = sum(n for o, n in ops if o in "MD=XN")
        qlen = sum(n for o, n in ops if o in "MIS=X")
        pos = int(rng.integers(0, max(1, len(contig) - span)))
        recs.append(dict(pos=pos, rc=bool(t & 1), cigar=api.cigar_words(ops), read_len=qlen, span=span, spliced=any(o == "N" for o, _ in ops)))
    dev = torch.device("cuda:0")
    n = len(recs)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    ref_begin = np.array([r["pos"] for r in recs], np.int64)
    ref_len = np.array([min(r["pos"] + max(r["span"], 1) + 1, len(contig)) - r["pos"] for r in recs], np.int32)
    cigar_off = np.zeros(n + 1, np.int64); cigar_off[1:] = np.cumsum([len(r["cigar"]) for r in recs])
    gcap = ref_len.astype(np.int64) // 2 + 2
    group_off = np.zeros(n + 1, np.int64); group_off[1:] = np.cumsum(gcap)
    rank_off = np.zeros(n + 1, np.int64); rank_off[1:] = np.cumsum(8 * ref_len.astype(np.int64) + 64)
    ns = int(group_off[-1])
    d = dict(genome=up(np.frombuffer(contig.encode(), np.uint8)), ref_begin=up(ref_begin), ref_len=up(ref_len),
             cigar=up(np.concatenate([r["cigar"] for r in recs])), cigar_off=up(cigar_off),
             read_len=up(np.array([r["read_len"] for r in recs], np.int32)), rc=up(np.array([r["rc"] for r in recs], np.uint8)),
             goff=up(group_off), roff=up(rank_off))
    d_jobs = torch.zeros(2 * ns * JOB_DT.itemsize, dtype=torch.uint8, device=dev)
    d_kpos = torch.zeros(4 * ns, dtype=torch.int32, device=dev)
    d_ranks = torch.zeros(int(rank_off[-1]), dtype=torch.int16, device=dev)
    d_first, d_last, d_nm = (torch.zeros(ns, dtype=torch.int32, device=dev) for _ in range(3))
    d_ng = torch.zeros(n, dtype=torch.int32, device=dev); d_deg = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = ctx.L.np_cm_build_jobs_cigar_dev(ctx.h, None, n, p(d["genome"]), p(d["ref_begin"]), p(d["ref_len"]), p(d["cigar"]), p(d["cigar_off"]),
                                          int(cigar_off[-1]), p(d["read_len"]), p(d["rc"]), api.alphabet_id("cpg"), 6, 10, 10, p(d["goff"]), ns,
                                          p(d["roff"]), p(d_jobs), p(d_kpos), p(d_ranks), p(d_first), p(d_last), p(d_nm), p(d_ng), p(d_deg))
    ctx._chk(rc, "np_cm_build_jobs_cigar_dev")
    ctx.sync()
    jobs = d_jobs.cpu().numpy().view(JOB_DT); kpos = d_kpos.cpu().numpy().reshape(-1, 2); ranks = d_ranks.cpu().numpy().view(np.uint16)
    first, last, nm, ng_dev, deg = d_first.cpu().numpy(), d_last.cpu().numpy(), d_nm.cpu().numpy(), d_ng.cpu().numpy(), d_deg.cpu().numpy().reshape(-1, 2)
    n_items = 0
    for i, r in enumerate(recs):
        g0 = int(group_off[i])
        if r["spliced"]:
            assert ng_dev[i] == 0 and tuple(deg[i]) == (-1, -1)
            continue
        seg = contig[r["pos"]:r["pos"] + int(ref_len[i])]
        want = api.cm_build_jobs_cigar(seg, r["cigar"], r["read_len"], r["rc"])
        ng = len(want["first"])
        assert ng_dev[i] == ng, i
        assert tuple(deg[i]) == tuple(want["deg_kpos"]), i
        assert np.array_equal(first[g0:g0 + ng], want["first"]) and np.array_equal(last[g0:g0 + ng], want["last"])
        assert np.array_equal(nm[g0:g0 + ng], want["n_motif"])
        for gi in range(ng):
            lo, hi = int(want["rank_off"][gi]), int(want["rank_off"][gi + 1])
            for v, key in ((0, "ranks_unmeth"), (1, "ranks_meth")):
                jb = jobs[2 * (g0 + gi) + v]
                assert jb["n_kmers"] == hi - lo and jb["read"] == i
                assert np.array_equal(ranks[jb["rank_off"]:jb["rank_off"] + jb["n_kmers"]], want[key][lo:hi])
                assert np.array_equal(kpos[2 * (g0 + gi) + v], want["kpos"][gi]), (i, gi)
                n_items += 1
    assert n_items > 200


def test_eventalign_chain_on_device_matches_align_read_to_ref(ctx, models, gold):
    """BASELINE config 3: the whole eventalign realignment -- from raw signal: detection, scalings, event alignment, calibration,
    then per read the chain of ~100-base Viterbi segments (np_eventalign_dev) -- against the rows the reference's own
    align_read_to_ref emitted (both strands), and against the CIGAR reads' chain on the oracle."""
    recs, want = [], []
    for j in range(2):
        p = "ea%d_" % j
        seq = _s(gold[p + "seq"]); rc = int(gold[p + "rc"])
        ref = api.reverse_complement("nucleotide", seq) if rc else seq
        recs.append(dict(seq=seq, raw=gold[p + "raw"], rc=rc, pos=0, cigar=api.cigar_words([("M", len(seq))]), contig=ref))
        want.append((gold[p + "ref_position"], gold[p + "event_idx"], gold[p + "hmm_state"]))
    hb = build_host_batch_records(models, recs, "")
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True)
    batch.step()
    got = batch.eventalign()
    for j, (g, w) in enumerate(zip(got, want)):
        assert g["status"] == 0 and g["n_calls"] > 20
        assert np.array_equal(g["ref_position"], w[0]) and np.array_equal(g["event_idx"], w[1]) and np.array_equal(g["hmm_state"], w[2]), j
    # records with insertions, deletions and soft clips (and one read without events)
    recs, _ = _records(gold)
    hb = build_host_batch_records(models, recs, _s(gold["contig"]))
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True)
    batch.step()
    got = batch.eventalign()
    rows = 0
    for i, g in enumerate(got):
        p = "r%d_" % i
        assert g["status"] == 0
        assert np.array_equal(g["ref_position"], gold[p + "ea_ref_position"]) and np.array_equal(g["event_idx"], gold[p + "ea_event_idx"]), i
        assert np.array_equal(g["hmm_state"], gold[p + "ea_hmm_state"])
        rows += len(g["event_idx"])
        # ... and printed from what the device holds (detected events, calibrated scalings): the reference's TSV, byte for byte
        import zlib
        from nanopolish_amd.eventalign import format_eventalign_tsv
        want_tsv = zlib.decompress(bytes(gold[p + "ea_tsv_z"])).decode()
        if len(g["event_idx"]):
            ne, _, length, mean, stdv = batch.detected(i)
            r = batch.reads_scored()[i]
            rec = recs[i]
            contig = _s(gold["contig"])
            span = int(sum(int(w) >> 4 for w in rec["cigar"] if (int(w) & 0xf) in (0, 2, 7, 8)))
            seg = contig[rec["pos"]:min(rec["pos"] + span + 1, len(contig))]
            got_tsv = format_eventalign_tsv(g, "contig", seg, rec["pos"], i, rec["rc"], mean, stdv, length, 4000.0, models["nucleotide"],
                                            r["shift"], r["scale"], r["var"])
            assert "".join(got_tsv) == want_tsv
        else:
            assert want_tsv == ""
    assert rows > 8000


def test_eventalign_chain_edge_cases(ctx, models, gold):
    """scratch too small for a segment (status NP_EA_OVERFLOW, what was emitted before is a prefix of the full result),
    a spliced record and a record whose CIGAR has no aligned base (no rows), an empty batch"""
    import ctypes as C
    recs, _ = _records(gold)
    good = recs[0]
    spliced = dict(good, cigar=api.cigar_words([("M", 300), ("N", 40), ("M", 300)]))
    clipped = dict(good, cigar=api.cigar_words([("S", len(good["seq"]))]))
    hb = build_host_batch_records(models, [good, spliced, clipped], _s(gold["contig"]))
    batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, workload="eventalign")
    batch.step()
    full = batch.eventalign_results()
    assert full[0]["status"] == 0 and np.array_equal(full[0]["event_idx"], gold["r0_ea_event_idx"])
    assert len(full[1]["event_idx"]) == 0 and len(full[2]["event_idx"]) == 0
    ctx._chk(ctx.L.np_set_option(ctx.h, b"ea_rows_cap", 100), "np_set_option")          # a ~165-row segment no longer fits
    try:
        batch.step()
        part = batch.eventalign_results()
    finally:
        ctx._chk(ctx.L.np_set_option(ctx.h, b"ea_rows_cap", 4096), "np_set_option")
    assert part[0]["status"] == 1
    m = len(part[0]["event_idx"])
    assert m < len(full[0]["event_idx"]) and np.array_equal(part[0]["event_idx"], full[0]["event_idx"][:m])
    # empty batch
    rc = ctx.L.np_eventalign_dev(ctx.h, None, 0, None, None, None, None, None, None, ctx.models["nucleotide"], None, None, None, None, None, 0,
                                 None, None, 6, None, None, None, None, None, None, None)
    assert rc == 0


def test_gpc_chain_on_device_matches_reference(ctx, models, gold):
    """the whole chain under --methylation gpc (device work items for GC sites, gpc model) vs the reference's own output"""
    z = np.load(os.path.join(os.path.dirname(GOLD), "models_r9.4_450bps_gpc.npz"))
    if "gpc" not in ctx.models:
        ctx.register_model(dict(k=6, level_mean=z["level_mean"], level_stdv=z["level_stdv"], level_log_stdv=z["level_log_stdv"]), "gpc")
    recs, _ = _records(gold)
    recs = recs[:2]
    hb = build_host_batch_records(models, recs, _s(gold["contig"]), alphabet="gpc")
    for on_dev in (False, True):
        batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=on_dev)
        batch.step()
        for i, rec in enumerate(recs):
            want = {int(s): (float(u), float(m)) for s, u, m in zip(gold["r%d_gpc_start" % i], gold["r%d_gpc_ll_unmeth" % i], gold["r%d_gpc_ll_meth" % i])}
            if on_dev:
                first, nm, u, m = batch.groups_of(i)
                got = {int(f) + rec["pos"]: (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
            else:
                sc, jobs = batch.scores(), batch.jobs_host()
                lo = int(hb["job_off"][i])
                got = {int(f) + rec["pos"]: (float(sc[lo + 2 * g]), float(sc[lo + 2 * g + 1])) for g, f in enumerate(hb["meta"][i]["first"])
                       if not (jobs[lo + 2 * g]["flags"] & 0x80000000)}
            assert got == want and len(got) > 30, (on_dev, i)


@pytest.mark.parametrize("alpha", ["dam", "dcm"])
def test_dam_and_dcm_chains_match_reference(ctx, models, gold, alpha):
    """--methylation dam / dcm (4- and 5-base sites, two sites for dcm): work items from the host builder and from the device
    builder, everything else on the device, against the reference's own output on a motif-rich contig"""
    z = np.load(os.path.join(os.path.dirname(GOLD), "models_r9.4_450bps_%s.npz" % alpha))
    if alpha not in ctx.models:
        ctx.register_model(dict(k=6, level_mean=z["level_mean"], level_stdv=z["level_stdv"], level_log_stdv=z["level_log_stdv"]), alpha)
    recs = []
    for j in range(2):
        p = "m%d_" % j
        rc, pos = (int(v) for v in gold[p + "rc_pos"])
        recs.append(dict(seq=_s(gold[p + "seq"]), raw=gold[p + "raw"], rc=rc, pos=pos, cigar=gold[p + "cigar"]))
    hb = build_host_batch_records(models, recs, _s(gold["m_contig"]), alphabet=alpha)
    for on_dev in (False, True):
        batch = CallMethylationBatch(ctx, hb, "cuda:0", calibrate=True, from_raw=True, jobs_on_device=on_dev)
        batch.step()
        sc, jobs = batch.scores(), batch.jobs_host()
        for i, rec in enumerate(recs):
            p = "m%d_" % i
            want = {int(s): (float(u), float(m)) for s, u, m in zip(gold[p + alpha + "_start"], gold[p + alpha + "_ll_unmeth"], gold[p + alpha + "_ll_meth"])}
            if on_dev:
                first, nm, u, m = batch.groups_of(i)
                got = {int(f) + rec["pos"]: (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
                assert np.array_equal(nm, hb["meta"][i]["n_motif"])
            else:
                lo = int(hb["job_off"][i])
                got = {int(f) + rec["pos"]: (float(sc[lo + 2 * g]), float(sc[lo + 2 * g + 1])) for g, f in enumerate(hb["meta"][i]["first"])
                       if not (jobs[lo + 2 * g]["flags"] & 0x80000000)}
            assert got == want and len(got) > 15, (alpha, i, on_dev)