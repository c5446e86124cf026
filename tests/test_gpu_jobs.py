"""f3: call-methylation work items generated on the device vs the host builder
(np_cm_build_jobs_identity, itself checked against the oracle's motif/window/k-mer rules in tests/test_host_logic.py)."""
import ctypes as C
import numpy as np
import pytest

from cases import synth_read
from nanopolish_amd import api, lib as _l
from nanopolish_amd.pipeline import JOB_DT

pytestmark = pytest.mark.gpu


def _device_jobs(ctx, refs, rcs, alphabet, k=6, min_separation=10, min_flank=10):
    import torch
    dev = torch.device("cuda:0")
    n = len(refs)
    seq_off = np.zeros(n + 1, np.int64); seq_off[1:] = np.cumsum([len(r) for r in refs])
    seq = np.frombuffer("".join(refs).encode(), np.uint8)
    gcap = np.array([len(r) // 2 + 1 for r in refs], np.int64)
    group_off = np.zeros(n + 1, np.int64); group_off[1:] = np.cumsum(gcap)
    rcap = np.array([8 * len(r) + 64 for r in refs], np.int64)
    rank_off = np.zeros(n + 1, np.int64); rank_off[1:] = np.cumsum(rcap)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    d_seq, d_seq_off, d_rc = up(seq), up(seq_off), up(np.array(rcs, np.uint8))
    d_goff, d_roff = up(group_off), up(rank_off)
    ns = int(group_off[-1])
    d_jobs = torch.zeros(2 * ns * JOB_DT.itemsize, dtype=torch.uint8, device=dev)
    d_kpos = torch.zeros(4 * ns, dtype=torch.int32, device=dev)
    d_ranks = torch.zeros(int(rank_off[-1]), dtype=torch.int16, device=dev)
    d_first = torch.zeros(ns, dtype=torch.int32, device=dev); d_last = torch.zeros(ns, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(ns, dtype=torch.int32, device=dev); d_ng = torch.zeros(n, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = ctx.L.np_cm_build_jobs_identity_dev(ctx.h, None, n, p(d_seq), p(d_seq_off), p(d_rc), api.alphabet_id(alphabet), k, min_separation,
                                             min_flank, p(d_goff), ns, p(d_roff), p(d_jobs), p(d_kpos), p(d_ranks), p(d_first), p(d_last),
                                             p(d_nm), p(d_ng))
    ctx._chk(rc, "np_cm_build_jobs_identity_dev")
    ctx.sync()
    return dict(group_off=group_off, jobs=d_jobs.cpu().numpy().view(JOB_DT), kpos=d_kpos.cpu().numpy().reshape(-1, 2),
                ranks=d_ranks.cpu().numpy().view(np.uint16), first=d_first.cpu().numpy(), last=d_last.cpu().numpy(), n_motif=d_nm.cpu().numpy(),
                n_groups=d_ng.cpu().numpy())


@pytest.mark.parametrize("alphabet", ["cpg", "gpc", "dam", "dcm"])
def test_device_work_items_equal_host_builder(ctx, models, alphabet):
    rng = np.random.default_rng(4)
    reads = [synth_read(500 + i, models["nucleotide"], L=L) for i, L in enumerate((1500, 900, 5450, 64, 23, 300, 2500, 700))]
    refs, rcs = [], []
    for rd in reads:
        refs.append(api.reverse_complement("nucleotide", rd["seq"]) if rd["rc"] else rd["seq"]); rcs.append(rd["rc"])
    # motif-dense, motif-free and motif-at-the-edges references
    refs += ["CG" * 300, "ACGT" * 200, "A" * 400, "CG" + "A" * 200 + "GC" + "T" * 200 + "CG", "GCGCGCGC" + "ACGTTGCA" * 60 + "GCGC"]
    rcs += [False, True, False, True, False]
    # 4- and 5-base sites (dam GATC, dcm CCAGG / CCTGG): dense, adjacent, cut by either end of the reference, near-misses
    refs += ["GATC" * 120, "CCAGGCCTGG" * 60, "ATC" + "A" * 30 + "GATCGATC" + "T" * 40 + "CCAGGACCTGG" + "C" * 50 + "GAT",
             "CCAGGT" * 3 + "ACGTTGCAAC" * 40 + "CCTGGATCCAGG" + "TTGACA" * 20 + "CCTG", "GATCCAGGATCCTGGATC" * 25,
             "GTTC" * 50 + "CCGGG" * 30 + "GATC"]
    rcs += [False, True, True, False, True, False]
    dv = _device_jobs(ctx, refs, rcs, alphabet)
    n_items = 0
    for i, (ref, rc) in enumerate(zip(refs, rcs)):
        want = api.cm_build_jobs_identity(ref, rc, 6, alphabet)
        ng = len(want["first"])
        assert dv["n_groups"][i] == ng, (i, alphabet)
        g0 = int(dv["group_off"][i])
        assert np.array_equal(dv["first"][g0:g0 + ng], want["first"]) and np.array_equal(dv["last"][g0:g0 + ng], want["last"])
        assert np.array_equal(dv["n_motif"][g0:g0 + ng], want["n_motif"])
        for g in range(ng):
            lo, hi = int(want["rank_off"][g]), int(want["rank_off"][g + 1])
            for v, key in ((0, "ranks_unmeth"), (1, "ranks_meth")):
                jb = dv["jobs"][2 * (g0 + g) + v]
                assert jb["n_kmers"] == hi - lo and jb["read"] == i and jb["flags"] == 3 and jb["stride"] == 1
                assert np.array_equal(dv["ranks"][jb["rank_off"]:jb["rank_off"] + jb["n_kmers"]], want[key][lo:hi]), (i, g, key)
                assert np.array_equal(dv["kpos"][2 * (g0 + g) + v], want["kpos"][g])
                n_items += 1
        # unused slots are marked skipped
        cap = int(dv["group_off"][i + 1]) - g0
        assert np.all(dv["jobs"][2 * (g0 + ng):2 * (g0 + cap)]["flags"] == 0x80000000)
    assert n_items > (500 if alphabet in ("cpg", "gpc") else 60)       # (4- and 5-base sites are rare in random sequence)


def test_whole_chain_on_device_matches_oracle(ctx, orc, models):
    """Only raw samples, k-mer ranks and the reference strands go to the device: event detection, MoM scalings, event alignment,
    recalibration, work-item generation and 2 x score per group all run there (f1 + f2 + f3) -- vs the same chain on the oracle."""
    from cases import call_methylation_read
    from nanopolish_amd.pipeline import build_host_batch, tile_host_batch, CallMethylationBatch
    from oracle.oracle_py import ED_DEFAULTS
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    hb = build_host_batch(models, list(range(110, 116)), L=2200, raw=True)
    batch = CallMethylationBatch(ctx, tile_host_batch(hb, 2), "cuda:0", calibrate=True, from_raw=True, jobs_on_device=True)
    batch.step(); batch.step()
    n_scored = 0
    for i, rd in enumerate(hb["reads"]):
        ev = orc.detect_events(rd["raw"], **ED_DEFAULTS)
        want = call_methylation_read(orc, mn, mc, dict(rd, events=ev["mean"]), calibrate=True)
        for copy in (0, 1):
            first, nm, u, m = batch.groups_of(i + copy * hb["n"])
            got = {int(f): (float(a), float(b)) for f, a, b in zip(first, u, m) if a == a}
            assert got == {int(f): (float(a), float(b)) for f, a, b in zip(want["first"], want["unmeth"], want["meth"])}
            n_scored += len(got)
    assert n_scored > 400


def test_slot_layout_changes_nothing(ctx, orc, models):
    """np_set_job_layout (round 5): the kernels between the work-item builder and the scorer visit the live items of every read's slot range
    only.  The whole pass -- identity and CIGAR work items, ragged reads, reads without a single site, from events and from raw signal --
    with the layout declared (the pipeline's default) and without: the same scores in every slot (NaN in the unused ones either way), the
    same groups, the same per-site table; and a layout that describes another array is refused."""
    import ctypes as C
    import torch
    from nanopolish_amd.pipeline import build_host_batch, tile_host_batch, CallMethylationBatch
    from nanopolish_amd.sites import site_table_dev
    for raw in (False, True):
        hb = build_host_batch(models, list(range(700, 720)), L=[64, 23, 300] + [700 + 211 * i for i in range(17)], raw=raw, with_jobs=False)
        res = []
        for use in (True, False, True):
            b = CallMethylationBatch(ctx, tile_host_batch(hb, 2), "cuda:0", calibrate=True, from_raw=raw, jobs_on_device=True)
            b.use_job_layout = use
            b.step(); b.step()
            sc = b.scores()
            b.max_len = int(max(len(q) for q in hb["ref_seqs"]))
            tbl = site_table_dev(ctx, torch, b.d_scores, b.d_first, b.d_n_motif, b.max_len); ctx.sync()
            groups = [tuple(np.asarray(x).tobytes() for x in b.groups_of(i)) for i in range(b.n_reads)]
            res.append((sc.tobytes(), tbl.cpu().numpy().tobytes(), groups, int(np.isfinite(sc).sum())))
            if use:      # a declared layout must describe the array the call works on
                p = lambda t: C.c_void_p(t.data_ptr())
                assert ctx.L.np_set_job_layout(ctx.h, b.n_reads, p(b.d_group_off), p(b.d_n_groups), b.n_slots + 1) == 0
                rc = ctx.L.np_hmm_score_dev(ctx.h, None, b.n_jobs, p(b.d_jobs), p(b.d_reads_b), p(b.d_events), p(b.d_job_ranks), b.m_cpg, p(b.d_scores))
                assert rc != 0
                assert ctx.L.np_set_job_layout(ctx.h, 0, None, None, 0) == 0
            del b
        assert res[0][3] > 1000
        assert res[0] == res[1] == res[2]


def test_host_scoring_ignores_a_layout_declared_for_dev_calls(ctx, orc, models):
    """np_set_job_layout is state of the context's *_dev calls.  A per-record np_hmm_score_host that arrives while a batched
    pass has its layout declared packs its OWN dense work-item array: it must neither be refused ('describes another work-item array') nor be
    scored through the foreign slots -- also on the general path (device-side binning), where run_hmm_forward used to pick the layout up."""
    import ctypes as C
    import torch
    from test_gpu_parity import _score_jobs, _reads
    jobs, want = _score_jobs(ctx, orc, models, _reads(models, range(40, 43), 900))
    want = np.array(want, np.float32)
    group_off = torch.tensor([0, 7], dtype=torch.int64, device="cuda:0")
    n_groups = torch.tensor([3], dtype=torch.int32, device="cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())
    try:
        for total in (11, len(jobs) // 2):        # a foreign count, and 2 x total == n_jobs (every job list here has two items per group)
            assert ctx.L.np_set_job_layout(ctx.h, 1, p(group_off), p(n_groups), total) == 0
            for small in (1, 0):
                ctx.set_option("small_batch_path", small)
                got = ctx.profile_hmm_score(jobs)
                assert np.array_equal(got, want)
    finally:
        ctx.set_option("small_batch_path", 1)
        assert ctx.L.np_set_job_layout(ctx.h, 0, None, None, 0) == 0
