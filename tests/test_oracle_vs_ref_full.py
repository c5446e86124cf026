"""CPU, build container only (needs oracle/_ref/libnp_ref_full.so = the reference's read-level code compiled in place):
pins the portable oracle's read-level helpers -- CIGAR walk, EventAlignmentRecord, _find_by_ref_bounds, the whole
calculate_methylation_for_read pass, modBAM tags -- and the product's host mirrors against the reference on fresh seeds,
incl. adversarial CIGARs.  Skipped where /root/reference does not exist (the committed goldens cover that case)."""
import numpy as np
import pytest

from oracle.ref_full import have_full, cigar_words

pytestmark = pytest.mark.skipif(not have_full(), reason="oracle/_ref/libnp_ref_full.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def full():
    from oracle.ref_full import FullRef
    return FullRef()


def _random_cigar(rng, n_ops):
    ops = []
    if rng.random() < 0.3:
        ops.append(("H", int(rng.integers(1, 9))))
    if rng.random() < 0.6:
        ops.append(("S", int(rng.integers(1, 30))))
    for _ in range(n_ops):
        ops.append((str(rng.choice(list("MMMM=XIDID"))), int(rng.integers(0 if rng.random() < 0.1 else 1, 25))))
    if rng.random() < 0.6:
        ops.append(("S", int(rng.integers(1, 30))))
    return ops


def test_cigar_walk_matches_get_aligned_segments(full, orc):
    from nanopolish_amd import api
    rng = np.random.default_rng(1)
    for _ in range(200):
        ops = _random_cigar(rng, int(rng.integers(1, 40)))
        cig = cigar_words(ops)
        qlen = sum(n for o, n in ops if o in "MIS=X")
        pos = int(rng.integers(0, 5000))
        want = full.aligned_bases(0, pos, cig, "A" * max(qlen, 1))
        assert np.array_equal(orc.cigar_aligned_bases(cig, pos), want)
        assert np.array_equal(api.cigar_aligned_bases(cig, pos), want)


def test_spliced_and_padded_cigars_are_rejected(orc):
    """N starts a second segment (SequenceAlignmentRecord exits), P hits get_aligned_segments' assert: both are errors in
    the restatement and in the product's host mirror"""
    from nanopolish_amd import api
    for bad in ("N", "P"):
        cig = cigar_words([("M", 10), (bad, 3), ("M", 10)])
        assert orc.cigar_aligned_bases(cig, 0) is None
        with pytest.raises(ValueError):
            api.cigar_aligned_bases(cig, 0)


def test_find_by_ref_bounds_matches_reference(full, orc):
    rng = np.random.default_rng(2)
    for _ in range(300):
        n = int(rng.integers(1, 60))
        ref = np.cumsum(rng.integers(1, 4, n)) + int(rng.integers(0, 50))
        rd = np.cumsum(rng.integers(0, 3, n))
        pairs = np.stack([ref, rd], 1).astype(np.int32)
        a = int(rng.integers(ref[0] - 5, ref[-1] + 5)); b = int(rng.integers(a, ref[-1] + 8))
        assert orc.find_by_ref_bounds(pairs, a, b) == full.find_by_ref_bounds(pairs, a, b)


def test_whole_read_pass_matches_calculate_methylation_for_read(full, orc, models):
    """fresh seeds (not the golden's): raw signal + BAM record with indels and clips -> sites, both strands; the oracle's
    restatement, the product's host work-item builder and the modBAM tags all against the reference running live."""
    from nanopolish_amd import api
    from nanopolish_amd.output import modbam_tags
    from nanopolish_amd.synth import synth_cigar_read, BASES
    from oracle.workloads import call_methylation_record, record_reference_segment
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    g = np.random.default_rng(77).integers(0, 4, 5000)
    contig = BASES[g].tobytes().decode()
    n_sites = 0
    for rid in range(300, 306):
        rd = synth_cigar_read(rid, g, models["nucleotide"], span=900, p_ins=0.03, p_del=0.03, max_indel=8)
        cig = cigar_words(rd["cigar_ops"])
        fr = full.read("r%d" % rid, rd["seq"], rd["raw"])
        want = fr.call_methylation(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig)
        got = call_methylation_record(orc, mn, mc, rd["seq"], rd["raw"], rd["rc"], rd["pos"], cig, contig)
        assert got["n_events"] == fr.n_events and tuple(got["scalings"]) == (fr.shift, fr.scale, fr.var)
        assert [(s["start"], s["end"], s["n_motif"], s["ll_unmeth"], s["ll_meth"], s["sequence"]) for s in got["sites"]] == \
            list(zip(want["start"].tolist(), want["end"].tolist(), want["n_motif"].tolist(), want["ll_unmeth"].tolist(),
                     want["ll_meth"].tolist(), want["sequence"]))
        # host work-item builder: same groups survive the CIGAR bounds as the reference's `bounded`
        seg = record_reference_segment(contig, rd["pos"], cig)
        jb = api.cm_build_jobs_cigar(seg, cig, len(rd["seq"]), rd["rc"])
        ms, _ = fr.event_map()
        kept = [int(f) + rd["pos"] for f, (k1, k2) in zip(jb["first"], jb["kpos"])
                if abs(fr.closest_event(int(k2)) - fr.closest_event(int(k1))) > 10]
        assert kept == want["start"].tolist()
        sites = [dict(start_position=int(s), sequence=q, ll_methylated=[m, 0.0], ll_unmethylated=[u, 0.0])
                 for s, q, m, u in zip(want["start"], want["sequence"], want["ll_meth"], want["ll_unmeth"])]
        mm, ml = modbam_tags(sites, cig, rd["pos"], rd["bam_seq"], rd["rc"])
        assert mm == want["Mm"] and ml == want["Ml"].tolist()
        n_sites += len(want["start"])
    assert n_sites > 100


def test_read_at_contig_edges_and_short_reads(full, orc, models):
    """records that start at position 0 / end at the last base of the contig (the fetched segment is clipped), and a read
    too short to score anything"""
    from nanopolish_amd.synth import synth_cigar_read, BASES
    from oracle.workloads import call_methylation_record
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    g = np.random.default_rng(78).integers(0, 4, 1000)
    contig = BASES[g].tobytes().decode()
    for rid, span in ((400, 1000), (401, 1000), (402, 260)):
        rd = synth_cigar_read(rid, g, models["nucleotide"], span=span, soft_clip=(0, 0))
        cig = cigar_words(rd["cigar_ops"])
        fr = full.read("e%d" % rid, rd["seq"], rd["raw"])
        want = fr.call_methylation(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig)
        got = call_methylation_record(orc, mn, mc, rd["seq"], rd["raw"], rd["rc"], rd["pos"], cig, contig)
        assert [(s["start"], s["ll_unmeth"], s["ll_meth"]) for s in got["sites"]] == \
            list(zip(want["start"].tolist(), want["ll_unmeth"].tolist(), want["ll_meth"].tolist()))


def test_eventalign_chain_on_indel_reads_matches_align_read_to_ref(full, orc, models):
    """the oracle's restatement of the segment chain for CIGAR-aligned reads (fresh seeds) against the reference running live"""
    from nanopolish_amd.synth import synth_cigar_read, BASES
    from oracle.workloads import eventalign_record, K
    mn = orc.model(models["nucleotide"])
    g = np.random.default_rng(79).integers(0, 4, 5000)
    contig = BASES[g].tobytes().decode()
    rows = 0
    for rid in range(500, 504):
        rd = synth_cigar_read(rid, g, models["nucleotide"], span=1100, p_ins=0.03, p_del=0.03, max_indel=6)
        cig = cigar_words(rd["cigar_ops"])
        fr = full.read("r%d" % rid, rd["seq"], rd["raw"])
        want = fr.eventalign(rd["rc"], rd["pos"], cig, rd["bam_seq"], contig)
        ev = fr.events(); ms, _ = fr.event_map()
        S = orc.scalings(fr.shift, fr.scale, fr.var)

        def cpu(fwd, rc_s, e1, e2, stride, do_rc):
            return orc.hmm_align(mn, S, ev, orc.sequence_kmer_ranks("nucleotide", fwd, rc_s, K, do_rc), e1, e2, stride, fr.events_per_base)

        got, n_calls = eventalign_record(orc, rd["seq"], rd["rc"], rd["pos"], cig, contig, ms, cpu)
        assert n_calls > 10
        assert got == list(zip(want["ref_position"].tolist(), want["event_idx"].tolist(), want["hmm_state"].tolist()))
        rows += len(got)
    assert rows > 5000
