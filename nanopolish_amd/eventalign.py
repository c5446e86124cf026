"""Output side of eventalign (host code mirroring the reference's TSV writer): the rows np_eventalign_dev produces --
(ref_position, event_idx, hmm_state) per aligned event -- printed as emit_event_alignment_tsv does with its default options
(src/alignment/nanopolish_eventalign.cpp:227-243,398-487): read index instead of read name, the model scaled to the read
(not --scale-events), no signal indices, no samples.

    contig  position  reference_kmer  read_index  strand  event_index  event_level_mean  event_stdv  event_length
    model_kmer  model_mean  model_stdv  standardized_level
"""
import numpy as np

TSV_COLUMNS = ["contig", "position", "reference_kmer", "read_index", "strand", "event_index", "event_level_mean", "event_stdv",
               "event_length", "model_kmer", "model_mean", "model_stdv", "standardized_level"]
_RC = str.maketrans("ACGT", "TGCA")


def eventalign_tsv_header():
    return "\t".join(TSV_COLUMNS) + "\n"


def _fmt(x, nd):
    """printf("%.<nd>lf") of a value that may be inf / nan (a 'B' row divides by a zero model stdv)"""
    x = float(x)
    if x != x:
        return "-nan" if np.signbit(x) else "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    return "%.*f" % (nd, x)


def format_eventalign_tsv(rows, contig_name, ref_seq, ref_offset, read_idx, is_rev, event_mean, event_stdv, event_length, sample_rate,
                          model, shift, scale, var, k=6):
    """rows: dict(ref_position, event_idx, hmm_state) as CallMethylationBatch.eventalign() returns them (absolute positions).
    ref_seq / ref_offset: the reference segment of the record (contig[pos .. bam_endpos]) and pos.
    event_mean / event_stdv / event_length: the read's detected events (event_t::mean, ::stdv, ::length in samples);
    model: the base pore model's tables (level_mean, level_stdv); shift / scale / var: the read's calibrated scalings.
    Returns the TSV lines (no header)."""
    f32 = np.float32
    lm, ls = model["level_mean"], model["level_stdv"]
    sqrt_var = np.sqrt(np.float64(var))
    out = []
    for rp, ei, st in zip(rows["ref_position"], rows["event_idx"], rows["hmm_state"]):
        rp, ei = int(rp), int(ei)
        ref_kmer = ref_seq[rp - ref_offset:rp - ref_offset + k]
        is_b = chr(int(st)) == "B"
        # HMMInputSequence::get_kmer(kmer_idx, k, rc): the reference k-mer, or its reverse complement for a reverse-strand read
        model_kmer = "N" * k if is_b else (ref_kmer[::-1].translate(_RC) if is_rev else ref_kmer)
        ev_mean = f32(event_mean[ei])                                   # get_unscaled_level (drift 0 on the R9 path)
        # SquiggleEvent::duration = (float)(event_t::length / sample_rate)   (squiggle_read.cpp:246-247)
        duration = f32(np.float64(f32(event_length[ei])) / np.float64(sample_rate))
        model_mean = f32(0.0); model_stdv = f32(0.0)
        if not is_b:
            rank = 0
            for ch in model_kmer:
                rank = rank * 4 + "ACGT".index(ch)
            # get_scaled_gaussian_from_pore_model_state (squiggle_read.h:217-226): double math, float store
            model_mean = f32(np.float64(scale) * lm[rank] + np.float64(shift))
            model_stdv = f32(ls[rank] * np.float64(var))
        with np.errstate(divide="ignore", invalid="ignore"):
            standard_level = f32(np.float64(ev_mean - model_mean) / (sqrt_var * np.float64(model_stdv)))
        out.append("%s\t%d\t%s\t%d\tt\t%d\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n" % (
            contig_name, rp, ref_kmer, read_idx, ei, _fmt(ev_mean, 2), _fmt(f32(event_stdv[ei]), 3), _fmt(duration, 5), model_kmer,
            _fmt(model_mean, 2), _fmt(model_stdv, 2), _fmt(standard_level, 2)))
    return out
