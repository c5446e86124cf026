"""Candidate-variant qualities from per-read haplotype scores (BASELINE config 4): Variant.quality of score_variant_thresholded
(src/common/nanopolish_variant.cpp:765-799) without its racy early-out -- the sum over reads of
profile_hmm_score_set(variant haplotype) - profile_hmm_score_set(base haplotype), accumulated in double.

Reads shard over GPUs like everywhere else (nanopolish_amd/shard.py), so every rank holds the scores of ITS reads for all
haplotypes and the job's only exchange is one all-reduce(sum, fp64) of the per-variant totals.  The
reference accumulates with `#pragma omp atomic` in thread order, i.e. its own low-order bits are not reproducible: results here are
compared with a tolerance, not bit for bit.
"""


def variant_quality(torch, scores, base_of):
    """scores: float32/float64 tensor [n_haplotypes, n_reads] of profile_hmm_score_set values, NaN where a read does not bound the
    haplotype's window (AlignmentDB::get_event_subsequences skips it); base_of: int64 tensor [n_haplotypes], the row of each
    haplotype's base haplotype (a base haplotype points at itself and gets quality 0).  Returns float64 [n_haplotypes]."""
    s = scores.to(torch.float64)
    d = s - s[base_of]
    d = torch.where(torch.isfinite(d), d, torch.zeros_like(d))          # a read scores a pair or contributes nothing
    return d.sum(dim=1)


def reduce_variant_quality(quality):
    """Sum the per-rank totals in place across the process group (no-op without one)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(quality, op=dist.ReduceOp.SUM)
    return quality
