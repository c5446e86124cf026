"""Per-site aggregation of call-methylation results: the semantics of the reference's
scripts/calculate_methylation_frequency.py:16-23,41-49 (default --call-threshold 2.0, groups not split),
kept on the device so the only inter-GPU exchange of the whole job is one all-reduce(sum) of this table.

table[pos] = (num_reads, called_sites, called_sites_methylated) keyed by the group's first motif position.
"""


def site_table(torch, first, n_motif, llr, n_pos, call_threshold=2.0):
    """first, n_motif: int64 tensors per group; llr: float64 tensor (NaN = group skipped by the caller rules)."""
    llr2 = torch.round(llr * 100.0) / 100.0                 # the TSV carries %.2lf (call_methylation.cpp:545)
    keep = torch.isfinite(llr2) & ~(llr2.abs() < call_threshold * n_motif.to(llr2.dtype))
    idx = first[keep]
    nm = n_motif[keep].to(torch.int32)
    meth = (llr2[keep] > 0).to(torch.int32) * nm
    table = torch.zeros((n_pos, 3), dtype=torch.int32, device=llr.device)
    table[:, 0].index_add_(0, idx, torch.ones_like(nm))
    table[:, 1].index_add_(0, idx, nm)
    table[:, 2].index_add_(0, idx, meth)
    return table
