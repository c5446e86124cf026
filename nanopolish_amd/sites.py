"""Per-site aggregation of call-methylation results: the semantics of the reference's TSV writer
(src/nanopolish_call_methylation.cpp:531-550) followed by scripts/calculate_methylation_frequency.py:16-23,41-49
(default --call-threshold 2.0, groups not split), kept on the device so that the only inter-GPU exchange of the whole job
is one all-reduce(sum) of this table.

table[pos] = (num_reads, called_sites, called_sites_methylated) keyed by the group's first motif position.
The log-likelihood ratio goes through the TSV's "%.2lf" text round trip exactly: printf's correctly rounded two decimals.
"""
import ctypes as C


def site_table_dev(ctx, torch, scores, first, n_motif, n_pos, call_threshold=2.0, jobs=None, read_base=None, stream=None, out=None):
    """The table of a batch on the device (np_site_table_dev).  scores: float32 device tensor, 2 per group (unmethylated,
    methylated; NaN = skipped); first, n_motif: int32 device tensors per group.  Returns an int32 [n_pos, 3] device tensor
    (accumulates into `out` when given)."""
    n_groups = first.numel()
    table = out if out is not None else torch.zeros((n_pos, 3), dtype=torch.int32, device=scores.device)
    if out is None:
        torch.cuda.current_stream().synchronize()        # the zero-fill ran on torch's stream, the kernel runs on the library's
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    rc = ctx.L.np_site_table_dev(ctx.h, C.c_void_p(stream) if stream else None, n_groups, p(scores), p(first), p(n_motif), p(jobs),
                                 p(read_base), float(call_threshold), int(n_pos), p(table))
    ctx._chk(rc, "np_site_table_dev")
    return table


def site_table(torch, first, n_motif, llr, n_pos, call_threshold=2.0):
    """Host mirror (CPU tensors; used by the gloo tests and as the checker of site_table_dev): first, n_motif: int64 tensors
    per group; llr: float64 tensor (NaN = group skipped by the caller rules).  The text round trip IS printf here."""
    vals = llr.detach().cpu().tolist()
    llr2 = torch.tensor([float("%.2f" % v) if v == v and abs(v) != float("inf") else float("nan") for v in vals],
                        dtype=torch.float64, device=llr.device)
    keep = torch.isfinite(llr2) & ~(llr2.abs() < call_threshold * n_motif.to(llr2.dtype))
    idx = first[keep]
    nm = n_motif[keep].to(torch.int32)
    meth = (llr2[keep] > 0).to(torch.int32) * nm
    table = torch.zeros((n_pos, 3), dtype=torch.int32, device=llr.device)
    table[:, 0].index_add_(0, idx, torch.ones_like(nm))
    table[:, 1].index_add_(0, idx, nm)
    table[:, 2].index_add_(0, idx, meth)
    return table
