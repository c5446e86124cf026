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


# ---- genome-keyed table: reads that overlap on a reference, keyed (contig, start, end) as the reference keys a site ---------------------
# (src/nanopolish_call_methylation.cpp:532-550, scripts/calculate_methylation_frequency.py:16-23,41-49).  table: int32 [n_pos, 6];
# columns 0-2 = (num_reads, called_sites, called_sites_methylated) of the key (start = row, end = the end of start's genome cluster),
# columns 3-5 = the same for the key (start = the start of end's cluster, end = row) whose end comes BEFORE the cluster's (a read that stops
# inside a cluster).  np_site_table_genome_dev in include/np_hmm.h has the argument for why these two are all the keys there are.
MOTIFS = {"cpg": ("CG",), "gpc": ("GC",), "dam": ("GATC",), "dcm": ("CCAGG", "CCTGG")}


def genome_site_index_dev(ctx, torch, genome, contig_off, alphabet="cpg", stream=None):
    """The motif sites of the resident genome as a rank structure (np_genome_site_index_dev), built once per genome: returns
    (site_mask int64-viewed uint64 [ceil(n_pos/64)], word_rank int32-viewed uint32 [ceil(n_pos/64) + 1], n_sites int).  Pass the pair as `index`
    to site_table_genome_dev for a table with one row per SITE."""
    from . import api
    n_pos = int(genome.numel()); n_words = (n_pos + 63) // 64
    mask = torch.empty(max(1, n_words), dtype=torch.int64, device=genome.device)
    rank = torch.empty(n_words + 1, dtype=torch.int32, device=genome.device)
    total = torch.zeros(1, dtype=torch.int64, device=genome.device)
    torch.cuda.current_stream().synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = ctx.L.np_genome_site_index_dev(ctx.h, C.c_void_p(stream) if stream else None, p(genome), p(contig_off), int(contig_off.numel()) - 1,
                                        api.alphabet_id(alphabet), n_pos, p(mask), p(rank), p(total))
    ctx._chk(rc, "np_genome_site_index_dev")
    ctx.sync()
    return mask, rank, int(total.item())


def site_table_genome_dev(ctx, torch, scores, first, last, n_motif, jobs, read_base, genome, contig_off, alphabet="cpg", min_separation=10,
                          call_threshold=2.0, stream=None, out=None, overflow=None, index=None):
    """The genome-keyed table of a batch on the device (np_site_table_genome_dev).  scores: float32, 2 per group (NaN = skipped); first / last /
    n_motif: int32 per group, segment-relative; jobs: the batch's work items (2 per group: the read of a group); read_base: int64 per read, the
    genome offset of its segment; genome: uint8 device tensor (the contigs, concatenated); contig_off: int64 device tensor [n_contigs + 1].
    index = genome_site_index_dev's (site_mask, word_rank, n_sites): one row per motif SITE (np_site_table_genome_indexed_dev) instead of per base.
    Returns (table int32 [n_pos or n_sites, 6], overflow uint64-as-int64 [1]); accumulates into `out` / `overflow` when given."""
    from . import api
    n_groups = first.numel()
    n_pos = int(genome.numel())
    fresh = out is None or overflow is None
    n_rows = n_pos if index is None else int(index[2])
    table = out if out is not None else torch.zeros((n_rows, 6), dtype=torch.int32, device=scores.device)
    if table.shape[0] != n_rows:
        raise ValueError("table has %d rows, the key space has %d" % (table.shape[0], n_rows))
    ovf = overflow if overflow is not None else torch.zeros(1, dtype=torch.int64, device=scores.device)
    if fresh:
        torch.cuda.current_stream().synchronize()        # the zero-fill ran on torch's stream, the kernel runs on the library's
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    head = (ctx.h, C.c_void_p(stream) if stream else None, n_groups, p(scores), p(first), p(last), p(n_motif), p(jobs),
            p(read_base), p(genome), p(contig_off), int(contig_off.numel()) - 1, api.alphabet_id(alphabet),
            int(min_separation), float(call_threshold), n_pos)
    if index is None:
        ctx._chk(ctx.L.np_site_table_genome_dev(*head, p(table), p(ovf)), "np_site_table_genome_dev")
    else:
        ctx._chk(ctx.L.np_site_table_genome_indexed_dev(*head, p(index[0]), p(index[1]), p(table), p(ovf)), "np_site_table_genome_indexed_dev")
    return table, ovf


def motif_sites(genome, contig_off, alphabet="cpg"):
    """bool array over the concatenated contigs: a recognition site of the alphabet STARTS here and lies inside its contig
    (Alphabet::is_motif_match on the contig, src/common/nanopolish_alphabet.h)"""
    import numpy as np
    g = genome if isinstance(genome, (bytes, bytearray)) else bytes(genome)
    hit = np.zeros(len(g), bool)
    for c in range(len(contig_off) - 1):
        lo, hi = int(contig_off[c]), int(contig_off[c + 1])
        seq = g[lo:hi]
        for m in MOTIFS[alphabet]:
            mb = m.encode(); i = seq.find(mb)
            while i >= 0:
                hit[lo + i] = True
                i = seq.find(mb, i + 1)
    return hit


def _cluster_bounds(hit, contig_off, min_separation):
    """per motif site: (start, end) of its genome cluster -- sites chained by gaps <= min_separation inside one contig (basemods.cpp:306-320)"""
    import numpy as np
    cstart = np.full(len(hit), -1, np.int64); cend = np.full(len(hit), -1, np.int64)
    for c in range(len(contig_off) - 1):
        pos = np.flatnonzero(hit[int(contig_off[c]):int(contig_off[c + 1])]) + int(contig_off[c])
        if len(pos) == 0:
            continue
        brk = np.flatnonzero(np.diff(pos) > min_separation)
        starts = np.concatenate([[0], brk + 1]); ends = np.concatenate([brk, [len(pos) - 1]])
        for a, b in zip(starts, ends):
            cstart[pos[a:b + 1]] = pos[a]; cend[pos[a:b + 1]] = pos[b]
    return cstart, cend


def site_table_genome(torch, start, end, n_motif, llr, genome, contig_off, alphabet="cpg", min_separation=10, call_threshold=2.0, compact=False):
    """Host mirror of site_table_genome_dev (CPU; the checker of the device kernel and the per-rank table of the gloo tests).  start / end: int64
    GENOME positions of every group's first / last motif site; llr: float64 (NaN = skipped).  The text round trip IS printf here.
    compact: one row per motif SITE (row = the site's ordinal in the genome), the layout of site_table_genome_dev(index=...).
    Returns (table int32 [n_pos or n_sites, 6], overflow int)."""
    import numpy as np
    hit = motif_sites(genome, contig_off, alphabet)
    cstart, cend = _cluster_bounds(hit, contig_off, min_separation)
    n_pos = len(hit)
    ordinal = np.cumsum(hit) - 1
    table = torch.zeros((int(hit.sum()) if compact else n_pos, 6), dtype=torch.int32)
    vals = llr.detach().cpu().tolist()
    llr2 = np.array([float("%.2f" % v) if v == v and abs(v) != float("inf") else float("nan") for v in vals], np.float64)
    s = start.cpu().numpy().astype(np.int64); e = end.cpu().numpy().astype(np.int64); nm = n_motif.cpu().numpy().astype(np.int64)
    keep = np.isfinite(llr2) & ~(np.abs(llr2) < call_threshold * nm) & (s >= 0) & (e < n_pos) & (e >= s)
    overflow = 0
    t = table.numpy()
    for i in np.flatnonzero(keep):
        if cend[s[i]] == e[i]:
            row, col = s[i], 0
        elif cstart[e[i]] == s[i]:
            row, col = e[i], 3
        else:
            overflow += 1
            continue
        if compact:
            row = ordinal[row]
        t[row, col] += 1; t[row, col + 1] += nm[i]
        if llr2[i] > 0:
            t[row, col + 2] += nm[i]
    return table, overflow


def genome_table_rows(table, genome, contig_off, alphabet="cpg", min_separation=10):
    """The keys of a genome-keyed table as the frequency script lists them: sorted [(start, end, num_reads, called_sites, called_sites_methylated)]
    with GENOME positions (the caller maps them to (contig, position)).  A table with one row per motif site (fewer rows than bases) is
    recognised by its length."""
    import numpy as np
    t = table.cpu().numpy()
    hit = motif_sites(genome, contig_off, alphabet)
    cstart, cend = _cluster_bounds(hit, contig_off, min_separation)
    pos_of = np.arange(len(hit)) if t.shape[0] == len(hit) else np.flatnonzero(hit)
    if len(pos_of) != t.shape[0]:
        raise ValueError("table has %d rows; the genome has %d bases and %d motif sites" % (t.shape[0], len(hit), int(hit.sum())))
    rows = []
    for r in np.flatnonzero(t[:, 0] > 0):
        g = int(pos_of[r])
        rows.append((g, int(cend[g]), int(t[r, 0]), int(t[r, 1]), int(t[r, 2])))
    for r in np.flatnonzero(t[:, 3] > 0):
        g = int(pos_of[r])
        rows.append((int(cstart[g]), g, int(t[r, 3]), int(t[r, 4]), int(t[r, 5])))
    return sorted(rows)
