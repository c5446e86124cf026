"""Multi-GPU layout: reads are independent, so each rank (one process per GPU) takes a contiguous range of read
ids and runs the whole per-read pass on its own device with no data-path collective.  The job's only exchange is
the final site-level reduction: one all-reduce(sum) of the dense per-site table (nanopolish_amd/sites.py) over
RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests)."""


def shard_read_ids(n_total, rank, world):
    """Contiguous [lo, hi) range of read ids for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_site_table(table):
    """Sum the per-rank site tables in place across the process group (no-op without one)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    return table
