"""world_size-2 run of the N>1 path on CPU (gloo): read sharding + the final site-table all-reduce give the same
table as a single process over all reads.  LLRs come from the CPU oracle here (no GPU in this test)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanopolish_amd.shard import shard_read_ids, reduce_site_table
from nanopolish_amd.sites import site_table

N_READS, READ_LEN = 6, 700


def _rank_table(lo, hi):
    from oracle import Oracle, load_models
    from cases import call_methylation_read, synth_read
    models = load_models(); orc = Oracle()
    mn, mc = orc.model(models["nucleotide"]), orc.model(models["cpg"])
    first, nm, llr = [], [], []
    for rid in range(lo, hi):
        r = call_methylation_read(orc, mn, mc, synth_read(rid, models["nucleotide"], L=READ_LEN))
        first += list(r["first"]); nm += [j["n_motif"] for j in r["jobs"]]
        llr += list(r["meth"].astype(np.float64) - r["unmeth"])
    return site_table(torch, torch.tensor(first, dtype=torch.int64), torch.tensor(nm, dtype=torch.int64),
                      torch.tensor(llr, dtype=torch.float64), READ_LEN)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_read_ids(N_READS, rank, world)
    t = reduce_site_table(_rank_table(lo, hi))
    if rank == 0:
        torch.save(t, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_everything_once():
    for n in (0, 1, 7, 100000):
        for w in (1, 2, 3, 8):
            r = [shard_read_ids(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_two_rank_site_reduction_equals_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "table.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    want = _rank_table(0, N_READS)
    assert want[:, 0].sum() > 10
    assert torch.equal(got, want)
