#!/usr/bin/env python3
"""Probe (manual, through gpurun): does RCCL run here at all?  (1) world_size 1 on the one GPU: communicator set-up + the site table's
all_reduce(sum, int32) + a float64 max + barrier; (2) world_size 2 with both ranks on the SAME GPU -- expected to be refused
("duplicate GPU"), tried so that the refusal is on record.  Prints one JSON line per attempt.
    python tools/rccl_probe.py"""
import json, os, socket, subprocess, sys


def child():
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    out = dict(world=world, rank=rank)
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        t = torch.arange(3 * 5413, dtype=torch.int32, device="cuda").reshape(-1, 3) * (rank + 1)
        want = t.clone() * sum(r + 1 for r in range(world)) // (rank + 1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        m = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda")
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        out.update(ok=bool(torch.equal(t, want) and m.item() == world), backend=dist.get_backend())
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        out.update(ok=False, error=repr(e)[:300])
    if rank == 0:
        print(json.dumps(out), flush=True)


def main():
    if "RANK" in os.environ:
        return child()
    for world in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        for p in procs:
            try:
                p.wait(timeout=120)
            except subprocess.TimeoutExpired:
                p.kill(); print(json.dumps(dict(world=world, ok=False, error="timeout")), flush=True)


if __name__ == "__main__":
    main()
