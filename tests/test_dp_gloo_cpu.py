"""CPU, world_size 2, gloo: the data-parallel host logic of my_trainer/fused.py (gradient all-reduce + 1/world scale,
rank-local shards) and bench.py's max-over-ranks timing.  No GPU compute is called."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    scale = FU.allreduce_grads(g)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py: time = max over ranks
    q.put((rank, scale, g[:4].tolist(), float(g.sum()), t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_grads_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, scale, head, tot, tmax in res:
        assert scale == 0.5
        assert head == [0.0, 3.0, 6.0, 9.0]              # (1 + 2) * arange
        assert tot == 3.0 * sum(range(1000))
        assert abs(tmax - 0.2) < 1e-12


def test_single_process_is_identity():
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    g = torch.ones(8)
    assert FU.allreduce_grads(g) == 1.0 and torch.equal(g, torch.ones(8))
