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


def _worker_split(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    # what elbo_grad_pass(overlap=True) does at the after_unet hook: the leading slice goes out early, asynchronously
    started = {"numel": 384, "work": dist.all_reduce(g[:384], op=dist.ReduceOp.SUM, async_op=True)}
    g[384:] += 1.0  # the rest of the backward still writes the remainder
    scale = FU.allreduce_grads(g, started)
    q.put((rank, scale, g[:2].tolist(), g[384:386].tolist(), float(g.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_grads_overlapped_slice_world2():
    """Two-part reduction (U-Net slice early + remainder) == one reduction of the final buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, scale, head, mid, tot in res:
        assert scale == 0.5
        assert head == [0.0, 3.0]
        assert mid == [3.0 * 384 + 2.0, 3.0 * 385 + 2.0]
        assert tot == 3.0 * sum(range(1000)) + 2.0 * (1000 - 384)


def test_unet_slice_of_the_flat_layout():
    from probabilistic_domain_adaptation_amd.engine import ParamLayout
    L = ParamLayout(1, 1, [64, 128, 256, 512], 6, 3)
    names = [nm for nm, off, n, _ in L.entries if off < L.unet_numel]
    assert L.unet_numel == 11690688 and all(nm.startswith("unet.") for nm in names)
    assert all(not nm.startswith("unet.") for nm, off, n, _ in L.entries if off >= L.unet_numel)
    assert L.unet_numel % 4 == 0


class _PointwiseModel:
    """Stand-in with the two calls mc_prediction makes; 'prediction' = sigmoid of the (standardised) tile, on the CPU."""
    device = torch.device("cpu")

    def forward(self, raw, segm, training=False):
        self._raw = raw

    def sample_consensus(self, n, eps=None):
        return torch.sigmoid(self._raw), None


def _tiled_reference(img, block, halo):
    """The oracle's (recalled) predict_with_halo blocking with the pointwise stand-in as the per-tile prediction."""
    import numpy as np
    from oracle import punet_oracle as O
    out = np.zeros_like(img)
    for (y0, y1, x0, x1) in O.halo_tiles(img.shape, (block, block), (halo, halo)):
        t = O.standardize_np(O.load_halo_block(img, y0, x0, (block, block), (halo, halo)))
        out[y0:y1, x0:x1] = (1.0 / (1.0 + np.exp(-t.astype("float64"))))[halo:halo + y1 - y0, halo:halo + x1 - x0]
    return out


def _worker_tiles(rank, world, port, q):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilistic_domain_adaptation_amd.my_predictions import predict_image
    img = np.random.default_rng(3).normal(size=(200, 168)).astype("float32")
    out = predict_image(_PointwiseModel(), img, prior_samples=2, block_shape=(64, 64), halo=(8, 8), tile_batch=2)
    q.put((rank, float(np.abs(out - _tiled_reference(img, 64, 8)).max()), float(out.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_tiled_prediction_sharded_over_ranks_world2():
    """predict_image deals its tile batches to the ranks and joins the disjoint maps with one all_reduce: every rank ends
    with the full map, equal to the single-process tiling (halo read with reflect padding, per-tile standardisation, halo
    cropped) of a pointwise stand-in model."""
    import numpy as np
    from probabilistic_domain_adaptation_amd.my_predictions import predict_image
    img = np.random.default_rng(3).normal(size=(200, 168)).astype("float32")
    single = predict_image(_PointwiseModel(), img, prior_samples=2, block_shape=(64, 64), halo=(8, 8), tile_batch=2)
    assert np.abs(single - _tiled_reference(img, 64, 8)).max() < 1e-6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tiles, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, tot in res:
        assert err < 1e-6
        assert abs(tot - float(single.sum())) < 1e-2
