"""GPU: data-parallel correctness with 2 ranks sharing the one GPU of the box (gloo carries the all-reduce; on the
8-GPU node the same code path uses RCCL).  After one fused step the replicas are bit-identical and equal to a
single-process emulation that averages the two shards' gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
NF = [32, 64, 64, 96]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_model(dev):
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam
    cfg = O.PUNetConfig(1, 1, NF, 6, 3, 1.0, True, False)
    Pnp = O.synth_params(cfg.spec(), 4)
    m = ProbabilisticUnet(1, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
    return m, FusedAdam(m, lr=1e-3)


def _shard(rank):
    from oracle import punet_oracle as O
    x, x1, x2, y = O.synth_batch(2, 1, 32, 32, 30 + rank)
    eps = torch.randn(1, 2, 6, generator=torch.Generator().manual_seed(40 + rank))
    return x2, y, eps


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    m, opt = _setup_model(dev)
    x2, y, eps = _shard(rank)
    loss = FU.fused_supervised_step(m, opt, x2.to(dev), y.to(dev), None, eps)
    torch.cuda.synchronize()
    q.put((rank, m.flat_params.cpu().numpy(), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process_emulation():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]), "replicas diverged"
    # single-process emulation: mean of the two shards' gradients, then the same Adam kernel
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    dev = torch.device("cuda:0")
    m, opt = _setup_model(dev)
    eng = m.engine
    gsum = torch.zeros_like(opt.grad_flat)
    for r in range(2):
        x2, y, eps = _shard(r)
        g = torch.zeros_like(opt.grad_flat)
        FU.elbo_grad_pass(m, g, x2.to(dev), y.to(dev), None, eps, 1.0, False)
        reg, norms = eng.l2_reg(m.flat_params)
        eng.l2_reg_add_grad(m.flat_params, g, norms, 1e-5)
        gsum += g
    opt.grad_flat.copy_(gsum)
    opt.fused_step(None, grad_scale=0.5)
    torch.cuda.synchronize()
    assert np.array_equal(m.flat_params.cpu().numpy(), res[0][1])


# ------------------------------------------------------------------------------------------------------------------
# fit() of a real trainer under 2-rank data parallelism (both ranks on the one GPU; gloo carries the collectives)
# ------------------------------------------------------------------------------------------------------------------
def _worker_fit(rank, world, port, q, save_root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.my_models import ProbabilisticUnet
    from probabilistic_domain_adaptation_amd.my_trainer import FusedAdam, MeanTeacherTrainer
    # seed=None-like start: every rank builds DIFFERENT weights; fit() must replicate rank 0's
    m = ProbabilisticUnet(1, 1, NF, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=100 + rank)
    w_start = m.flat_params.clone()
    data = [O.synth_batch(2, 1, 32, 32, 200 + i) for i in range(4)]  # one common loader, sharded by fit()
    opt = FusedAdam(m, lr=1e-3)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=0)
    tr = MeanTeacherTrainer(name="dp", train_loader=data, val_loader=data, model=m, optimizer=opt, device=dev, lr_scheduler=sched,
                            do_consensus_masking=True, save_root=save_root, shard_loaders=True)
    tr.n_samples = 4
    g = torch.Generator().manual_seed(50 + rank)
    tr.eps_source = lambda kind, n, b: torch.randn(n, b, 6, generator=g)
    best = tr.fit(epochs=2)
    torch.cuda.synchronize()
    q.put((rank, w_start.cpu().numpy(), m.flat_params.cpu().numpy(), tr.teacher.flat_params.cpu().numpy(), opt.param_groups[0]["lr"], best,
           len(tr.train_loader), opt.step_count, tr.last_validation["metric"]))
    dist.barrier()
    dist.destroy_process_group()


def test_mean_teacher_fit_two_ranks(tmp_path):
    """fit() of MeanTeacherTrainer with 2 ranks: different initial weights are replaced by rank 0's, each rank trains on its
    shard of the common loader (2 of 4 batches per epoch), validates on its shard, and both end with bit-equal student and
    teacher, the same learning rate and best metric (all-reduced), and ONE set of checkpoints written by rank 0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fit, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, s0, w0, t0, lr0, b0, n0, k0, v0), (_, s1, w1, t1, lr1, b1, n1, k1, v1) = res
    assert not np.array_equal(s0, s1)                       # the ranks started apart
    assert np.array_equal(w0, w1) and np.array_equal(t0, t1)  # and ended as replicas
    assert lr0 == lr1 and b0 == b1 and n0 == n1 == 2 and k0 == k1 == 4
    assert v0 != v1                                          # rank-local validation metrics differ (different shards) ...
    assert min(v0, v1) - 1e-12 <= b0 <= max(v0, v1) + 1e-12 or b0 <= max(v0, v1)  # ... the returned best is the ranks' mean of some epoch
    files = sorted(os.listdir(os.path.join(str(tmp_path), "checkpoints", "dp")))
    assert files == ["best.pt", "latest.pt"], files
    ck = torch.load(os.path.join(str(tmp_path), "checkpoints", "dp", "latest.pt"), weights_only=False)
    assert ck["epoch"] == 2 and ck["iteration"] == 4 and "teacher_state" in ck


# ------------------------------------------------------------------------------------------------------------------
# the REAL backend of the 8-GPU run (RCCL through ProcessGroupNCCL: its own stream, event hand-over, async work handles) on
# the one GPU of the box: a one-rank group, with the trainers' world size forced to 2 so that every data-parallel code path
# runs (the U-Net slice's asynchronous all-reduce under the encoders' backward, the remainder before the update, the graph
# cut at both).  A one-rank all-reduce is the identity, so the step must equal the single-process step at gradient scale 1/2.
def _rccl_worker(port, q, save_root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import copy
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from oracle import punet_oracle as O
    from probabilistic_domain_adaptation_amd.my_trainer import fused as FU
    x, x1, x2, y = [v.to(dev) for v in O.synth_batch(2, 1, 32, 32, 3)]
    g = torch.Generator().manual_seed(2)
    e_pl, e_s = torch.randn(4, 2, 6, generator=g).to(dev), torch.randn(1, 2, 6, generator=g).to(dev)

    def run(graph, steps=3):
        m, opt = _setup_model(dev)
        t = copy.deepcopy(m)
        body = lambda s: FU.fused_self_training_step(m, t, opt, s["x1"], s["x2"], 4, True, s["eps_pl"], s["eps_s"], 0.99, 1e-5)
        inp = {"x1": x1.clone(), "x2": x2.clone(), "eps_pl": e_pl.clone(), "eps_s": e_s.clone()}
        if graph:
            gs = FU.GraphedStep(body, inp, opt, ema=True, ema_momentum=0.99, extra_state=[t.flat_params])
            nseg = len(gs.segments)
            for _ in range(steps):
                out = gs.step()
        else:
            nseg = 0
            for _ in range(steps):
                out = body(inp)
        torch.cuda.synchronize()
        return m.flat_params.cpu().numpy(), t.flat_params.cpu().numpy(), float(out[0]), nseg

    # (a) no process group: the single-process step with the 1/2 scale forced in by hand
    real_world = FU._world
    FU._world = lambda: 2
    calls = []
    real_allreduce = dist.all_reduce

    class _Done:
        def wait(self):
            return True

    def identity(t, op=None, async_op=False):
        calls.append(t.numel())
        return _Done() if async_op else None
    dist.all_reduce = identity
    want = run(False)
    dist.all_reduce = real_allreduce
    # (b) the same through RCCL, eagerly and as graph segments with the collectives between them
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eager = run(False)
    graphed = run(True)
    # (c) the control plane of fit() over the same backend (state broadcast, metric / loss all-reduce on device tensors, loader
    # sharding, rank-0 checkpoints): two epochs of a real trainer complete and leave finite replicas
    from probabilistic_domain_adaptation_amd.my_trainer import MeanTeacherTrainer, default_trainer as DT
    real_dp_world = DT.dp_world
    DT.dp_world = lambda: 2
    m, opt = _setup_model(dev)
    data = [O.synth_batch(2, 1, 32, 32, 200 + i) for i in range(4)]
    tr = MeanTeacherTrainer(name="rccl", train_loader=data, val_loader=data, model=m, optimizer=opt, device=dev, do_consensus_masking=True,
                            save_root=save_root, shard_loaders=True)
    tr.n_samples = 4
    best = tr.fit(epochs=2)
    torch.cuda.synchronize()
    fit = (len(tr.train_loader), opt.step_count, bool(torch.isfinite(m.flat_params).all()), bool(np.isfinite(best)))
    DT.dp_world = real_dp_world
    FU._world = real_world
    dist.barrier()
    dist.destroy_process_group()
    q.put((want, eager, graphed, calls, fit))


def test_rccl_backend_one_rank_group_eager_and_graph_segments(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q, str(tmp_path)))
    p.start()
    want, eager, graphed, calls, fit = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert len(calls) == 6 and calls[0] + calls[1] == calls[2] + calls[3]  # two collectives per step: U-Net slice + remainder
    assert graphed[3] == 3, "the capture is cut at both all-reduces"
    for got in (eager, graphed):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    assert fit == (2, 4, True, True), fit  # half of the 4 batches per epoch (rank 0 of "2"), two epochs
    assert sorted(os.listdir(os.path.join(str(tmp_path), "checkpoints", "rccl"))) == ["best.pt", "latest.pt"]
