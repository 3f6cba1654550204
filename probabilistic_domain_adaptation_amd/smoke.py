"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (called by __graft_entry__.smoke()).

A full Mean-Teacher iteration (teacher MC pass + consensus masking, student ELBO forward/backward, L2 regulariser,
Adam, EMA) through the trainer API in fused mode, with injected latent draws."""
import numpy as np
import torch


def run(verbose=True):
    from oracle import punet_oracle as O  # the checker only; the product path never imports it
    from .my_models import ProbabilisticUnet
    from .my_trainer import FusedAdam, MeanTeacherTrainer

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = torch.device("cuda:0")
    nf, B, H, W, N, lr = [32, 64, 64, 96], 2, 32, 32, 4, 1e-3
    cfg = O.PUNetConfig(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False)
    Pnp = O.synth_params(cfg.spec(), 4)
    x, x1, x2, gt = O.synth_batch(B, 1, H, W, 3)
    g = torch.Generator().manual_seed(5)
    eps_t, eps_s = torch.randn(N, B, 6, generator=g), torch.randn(1, B, 6, generator=g)

    model = ProbabilisticUnet(1, 1, nf, 6, 3, 1.0, consensus_masking=True, rl_swap=False, device=dev, seed=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in Pnp.items()})
    opt = FusedAdam(model, lr=lr)
    tr = MeanTeacherTrainer(name=None, train_loader=[(x, x1, x2, gt)], val_loader=None, model=model, optimizer=opt,
                            device=dev, do_consensus_masking=True)
    tr.n_samples = N
    tr.eps_source = lambda kind, n, b: eps_t if kind == "teacher" else eps_s
    from .my_trainer.default_trainer import _Progress
    tr._train_epoch_impl(_Progress(), None, None)
    torch.cuda.synchronize()

    Ps, Pt = O.to_torch(Pnp, True), O.to_torch(Pnp)
    r = O.mean_teacher_step(Ps, Pt, cfg, O.AdamState(Ps, lr=lr), x1, x2, eps_t, eps_s[0], True)
    loss = tr.last_loss.item()
    assert abs(loss - r["loss"].item()) <= 2e-4 * abs(r["loss"].item()), (loss, r["loss"].item())
    worst_s = worst_t = 0.0
    n_all = n_off = 0
    for (k, v), p, q in zip(Ps.items(), model.parameters(), tr.teacher.parameters()):
        ds = (p.detach().cpu() - v.detach()).abs()
        worst_s = max(worst_s, ds.max().item())
        worst_t = max(worst_t, (q.detach().cpu() - Pt[k]).abs().max().item())
        n_all += ds.numel()
        n_off += int((ds > 0.05 * lr).sum())
    # The first Adam step is sign-like (update = lr * g / (|g| + 1e-8)): a gradient entry whose true value is ~0 may
    # come out with either sign within fp32 rounding, which moves that one weight by up to 2 lr.  So: every entry
    # within the 2 lr bound, all but a vanishing fraction within 0.05 lr; the EMA teacher (x 1e-3) within 0.05 lr.
    assert worst_s <= 2.05 * lr and n_off <= 1e-4 * n_all and worst_t <= 0.05 * lr, (worst_s, n_off, n_all, worst_t)
    if verbose:
        print(f"[smoke] mean-teacher step ok: loss {loss:.4f} (oracle {r['loss'].item():.4f}), "
              f"max |dparam| student {worst_s:.2e} ({n_off} of {n_all} entries beyond 0.05 lr) teacher {worst_t:.2e}")
