"""The reference's six PUNet trainers, re-hosted on the HIP hot path.

Class names, constructor kwargs, attributes (`n_samples`, `momentum`, `do_consensus_masking`, `teacher`, ...) and the
`_train_epoch_impl(progress, forward_context, backprop)` / `_validate_impl(forward_context)` contract follow
prob_utils/my_trainer/{punet_trainer,mean_teacher_trainer,fixmatch_trainer,adamt_trainer,adamatch_trainer,
punet_target_trainer}.py.  Two execution modes per iteration:

  * fused (default when the optimizer is a `FusedAdam`): the whole iteration is kernel launches on flat buffers, no
    autograd, no host sync (my_trainer/fused.py); `backprop` is not used because backward + Adam (+ EMA) are fused.
  * drop-in (any torch optimizer): `model.elbo()` returns an autograd-connected scalar, the regulariser is computed with
    torch ops exactly as the reference writes it, and the trainer calls `backprop(loss)`.

Loggers (tensorboard image dumps) are out of scope; `self.logger` is always None.
Each latent draw may be injected through `self.eps_source(kind, n, batch)` for reproducible parity tests.
"""
from __future__ import annotations

import time
from collections import OrderedDict
from copy import deepcopy

import torch

from ..my_models import ProbabilisticUnet, l2_regularisation
from . import fused as FU
from .default_trainer import DefaultTrainer


class _PUNetTrainerBase(DefaultTrainer):
    reg_weight = 1e-5  # mean_teacher_trainer.py:117

    def __init__(self, fused=None, graph=False, **kwargs):
        """fused / graph are extensions: fused (default: the optimizer is a FusedAdam) runs the iteration as kernel launches
        on flat buffers; graph=True additionally captures that iteration once per input shape into a hipGraph and replays it
        (FU.GraphedStep: bit-identical results, ~1 ms of host time per step instead of 10-30 ms)."""
        super().__init__(**kwargs)
        self._kwargs = kwargs
        self.fused = isinstance(self.optimizer, FU.FusedAdam) if fused is None else fused
        if self.fused and not isinstance(self.optimizer, FU.FusedAdam):
            raise ValueError("fused=True needs a FusedAdam optimizer")
        self.graph = bool(graph)
        if self.graph and not self.fused:
            raise ValueError("graph=True needs the fused execution mode (a FusedAdam optimizer)")
        self._graphs = OrderedDict()  # captured steps, least recently used first; at most `max_graphs` are kept
        self.max_graphs = 4           # (each holds a private memory pool the size of one step's activations)
        self.sigmoid = torch.nn.Sigmoid()
        self.eps_source = None  # callable(kind, n, batch) -> [n, batch, L] tensor, or None for fresh draws
        self.last_loss = None

    def _eps(self, kind, n, batch):
        return None if self.eps_source is None else self.eps_source(kind, n, batch)

    def _eps_drawn(self, kind, n, batch):
        """Like _eps, but never None: a graph replay takes its latent draws as inputs, so fresh ones are drawn here."""
        e = self._eps(kind, n, batch)
        m = self.model
        if e is None:
            return torch.randn(n, batch, m.latent_dim, device=m.device, generator=m.generator)
        return e.to(device=m.device, dtype=torch.float32).reshape(n, batch, m.latent_dim)

    def _graphed(self, kind, tensors, body, **ema):
        """Replay (capturing first, once per kind / shapes / lr truthiness) the fused step `body` on `tensors`."""
        nets = [m for m in (self.model, getattr(self, "teacher", None)) if m is not None]
        # a graph is captured against the ENGINES (and native runtimes) of the models it runs - weight planes, tables,
        # workspaces, arenas they own - in one conv arithmetic: the arithmetic is part of the key (a graph of each mode can
        # coexist: mixed_precision switches modes on entry to and exit from fit()), and the captured entry holds STRONG
        # references to those engines, compared by identity - never by id(), which CPython reuses once an object is freed
        # (a stale graph replaying through freed planes).  An entry whose engines are no longer the models' is dropped.
        key = (kind, bool(self._lr()), tuple(m.engine.conv_mode for m in nets)) + tuple((k, tuple(v.shape)) for k, v in tensors.items())
        engines = tuple(m.engine for m in nets)
        gs = self._graphs.get(key)
        if gs is not None and not (len(gs.owners) == len(engines) and all(a is b for a, b in zip(gs.owners, engines))):
            del self._graphs[key]
            gs = None
        if gs is None:
            static = {k: v.to(device=self.model.device, dtype=torch.float32).clone() for k, v in tensors.items()}
            gs = self._graphs[key] = FU.GraphedStep(body, static, self.optimizer, **ema)
            gs.owners = engines  # keeps them (and through _keep their buffers) alive as long as this graph is
            gs.owner_runtimes = tuple(getattr(m, "_native_rts", {}).get((m.engine.conv_mode, bool(m.rl_swap))) for m in nets)
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        self._graphs.move_to_end(key)
        return gs.step(warmup_iteration=ema.get("warmup_iteration"), **tensors)

    def _lr(self):
        return [pm["lr"] for pm in self.optimizer.param_groups][0]

    def _reg_loss(self):
        m = self.model
        return l2_regularisation(m.posterior) + l2_regularisation(m.prior) + l2_regularisation(m.fcomb.layers)

    def _dropin_loss(self, x, y, z, kind):
        """forward(x, y, True); elbo(y[, z]); loss = -elbo + 1e-5 * reg   (e.g. mean_teacher_trainer.py:113-117)."""
        self.model.forward(x, y, training=True)
        e = self._eps(kind, 1, x.shape[0])
        elbo = self.model.elbo(y, z, eps=e) if z is not None else self.model.elbo(y, eps=e)
        return -elbo + self.reg_weight * self._reg_loss()

    def _bookkeeping(self, progress):
        self._iteration += 1
        progress.update(1)
        return self._iteration >= self.max_iteration

    def _validation_metric(self, samples, y, gt=None):
        """1 - dice_score(mean prediction, target) (e.g. mean_teacher_trainer.py:166-173), reduced on the device."""
        eng = self.model.engine
        d = eng.dice_score(samples, y.to(samples.device, torch.float32))
        dg = eng.dice_score(samples, gt.to(samples.device, torch.float32)) if gt is not None else None
        return d, dg


class PUNetTrainer(_PUNetTrainerBase):
    """Supervised source training (punet_trainer.py:13-97)."""

    def _sample(self, n_samples=16):
        return [self.model.sample() for _ in range(n_samples)]

    def _batch(self, batch):
        x, y = batch
        return x.to(self.device), y.to(self.device), None

    def _train_epoch_impl(self, progress, forward_context, backprop):
        self.model.train()
        n_iter, t0 = 0, time.time()
        for batch in self.train_loader:
            x, y, z = self._batch(batch)
            if self.fused and self.graph:
                t = {"x": x, "y": y, "eps": self._eps_drawn("student", 1, x.shape[0])}
                if z is not None:
                    t["z"] = z
                self.last_loss = self._graphed("supervised", t, lambda s: FU.fused_supervised_step(
                    self.model, self.optimizer, s["x"], s["y"], s.get("z"), s["eps"], self.reg_weight))
            elif self.fused:
                self.last_loss = FU.fused_supervised_step(self.model, self.optimizer, x, y, z, self._eps("student", 1, x.shape[0]),
                                                          self.reg_weight)
            else:
                self.optimizer.zero_grad()
                with forward_context():
                    loss = self._dropin_loss(x, y, z, "student")
                backprop(loss)
                self.last_loss = loss.detach()
            n_iter += 1
            if self._bookkeeping(progress):
                break
        return (time.time() - t0) / max(n_iter, 1)

    def _validate_impl(self, forward_context):
        self.model.eval()
        metric_val = loss_val = dice_metric = 0.0
        with torch.no_grad():
            for batch in self.val_loader:
                x, y, z = self._batch(batch)
                with forward_context():
                    loss = self._dropin_loss(x, y, z, "student")
                    mypred, _ = self.model.sample_consensus(8, eps=self._eps("val", 8, x.shape[0]))  # punet_trainer.py:70-78
                    d, _ = self._validation_metric(mypred, y)
                dice_metric += d
                loss_val += loss.item()
                metric_val += 1.0 - d
        n = len(self.val_loader)
        print(f"The Average Dice Score for the Current Epoch is {dice_metric / n}")
        # what the reference hands to its tensorboard logger (punet_trainer.py:94-96); kept as an attribute instead
        self.last_validation = {"metric": metric_val / n, "loss": loss_val / n, "dice": dice_metric / n}
        return metric_val / n


class PseudoTrainerPUNet(PUNetTrainer):
    """PUNet training on offline pseudo-labels with consensus weights: batches are (x, y, z)
    (punet_target_trainer.py:92-183)."""

    def _batch(self, batch):
        x, y, z = batch
        return x.to(self.device), y.to(self.device), z.to(self.device)


class _SelfTrainingBase(_PUNetTrainerBase):
    def __init__(self, do_consensus_masking=False, **kwargs):
        super().__init__(**kwargs)
        self.n_samples = 16  # mean_teacher_trainer.py:36 - an attribute, not a ctor argument
        self.do_consensus_masking = do_consensus_masking

    def _pseudo_label_net(self):
        return self.model

    def _pseudo_labels(self, inputs, upper_thres=0.9, lower_thres=0.1):
        y, z = FU.pseudo_labels(self._pseudo_label_net(), inputs, self.n_samples, self.do_consensus_masking,
                                self._eps("teacher", self.n_samples, inputs.shape[0]), upper_thres, lower_thres)
        if self.do_consensus_masking:
            z = z.to(torch.int64)  # torch.where(consensus == 1, 1, 0) (mean_teacher_trainer.py:85-86)
        return y, z

    def sample_from_model(self):
        """Mean of n_samples sigmoid(model.sample()) from the model's current features (mean_teacher_trainer.py:90-93)."""
        y, _ = self.model.sample_consensus(self.n_samples, eps=self._eps("val", self.n_samples, self.model._state["x"].shape[0]))
        return y

    def _validation_uses_consensus(self):
        return True

    def _validate_impl(self, forward_context):
        self.model.eval()
        metric_val = loss_val = dice_metric = gt_metric_val = 0.0
        with torch.no_grad():
            for x, x1, x2, gt_ in self.val_loader:
                x1, x2 = x1.to(self.device), x2.to(self.device)
                y, z = self._pseudo_labels(x1)
                with forward_context():
                    loss = self._dropin_loss(x2, y, z if self._validation_uses_consensus() else None, "student")
                    samples = self.sample_from_model()
                    d, dg = self._validation_metric(samples, y, gt_)
                dice_metric += d
                loss_val += loss.item()
                metric_val += 1.0 - d
                gt_metric_val += 1.0 - dg
        n = len(self.val_loader)
        print(f"The Average Dice Score for the Current Epoch is {dice_metric / n}")
        # what the reference hands to its tensorboard logger (mean_teacher_trainer.py:185-190); kept as an attribute instead
        self.last_validation = {"metric": metric_val / n, "loss": loss_val / n, "dice": dice_metric / n,
                                "gt_metric": gt_metric_val / n, "y": y, "z": z, "samples": samples}
        return metric_val / n


class MeanTeacherTrainer(_SelfTrainingBase):
    """Mean-Teacher self-training with consensus weighting / masking (mean_teacher_trainer.py:15-191)."""

    def __init__(self, ckpt_model=None, ckpt_teacher=None, momentum=0.999, momentum_update_is_activated=False,
                 do_consensus_masking=False, **kwargs):
        super().__init__(do_consensus_masking=do_consensus_masking, **kwargs)
        self.momentum = momentum
        self.ckpt_model, self.ckpt_teacher = ckpt_model, ckpt_teacher
        self.momentum_update_is_activated = momentum_update_is_activated
        with torch.no_grad():
            self.teacher = deepcopy(self.model)  # :40
            for param in self.teacher.parameters():
                param.requires_grad = False
        if self.ckpt_model is not None:
            self.model.load_state_dict(torch.load(self.ckpt_model, weights_only=False)["model_state"])
        if self.ckpt_teacher is not None:
            self.teacher.load_state_dict(torch.load(self.ckpt_teacher, weights_only=False)["model_state"])

    def _pseudo_label_net(self):
        return self.teacher

    def sample_from_teacher(self, teacher_inputs, upper_thres=0.9, lower_thres=0.1):
        return self._pseudo_labels(teacher_inputs, upper_thres, lower_thres)

    def _momentum_update(self):
        FU.ema_update(self.model, self.teacher, self.momentum)  # :52-55

    def save_checkpoint(self, name, current_metric, best_metric, train_time=0.0, **extra):
        super().save_checkpoint(name, current_metric, best_metric, train_time=train_time,
                                teacher_state=self.teacher.state_dict(), **extra)

    def load_checkpoint(self, checkpoint="best"):
        save_dict = super().load_checkpoint(checkpoint)
        self.teacher.load_state_dict(save_dict["teacher_state"])
        return save_dict

    def _train_epoch_impl(self, progress, forward_context, backprop):
        self.model.train()
        n_iter, t0 = 0, time.time()
        for x, x1, x2, gt_ in self.train_loader:
            x1, x2 = x1.to(self.device), x2.to(self.device)
            B = x1.shape[0]
            if self.fused and self.graph:
                t = {"x1": x1, "x2": x2, "eps_pl": self._eps_drawn("teacher", self.n_samples, B),
                     "eps_s": self._eps_drawn("student", 1, B)}
                n, masking, mom = self.n_samples, self.do_consensus_masking, self.momentum
                self.last_loss, y, z, _ = self._graphed(("mean_teacher", n, masking, mom), t, lambda s: FU.fused_self_training_step(
                    self.model, self.teacher, self.optimizer, s["x1"], s["x2"], n, masking, s["eps_pl"], s["eps_s"], mom,
                    self.reg_weight), ema=bool(self._lr()), ema_momentum=mom, extra_state=[self.teacher.flat_params])
                if self._lr():
                    self.momentum_update_is_activated = True
            elif self.fused:
                self.last_loss, y, z, _ = FU.fused_self_training_step(
                    self.model, self.teacher, self.optimizer, x1, x2, self.n_samples, self.do_consensus_masking,
                    self._eps("teacher", self.n_samples, B), self._eps("student", 1, B), self.momentum, self.reg_weight)
                if self._lr():
                    self.momentum_update_is_activated = True
            else:
                with torch.no_grad():
                    lr = self._lr()
                    y, z = self.sample_from_teacher(x1)
                self.optimizer.zero_grad()
                with forward_context():
                    loss = self._dropin_loss(x2, y, z, "student")
                backprop(loss)
                self.last_loss = loss.detach()
                if lr:  # :126-131
                    if not self.momentum_update_is_activated:
                        print("Activating momentum update in iteration : ", self._iteration)
                        self.momentum_update_is_activated = True
                    with torch.no_grad():
                        self._momentum_update()
            n_iter += 1
            if self._bookkeeping(progress):
                break
        return (time.time() - t0) / max(n_iter, 1)


class FixMatchTrainer(_SelfTrainingBase):
    """FixMatch self-training: pseudo-labels from the model itself on the weak view (fixmatch_trainer.py:15-160)."""

    def __init__(self, ckpt_model=None, source_distribution=None, do_consensus_masking=False, **kwargs):
        super().__init__(do_consensus_masking=do_consensus_masking, **kwargs)
        self.ckpt_model = ckpt_model
        self.source_distribution = None if source_distribution is None else \
            torch.as_tensor(source_distribution, dtype=torch.float32, device=self.device)
        if self.ckpt_model is not None:
            self.model.load_state_dict(torch.load(self.ckpt_model, map_location="cpu", weights_only=False)["model_state"])

    def sample_from_weak_model(self, weak_inputs, upper_thres=0.9, lower_thres=0.1):
        return self._pseudo_labels(weak_inputs, upper_thres, lower_thres)

    def _train_epoch_impl(self, progress, forward_context, backprop):
        self.model.train()
        n_iter, t0 = 0, time.time()
        for x, x1, x2, gt_ in self.train_loader:
            x1, x2 = x1.to(self.device), x2.to(self.device)
            B = x1.shape[0]
            if self.fused and self.graph:
                t = {"x1": x1, "x2": x2, "eps_pl": self._eps_drawn("teacher", self.n_samples, B),
                     "eps_s": self._eps_drawn("student", 1, B)}
                n, masking = self.n_samples, self.do_consensus_masking
                self.last_loss, y, z, _ = self._graphed(("fixmatch", n, masking), t, lambda s: FU.fused_self_training_step(
                    self.model, self.model, self.optimizer, s["x1"], s["x2"], n, masking, s["eps_pl"], s["eps_s"], None,
                    self.reg_weight, self.source_distribution))
            elif self.fused:
                self.last_loss, y, z, _ = FU.fused_self_training_step(
                    self.model, self.model, self.optimizer, x1, x2, self.n_samples, self.do_consensus_masking,
                    self._eps("teacher", self.n_samples, B), self._eps("student", 1, B), None, self.reg_weight,
                    self.source_distribution)
            else:
                with torch.no_grad():
                    y, z = self.sample_from_weak_model(x1)
                if self.source_distribution is not None:
                    y, _ = FU.distribution_alignment(y, self.source_distribution, self.model.engine)  # :77-84
                self.optimizer.zero_grad()
                with forward_context():
                    loss = self._dropin_loss(x2, y, z, "student")
                backprop(loss)
                self.last_loss = loss.detach()
            n_iter += 1
            if self._bookkeeping(progress):
                break
        return (time.time() - t0) / max(n_iter, 1)


class _JointBase(_SelfTrainingBase):
    """zip(source_loader, target_loader); loss = (supervised + target) / 2 with one backward."""

    def __init__(self, source_train_loader, target_train_loader, do_consensus_masking=False, **kwargs):
        self.source_train_loader, self.target_train_loader = source_train_loader, target_train_loader
        train_loader = source_train_loader if len(source_train_loader) < len(target_train_loader) else target_train_loader
        kwargs.pop("train_loader", None)
        super().__init__(do_consensus_masking=do_consensus_masking, train_loader=train_loader, **kwargs)
        self.target_val_loader = None

    def _use_consensus(self):
        return True

    def _ema_momentum(self):
        return None

    def _ema_rule(self):
        """(momentum cap, iteration for the warm-up rule) of the fused step; (None, None) = no EMA."""
        return None, None

    def _train_epoch_impl(self, progress, forward_context, backprop):
        self.model.train()
        n_iter, t0 = 0, time.time()
        for (xs, ys), (xt, xt1, xt2, yt) in zip(self.source_train_loader, self.target_train_loader):
            xs, ys = xs.to(self.device), ys.to(self.device)
            xt1, xt2 = xt1.to(self.device), xt2.to(self.device)
            B = xt1.shape[0]
            use_z = self._use_consensus()
            mom = self._ema_momentum()
            cap, warm = self._ema_rule()  # fused: the momentum rule is evaluated on the device
            net = self._pseudo_label_net()
            if self.fused and self.graph:
                t = {"xs": xs, "ys": ys, "xt1": xt1, "xt2": xt2, "eps_sup": self._eps_drawn("sup", 1, xs.shape[0]),
                     "eps_pl": self._eps_drawn("teacher", self.n_samples, B), "eps_s": self._eps_drawn("student", 1, B)}
                n, masking = self.n_samples, self.do_consensus_masking
                extra = [net.flat_params] if net is not self.model else []
                self.last_loss, y, z = self._graphed(("joint", n, masking, use_z, cap), t, lambda s: FU.fused_joint_step(
                    self.model, net, self.optimizer, s["xs"], s["ys"], s["xt1"], s["xt2"], n, masking, s["eps_sup"], s["eps_pl"],
                    s["eps_s"], cap, use_z, self.reg_weight, warm), ema=cap is not None, ema_momentum=cap or 0.0,
                    warmup_iteration=warm, extra_state=extra)
            elif self.fused:
                self.last_loss, y, z = FU.fused_joint_step(
                    self.model, net, self.optimizer, xs, ys, xt1, xt2, self.n_samples,
                    self.do_consensus_masking, self._eps("sup", 1, xs.shape[0]), self._eps("teacher", self.n_samples, B),
                    self._eps("student", 1, B), cap, use_z, self.reg_weight, warm)
            else:
                self.optimizer.zero_grad()
                with forward_context():
                    supervised_loss = self._dropin_loss(xs, ys, None, "sup")
                with torch.no_grad():
                    y, z = self._pseudo_labels(xt1)
                with forward_context():
                    target_loss = self._dropin_loss(xt2, y, z if use_z else None, "student")
                loss = (supervised_loss + target_loss) / 2
                backprop(loss)
                self.last_loss = loss.detach()
                if mom is not None:
                    with torch.no_grad():
                        FU.ema_update(self.model, self.teacher, mom)
            n_iter += 1
            if self._bookkeeping(progress):
                break
        return (time.time() - t0) / max(n_iter, 1)


class AdaMTTrainer(_JointBase):
    """Joint Mean-Teacher (adamt_trainer.py:15-189): EMA momentum min(1 - 1/(it+1), m), ungated."""

    def __init__(self, source_train_loader, target_train_loader, momentum=0.999, do_consensus_masking=False, **kwargs):
        super().__init__(source_train_loader, target_train_loader, do_consensus_masking, **kwargs)
        self.momentum = momentum
        with torch.no_grad():
            self.teacher = deepcopy(self.model)
            for param in self.teacher.parameters():
                param.requires_grad = False

    def _pseudo_label_net(self):
        return self.teacher

    def sample_from_teacher(self, weak_inputs, upper_thres=0.9, lower_thres=0.1):
        return self._pseudo_labels(weak_inputs, upper_thres, lower_thres)

    def _ema_momentum(self):
        return min(1 - 1 / (self._iteration + 1), self.momentum)  # adamt_trainer.py:41

    def _ema_rule(self):
        return self.momentum, self._iteration

    def _momentum_update(self):
        FU.ema_update(self.model, self.teacher, self._ema_momentum())

    def _validate_impl(self, forward_context):
        self.target_val_loader = self.val_loader  # adamt_trainer.py:148
        return super()._validate_impl(forward_context)

    def save_checkpoint(self, name, current_metric, best_metric, train_time=0.0, **extra):
        super().save_checkpoint(name, current_metric, best_metric, train_time=train_time,
                                teacher_state=self.teacher.state_dict(), **extra)

    def load_checkpoint(self, checkpoint="best"):
        save_dict = super().load_checkpoint(checkpoint)
        self.teacher.load_state_dict(save_dict["teacher_state"])
        return save_dict


class AdaMatchTrainer(_JointBase):
    """Joint FixMatch (adamatch_trainer.py:14-169): uses the consensus only while lr is truthy (:87-94)."""

    def __init__(self, source_train_loader, target_train_loader, do_consensus_masking=False, **kwargs):
        super().__init__(source_train_loader, target_train_loader, do_consensus_masking, **kwargs)
        self.consensus_masking_is_activated = False

    def sample_from_weak_model(self, weak_inputs, upper_thres=0.9, lower_thres=0.1):
        return self._pseudo_labels(weak_inputs, upper_thres, lower_thres)

    def _use_consensus(self):
        if self._lr():
            if not self.consensus_masking_is_activated:
                print("Activating consensus masking the reconstruction loss in iteration :", self._iteration)
                self.consensus_masking_is_activated = True
            return True
        return False

    def _validation_uses_consensus(self):
        return self.consensus_masking_is_activated  # adamatch_trainer.py:138: elbo(y, z) only once training switched it on

    def _validate_impl(self, forward_context):
        self.target_val_loader = self.val_loader  # adamatch_trainer.py:127
        return super()._validate_impl(forward_context)
