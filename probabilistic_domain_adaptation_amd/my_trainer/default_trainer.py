"""Minimal `DefaultTrainer` with the surface the reference's trainers rely on.

The reference subclasses the third-party `torch_em.trainer.DefaultTrainer` (e.g. mean_teacher_trainer.py:15), which is
not vendored (environment.yaml:18) and not installed here.  Only the contract that the hot path touches is rebuilt
(SURVEY.md §8b-2): constructor attributes, `fit()` driving `_train_epoch_impl(progress, forward_context, backprop)` and
`_validate_impl(forward_context)`, `ReduceLROnPlateau`-style scheduler stepping on the validation metric, and
`best.pt` / `latest.pt` checkpoints holding `model_state` / `optimizer_state` (+ `teacher_state` from subclasses,
mean_teacher_trainer.py:57-65).  AMP is not emulated: this build computes in fp32 (parity mode), so
`mixed_precision` is accepted and ignored.
"""
from __future__ import annotations

import contextlib
import os
import time

import torch


class _Progress:
    def __init__(self, total=None, verbose=False):
        self.n, self.total, self.verbose = 0, total, verbose

    def update(self, k=1):
        self.n += k

    def set_description(self, *a, **k):
        pass


class DefaultTrainer:
    def __init__(self, name=None, train_loader=None, val_loader=None, model=None, loss=None, optimizer=None, metric=None,
                 device=None, lr_scheduler=None, log_image_interval=100, mixed_precision=True, early_stopping=None,
                 logger=None, logger_kwargs=None, save_root=None, compile_model=None, rank=None, **unused):
        self.name = name
        self.train_loader, self.val_loader = train_loader, val_loader
        self.model, self.loss, self.optimizer, self.metric = model, loss, optimizer, metric
        self.device = torch.device(device) if device is not None else getattr(model, "device", torch.device("cuda"))
        self.lr_scheduler = lr_scheduler
        self.log_image_interval = log_image_interval
        self.mixed_precision = mixed_precision  # accepted, ignored (fp32 parity mode)
        self.early_stopping = early_stopping
        self.logger = None  # tensorboard loggers are out of scope
        self.save_root = save_root
        self._iteration, self._epoch = 0, 0
        self.max_iteration, self.max_epoch = 10 ** 9, 10 ** 9
        self._best_metric = float("inf")
        self.train_time = 0.0

    # ------------------------------------------------------------------ checkpoints
    @property
    def checkpoint_folder(self):
        root = self.save_root if self.save_root is not None else "."
        return os.path.join(root, "checkpoints", self.name or "punet")

    def save_checkpoint(self, name, current_metric, best_metric, train_time=0.0, **extra_save_dict):
        os.makedirs(self.checkpoint_folder, exist_ok=True)
        save_dict = {"iteration": self._iteration, "epoch": self._epoch, "best_epoch": getattr(self, "_best_epoch", 0),
                     "best_metric": best_metric, "current_metric": current_metric, "train_time": train_time,
                     "model_state": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
                     "optimizer_state": self.optimizer.state_dict() if self.optimizer is not None else None}
        if self.lr_scheduler is not None:
            save_dict["scheduler_state"] = self.lr_scheduler.state_dict()
        for k, v in extra_save_dict.items():
            save_dict[k] = {kk: vv.detach().cpu().contiguous() for kk, vv in v.items()} if isinstance(v, dict) else v
        torch.save(save_dict, os.path.join(self.checkpoint_folder, f"{name}.pt"))

    def load_checkpoint(self, checkpoint="best"):
        path = checkpoint if os.path.isfile(str(checkpoint)) else os.path.join(self.checkpoint_folder, f"{checkpoint}.pt")
        save_dict = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(save_dict["model_state"])
        if self.optimizer is not None and save_dict.get("optimizer_state") is not None:
            self.optimizer.load_state_dict(save_dict["optimizer_state"])
        if self.lr_scheduler is not None and "scheduler_state" in save_dict:
            self.lr_scheduler.load_state_dict(save_dict["scheduler_state"])
        self._iteration, self._epoch = save_dict["iteration"], save_dict["epoch"]
        self._best_metric = save_dict["best_metric"]
        self.train_time = save_dict.get("train_time", 0.0)
        return save_dict

    def _initialize(self, iterations, load_from_checkpoint, epochs=None):
        if iterations is not None:
            self.max_iteration = self._iteration + iterations if load_from_checkpoint is None else iterations
        if epochs is not None:
            self.max_epoch = epochs
        best_metric = float("inf")
        if load_from_checkpoint is not None:
            self.load_checkpoint(load_from_checkpoint)
            best_metric = self._best_metric
        return best_metric

    # ------------------------------------------------------------------ the loop
    def _backprop(self, loss):
        loss.backward()
        self.optimizer.step()

    def fit(self, iterations=None, load_from_checkpoint=None, epochs=None, save_every_kth_epoch=None,
            overwrite_training=True, progress=None):
        if not overwrite_training and os.path.exists(os.path.join(self.checkpoint_folder, "latest.pt")) \
                and load_from_checkpoint is None:
            load_from_checkpoint = "latest"
        best_metric = self._initialize(iterations, load_from_checkpoint, epochs)
        progress = progress or _Progress(self.max_iteration)
        forward_context = contextlib.nullcontext
        t0 = time.time()
        while self._iteration < self.max_iteration and self._epoch < self.max_epoch:
            self._train_epoch_impl(progress, forward_context, self._backprop)
            current_metric = float("nan")
            if self.val_loader is not None:
                current_metric = self._validate_impl(forward_context)
                if self.lr_scheduler is not None:
                    self.lr_scheduler.step(current_metric)
            self._epoch += 1
            total = self.train_time + time.time() - t0
            if self.save_root is not None or self.name is not None:
                if current_metric < best_metric:
                    best_metric = current_metric
                    self._best_epoch = self._epoch
                    self.save_checkpoint("best", current_metric, best_metric, train_time=total)
                self.save_checkpoint("latest", current_metric, best_metric, train_time=total)
        self._best_metric = best_metric
        return best_metric

    def _train_epoch_impl(self, progress, forward_context, backprop):
        raise NotImplementedError

    def _validate_impl(self, forward_context):
        raise NotImplementedError
