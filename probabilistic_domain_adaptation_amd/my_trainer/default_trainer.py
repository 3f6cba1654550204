"""Minimal `DefaultTrainer` with the surface the reference's trainers rely on.

The reference subclasses the third-party `torch_em.trainer.DefaultTrainer` (e.g. mean_teacher_trainer.py:15), which is
not vendored (environment.yaml:18) and not installed here.  Only the contract that the hot path touches is rebuilt
(SURVEY.md §8b-2): constructor attributes, `fit()` driving `_train_epoch_impl(progress, forward_context, backprop)` and
`_validate_impl(forward_context)`, `ReduceLROnPlateau`-style scheduler stepping on the validation metric, and
`best.pt` / `latest.pt` checkpoints holding `model_state` / `optimizer_state` (+ `teacher_state` from subclasses,
mean_teacher_trainer.py:57-65).  AMP is not emulated: this build computes in fp32 (parity mode), so
`mixed_precision=True` selects the fast conv arithmetic $PUNET_MIXED_PRECISION_MODE (default "f16x3"); the default is False.
"""
from __future__ import annotations

import contextlib
import os
import time
import warnings

import torch
import torch.distributed as dist


class _Progress:
    def __init__(self, total=None, verbose=False):
        self.n, self.total, self.verbose = 0, total, verbose

    def update(self, k=1):
        self.n += k

    def set_description(self, *a, **k):
        pass


# ---------------------------------------------------------------------------------------- data-parallel control plane
def dp_world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dp_rank():
    return dist.get_rank() if dp_world() > 1 else 0


def _comm_device(device):
    """Collectives of an RCCL ("nccl") group need device tensors, gloo takes CPU tensors."""
    return device if dp_world() > 1 and dist.get_backend() == "nccl" else torch.device("cpu")


def allreduce_scalars(values, device, op="mean"):
    """All-reduce a short list of host floats over the ranks (the validation metric and the logged loss, so
    that ReduceLROnPlateau, the best-checkpoint rule and early stopping decide identically everywhere).  fp64 on the
    wire; identity in a single process."""
    w = dp_world()
    if w == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.SUM)
    if op == "mean":
        t /= w
    return [float(v) for v in t.cpu().tolist()]


def broadcast_tensors(tensors, device, src=0):
    """Make every rank hold rank `src`'s values of the given (flat) tensors, in place."""
    if dp_world() == 1:
        return
    cd = _comm_device(device)
    for t in tensors:
        if t.device == cd:
            dist.broadcast(t, src=src)
        else:
            buf = t.detach().to(cd)
            dist.broadcast(buf, src=src)
            with torch.no_grad():
                t.copy_(buf)


class RankShard:
    """View of a loader that yields every world-th batch, starting at `rank`; all ranks get the same number of batches
    (the remainder is dropped), so that the per-step gradient all-reduce always has all its participants.
    The SLOW fallback of `shard_loader`: every rank still loads (and augments) the whole epoch and discards
    (world - 1) / world of it - used only for loaders that are not torch DataLoaders (lists, generators with a length)."""

    def __init__(self, loader, rank=None, world=None):
        self.loader = loader
        self.rank = dp_rank() if rank is None else rank
        self.world = dp_world() if world is None else world

    def __len__(self):
        return len(self.loader) // self.world

    def __iter__(self):
        n = len(self)
        for i, batch in enumerate(self.loader):
            if i >= n * self.world:
                break
            if i % self.world == self.rank:
                yield batch


class ShardedBatchSampler:
    """Every world-th batch of INDICES of a batch sampler, starting at `rank` (remainder dropped): the sharding happens
    before anything is loaded, so a rank reads and augments only its own 1 / world of the epoch.  A shuffling
    base sampler must draw the same permutation on every rank (seed its generator identically, as for a DistributedSampler)."""

    def __init__(self, batch_sampler, rank, world):
        self.batch_sampler, self.rank, self.world = batch_sampler, rank, world

    def __len__(self):
        return len(self.batch_sampler) // self.world

    def __iter__(self):
        n = len(self) * self.world
        for i, idx in enumerate(self.batch_sampler):
            if i >= n:
                break
            if i % self.world == self.rank:
                yield idx


def shard_loader(loader, rank=None, world=None):
    """The rank's shard of a common loader: a torch DataLoader is rebuilt around a ShardedBatchSampler (index level, no wasted
    loading); anything else falls back to RankShard (batch level)."""
    rank = dp_rank() if rank is None else rank
    world = dp_world() if world is None else world
    from torch.utils.data import DataLoader
    if isinstance(loader, DataLoader) and getattr(loader, "batch_sampler", None) is not None \
            and not isinstance(loader.batch_sampler, ShardedBatchSampler):
        kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory,
                  timeout=loader.timeout, worker_init_fn=loader.worker_init_fn, generator=loader.generator,
                  persistent_workers=loader.persistent_workers)
        if loader.num_workers > 0:
            kw.update(prefetch_factor=loader.prefetch_factor, multiprocessing_context=loader.multiprocessing_context)
        return DataLoader(loader.dataset, batch_sampler=ShardedBatchSampler(loader.batch_sampler, rank, world), **kw)
    return RankShard(loader, rank, world)


class DefaultTrainer:
    def __init__(self, name=None, train_loader=None, val_loader=None, model=None, loss=None, optimizer=None, metric=None,
                 device=None, lr_scheduler=None, log_image_interval=100, mixed_precision=False, early_stopping=None,
                 logger=None, logger_kwargs=None, save_root=None, compile_model=None, rank=None, shard_loaders=False,
                 **unused):
        """shard_loaders (extension, data parallel): False = the caller hands every rank its own loader shard (e.g. a
        DistributedSampler) - fit() then checks that all ranks see the same number of batches; True = every rank was handed
        the SAME loader and takes every world-th batch of it (shard_loader: at the index level for a torch DataLoader, so
        nothing is loaded twice; RankShard, batch level, for other iterables)."""
        self.name = name
        self.train_loader, self.val_loader = train_loader, val_loader
        self.model, self.loss, self.optimizer, self.metric = model, loss, optimizer, metric
        self.device = torch.device(device) if device is not None else getattr(model, "device", torch.device("cuda"))
        self.lr_scheduler = lr_scheduler
        self.log_image_interval = log_image_interval
        # mixed_precision: the reference's scripts pass True (LIVECell/livecell_mt.py:107): torch_em then runs the forward under
        # fp16 autocast with a GradScaler [recalled].  Here True selects the fast conv arithmetic $PUNET_MIXED_PRECISION_MODE
        # (default "f16x3": fp16 planes with per-tensor power-of-two scaling - logits within 1e-4 of the fp32 reference, 1.5x the
        # parity mode's throughput; "f16" = one fp16 plane, the literal analogue of autocast's arithmetic; "off" = ignore).  The
        # DEFAULT here is False (torch_em's is True [recalled]): a trainer built without the argument runs the parity arithmetic.
        # The switch is made ON THE CALLER'S MODEL, in place (the arithmetic is a property of the model's engine, not of a call;
        # see _apply_mixed_precision): fit() undoes it when it returns, restore_precision() does so on request.
        self.mixed_precision = bool(mixed_precision)
        self._mp_restore = {}  # id(model) -> (model, the arithmetic it had before this trainer switched it)
        self._apply_mixed_precision()
        self.early_stopping = early_stopping
        self.logger = None  # tensorboard loggers are out of scope
        self.save_root = save_root
        self.shard_loaders = shard_loaders
        self._iteration, self._epoch = 0, 0
        self.max_iteration, self.max_epoch = 10 ** 9, 10 ** 9
        self._best_metric = float("inf")
        self.train_time = 0.0
        self.last_loss = None
        self.last_epoch_loss = float("nan")  # logged training loss of the last epoch's last step, mean over ranks

    # ------------------------------------------------------------------ mixed precision
    def _mp_mode(self):
        mode = os.environ.get("PUNET_MIXED_PRECISION_MODE", "f16x3")
        return None if (mode == "off" or not self.mixed_precision) else mode

    def _apply_mixed_precision(self):
        """mixed_precision=True: student AND teacher (the trainers' own deep copy, or one the caller assigned later) compute in
        $PUNET_MIXED_PRECISION_MODE from here on; what they computed in before is remembered for restore_precision().  Autocast
        in the reference is scoped to the trainer's passes; here the arithmetic belongs to the model's engine, so the switch is
        visible on the model object while the trainer is at work - fit() re-applies it on entry and restores on exit.
        mixed_precision=False: nothing is switched, but a student and a teacher that compute in DIFFERENT arithmetics are
        reported (the fused steps then leave the native runtime for the Python engine: fused.native_ok)."""
        mode = self._mp_mode()
        nets = [m for m in (self.model, getattr(self, "teacher", None)) if m is not None and hasattr(m, "set_conv_mode")]
        if mode is None:
            if len({m.engine.conv_mode for m in nets}) > 1:
                import warnings
                warnings.warn("student and teacher compute in different conv arithmetics ("
                              + ", ".join(m.engine.conv_mode for m in nets) + "): pseudo-labels and the gradient pass use different "
                              "precisions and the fused steps take the Python engine; call teacher.set_conv_mode(...) to align them")
            return
        for m in nets:
            if m.engine.conv_mode != mode:
                self._mp_restore.setdefault(id(m), (m, m.engine.conv_mode))
                m.set_conv_mode(mode)

    def restore_precision(self):
        """Give every model this trainer switched for mixed_precision=True its previous arithmetic back (fit() calls this when it
        returns; call it yourself after driving _train_epoch_impl directly, before a parity evaluation or an export of the same
        model object).  A teacher deep-copied from an already switched student is put back to the student's arithmetic."""
        mode = self._mp_mode()
        back = None
        for m, prev in self._mp_restore.values():
            m.set_conv_mode(prev)
            if m is self.model:
                back = prev
        t = getattr(self, "teacher", None)
        if back is not None and t is not None and id(t) not in self._mp_restore and hasattr(t, "set_conv_mode") \
                and t.engine.conv_mode == mode:
            t.set_conv_mode(back)
        self._mp_restore = {}

    # ------------------------------------------------------------------ checkpoints
    @property
    def checkpoint_folder(self):
        root = self.save_root if self.save_root is not None else "."
        return os.path.join(root, "checkpoints", self.name or "punet")

    def _init_record(self):
        """[recalled] torch_em stores the trainer's constructor arguments under "init" so that a checkpoint can rebuild
        its trainer; loaders / model / optimizer objects are not serialisable here, so only their class names go in."""
        return {"trainer_class": type(self).__module__ + "." + type(self).__name__, "name": self.name,
                "model_class": type(self.model).__name__ if self.model is not None else None,
                "optimizer_class": type(self.optimizer).__name__ if self.optimizer is not None else None,
                "device": str(self.device), "mixed_precision": self.mixed_precision,
                "log_image_interval": self.log_image_interval}

    def save_checkpoint(self, name, current_metric, best_metric, train_time=0.0, **extra_save_dict):
        """torch_em-style `best.pt` / `latest.pt`.  Under data parallelism rank 0 writes (the replicas are identical),
        the others wait at the barrier so that nobody reads a half-written file."""
        if dp_rank() == 0:
            os.makedirs(self.checkpoint_folder, exist_ok=True)
            save_dict = {"iteration": self._iteration, "epoch": self._epoch, "best_epoch": getattr(self, "_best_epoch", 0),
                         "best_metric": best_metric, "current_metric": current_metric, "train_time": train_time,
                         "init": self._init_record(),
                         "model_state": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
                         "optimizer_state": self.optimizer.state_dict() if self.optimizer is not None else None}
            if self.lr_scheduler is not None:
                save_dict["scheduler_state"] = self.lr_scheduler.state_dict()
            for k, v in extra_save_dict.items():
                save_dict[k] = {kk: vv.detach().cpu().contiguous() for kk, vv in v.items()} if isinstance(v, dict) else v
            path = os.path.join(self.checkpoint_folder, f"{name}.pt")
            torch.save(save_dict, path + ".tmp")
            os.replace(path + ".tmp", path)
        if dp_world() > 1:
            dist.barrier()

    def load_checkpoint(self, checkpoint="best"):
        path = checkpoint if os.path.isfile(str(checkpoint)) else os.path.join(self.checkpoint_folder, f"{checkpoint}.pt")
        save_dict = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(save_dict["model_state"])
        if self.optimizer is not None and save_dict.get("optimizer_state") is not None:
            try:
                self.optimizer.load_state_dict(save_dict["optimizer_state"])
            except (KeyError, ValueError) as e:  # a foreign optimizer's state: keep the weights, restart the moments.  (Not
                # RuntimeError: a HIP / copy failure must not be reported as a foreign checkpoint.)  FusedAdam validates every
                # entry before it touches its buffers; any other optimizer may be left half-loaded, so it is reset explicitly.
                reset = getattr(self.optimizer, "reset_state", None)
                if reset is not None:
                    reset()
                else:
                    self.optimizer.state.clear()
                warnings.warn(f"optimizer_state of {path} does not fit {type(self.optimizer).__name__} ({e!r}); "
                              "continuing with fresh optimizer state")
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

    # ------------------------------------------------------------------ data parallel
    def _replicated_tensors(self):
        """Everything that must be identical on all ranks before the first step: student (and teacher) weights and the
        optimizer's moments.  Flat buffers when the objects have them, per-tensor otherwise."""
        out = []
        for mod in (self.model, getattr(self, "teacher", None)):
            if mod is None:
                continue
            flat = getattr(mod, "flat_params", None)
            out += [flat] if flat is not None else [p.data for p in mod.parameters()]
        opt = self.optimizer
        if opt is not None:
            if hasattr(opt, "exp_avg"):
                out += [opt.exp_avg, opt.exp_avg_sq]
            else:
                for stt in opt.state.values():
                    out += [v for v in stt.values() if torch.is_tensor(v) and v.numel() > 1]
        return out

    def _dp_prepare(self):
        """Start of fit() under torch.distributed: replicate rank 0's state (a model built with seed=None differs per
        process), shard or verify the loaders."""
        if dp_world() == 1:
            return
        broadcast_tensors(self._replicated_tensors(), self.device, src=0)
        step = getattr(self.optimizer, "step_count", None)
        if step is not None:
            self.optimizer.step_count = int(allreduce_scalars([step], self.device, "max")[0])
        for attr in ("train_loader", "val_loader", "source_train_loader", "target_train_loader"):
            ld = getattr(self, attr, None)
            if ld is None or isinstance(ld, RankShard) or isinstance(getattr(ld, "batch_sampler", None), ShardedBatchSampler):
                continue
            if self.shard_loaders:
                setattr(self, attr, shard_loader(ld))
            else:
                n = len(ld)
                lo, = allreduce_scalars([n], self.device, "min")
                hi, = allreduce_scalars([n], self.device, "max")
                if lo != hi:
                    raise RuntimeError(f"{attr}: ranks see {int(lo)}..{int(hi)} batches per epoch; every rank must step the "
                                       "same number of times (pass equal shards, or shard_loaders=True with one common loader)")

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
        self._dp_prepare()
        progress = progress or _Progress(self.max_iteration)
        forward_context = contextlib.nullcontext
        t0 = time.time()
        self._apply_mixed_precision()  # (again: a previous fit() restored the models on its way out; also covers a teacher assigned since)
        try:
            best_metric = self._fit_loop(best_metric, progress, forward_context, t0, save_every_kth_epoch)
        finally:
            self.restore_precision()
        self._best_metric = best_metric
        return best_metric

    def _fit_loop(self, best_metric, progress, forward_context, t0, save_every_kth_epoch):
        while self._iteration < self.max_iteration and self._epoch < self.max_epoch:
            self._train_epoch_impl(progress, forward_context, self._backprop)
            current_metric = float("nan")
            if self.val_loader is not None:
                current_metric = self._validate_impl(forward_context)
            loss = float(self.last_loss) if self.last_loss is not None else float("nan")
            # ONE scalar all-reduce per epoch: every rank steps the scheduler / picks "best" on the same numbers
            current_metric, self.last_epoch_loss = allreduce_scalars([current_metric, loss], self.device, "mean")
            if self.val_loader is not None and self.lr_scheduler is not None:
                self.lr_scheduler.step(current_metric)
            self._epoch += 1
            total = self.train_time + time.time() - t0
            if self.save_root is not None or self.name is not None:
                if current_metric < best_metric:
                    best_metric = current_metric
                    self._best_epoch = self._epoch
                    self.save_checkpoint("best", current_metric, best_metric, train_time=total)
                self.save_checkpoint("latest", current_metric, best_metric, train_time=total)
                if save_every_kth_epoch and self._epoch % save_every_kth_epoch == 0:
                    self.save_checkpoint(f"epoch-{self._epoch}", current_metric, best_metric, train_time=total)
        return best_metric

    def _train_epoch_impl(self, progress, forward_context, backprop):
        raise NotImplementedError

    def _validate_impl(self, forward_context):
        raise NotImplementedError
