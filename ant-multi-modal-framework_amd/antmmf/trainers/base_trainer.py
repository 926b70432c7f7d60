"""BaseTrainer: step orchestration around the contrastive hot path (reference: antmmf/trainers/base_trainer.py:112-139,
184-218,274-371,473-717 -- load -> wrap -> loop{forward, extract loss, backward, clip, step, meter}).

What is kept: the registry name ("base_trainer"), `Trainer(config)`, `.load()`, `.train()`, the `training_parameters.*`
keys the loop reads, `_forward_pass / _extract_loss / _backward / _update_meter`, and the one-process-per-GPU contract
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).  What is re-designed for MI355X:
  * data parallelism is NOT torch DDP: parameters live in a flat arena (antmmf.hip.arena) and the gradient reduction is a
    few large RCCL all-reduces over contiguous ranges of it, started from inside the backward pass as soon as the fused layers report
    a bucket's gradients final (`overlap_grad_allreduce`, default on; `grad_allreduce_dtype: bf16` halves the xGMI bytes), the 1/world
    mean and the clip coefficient folded into the fused AdamW launch -- no per-bucket copies, no find_unused_parameters graph walk;
  * bf16 compute with fp32 masters instead of fp16 autocast + GradScaler (no scaler, no unscale pass);
  * `current_iteration` advances ONCE per batch (the reference increments it twice, base_trainer.py:551,589, so its
    max_iterations / lr steps count half-steps; documented deviation);
  * the per-iteration reduce_dict of losses is skipped when the model already returns globally reduced losses;
  * meters stay ON THE DEVICE: losses and the gradient norm are accumulated as tensors and read (one sync) only at `log_interval`,
    evaluation and the end of training -- the reference's `float(v)` per loss per iteration and the `.item()` in clip_gradients
    are host syncs that stall kernel launch-ahead; the clip coefficient is computed on the device and handed to the fused AdamW as a tensor.
Also kept: the LR schedule (`lr_scheduler: true` -> LambdaLR over `lr_lambda_update`: warm-up, `lr_steps` / `lr_epochs`, `lr_ratio`;
a model's `get_custom_scheduler(trainer)` wins; reference base_trainer.py:445-464, 604-607, utils/general.py:27-44), gradient accumulation under
the reference key `gradient_accumulation_steps` (base.yml:174; `update_frequency` kept as an alias), evaluation every `evaluation_interval`
iterations with early stopping on `monitored_metric` (`should_early_stop`, `patience`, `metric_minimize`; base_trainer.py:473-530 `_logistics`),
the scheduled hard-mining ratio `incre_num` of the CN-VID configs (base_trainer.py:552-571) and the fp32 escape list
`amp_attributes.amp_escapes` (antmmf/utils/register_fp32.py:42-69).
Checkpointing (antmmf.common.checkpoint.Checkpoint, same files and keys as the reference's) is active when
`training_parameters.save_dir` / `resume_file` / `resume` is set: `load()` restores, `train()` snapshots every
`snapshot_interval` iterations and writes `<model>_final.pth` at the end (reference: base_trainer.py:184-218,373-397,555-607).
Data loading (task_loader), validation and early stopping are outside the step path: `load_task()` takes a user-supplied
iterable of SampleLists (tests / bench feed synthetic ones).
"""
import math
import os
import time
import warnings
from bisect import bisect

import torch
import torch.distributed as dist

from antmmf.common.registry import registry
from antmmf.models.build import build_model
from antmmf.optimizer import build_optimizer
from antmmf.structures.sample import SampleList
from antmmf.utils.distributed_utils import get_rank, get_world_size, is_main_process, reduce_dict, synchronize


class EarlyStopping:
    """Best-so-far bookkeeping of the monitored validation metric.  Attribute names are the reference's (antmmf/utils/early_stopping.py:8-102:
    `best_monitored_value`, `best_monitored_iteration`, `activated`, `init_from_checkpoint`) because Checkpoint.save / _load read and restore
    them by those names; the decision itself is taken in BaseTrainer._logistics on a value every rank agrees on."""

    def __init__(self, monitored_metric="total_loss", patience=30000, minimize=True, should_stop=False):
        self.monitored_metric, self.patience, self.minimize, self.should_stop = monitored_metric, patience, minimize, should_stop
        self.best_monitored_value = math.inf if minimize else -math.inf
        self.best_monitored_iteration = 0
        self.activated = False

    def improved(self, value):
        return value < self.best_monitored_value if self.minimize else value > self.best_monitored_value

    def update(self, iteration, value):
        """-> (is a new best, training should stop)."""
        if self.improved(value):
            self.best_monitored_value, self.best_monitored_iteration = value, iteration
            return True, False
        if iteration - self.best_monitored_iteration > self.patience:
            self.activated = True
            return False, bool(self.should_stop)
        return False, False

    def init_from_checkpoint(self, ckpt):
        if ckpt.get("best_iteration") is not None:
            self.best_monitored_iteration = ckpt["best_iteration"]
        if ckpt.get("best_metric_value") is not None:
            self.best_monitored_value = ckpt["best_metric_value"]

    def get_info(self):
        return {"best iteration": self.best_monitored_iteration, f"best {self.monitored_metric}": self.best_monitored_value}


@registry.register_trainer("base_trainer")
class BaseTrainer:
    def __init__(self, config, train_batches=None):
        self.config = config
        self.train_batches = train_batches
        self.profiler = {}
        self.current_iteration = 0
        self.current_epoch = 0
        self.meters = {}
        self._dev_meters = {}
        self.val_batches = None
        self.checkpoint = None

    # ------------------------------------------------------------------ load
    def load(self):
        self._init_process_group()
        self.load_model()
        self.load_optimizer()
        tp = self.config.training_parameters
        self.max_iterations = tp.get("max_iterations", math.inf)
        self.log_interval = tp.get("log_interval", 100)
        gas = tp.get("gradient_accumulation_steps", None)
        if gas is None:
            gas = tp.get("update_frequency", 1)      # this build's earlier name, kept as an alias
        self.gradient_accumulation_steps = int(gas)
        assert self.gradient_accumulation_steps >= 1
        self.should_clip_gradients = bool(tp.get("clip_gradients", False))
        self.max_grad_l2_norm = tp.get("max_grad_l2_norm", None)
        self.snapshot_interval = tp.get("snapshot_interval", None)
        self.evaluation_interval = tp.get("evaluation_interval", None)
        self.should_early_stop = bool(tp.get("should_early_stop", False))
        self.patience = tp.get("patience", 30000)
        self.monitored_metric = tp.get("monitored_metric", "total_loss")
        self.metric_minimize = bool(tp.get("metric_minimize", True))
        self.early_stopping = EarlyStopping(self.monitored_metric, self.patience, self.metric_minimize, self.should_early_stop)
        self.epoch_iterations = len(self.train_batches) if hasattr(self.train_batches, "__len__") else 0
        # ragged per-rank batches (the reference pads + trims on every gather, distributed_utils.py:131-160): with `pad_ragged_batches` the row-sharded
        # losses pad every rank to the per-rank batch size of the configuration (reference: batch_size is the GLOBAL size, utils/general.py get_batch_size).
        # SEPARATE maxima for training and evaluation (ceil: a global size that does not divide by the world leaves one more row on the first ranks): the training steps pad to
        # batch_size / W, `evaluate` switches to test_batch_size / W for its loop and back (round 5 took the larger of the two for the whole run: with a bigger
        # test_batch_size every TRAINING step padded its embeddings and slabs to the eval size).  A rank over the maximum fails on every rank together (contrastive._Rows)
        self._max_rows_train = self._max_rows_eval = None
        if tp.get("pad_ragged_batches", False) and tp.get("batch_size", None):
            from antmmf.hip import contrastive

            world = get_world_size()
            self._max_rows_train = max(1, -(-int(tp.batch_size) // world))
            self._max_rows_eval = max(1, -(-int(tp.get("test_batch_size", 0) or tp.batch_size) // world))
            contrastive.set_max_rows_per_rank(self._max_rows_train)
        self.setup_lr_scheduler()
        self.load_extras()

    def load_extras(self):
        """Checkpoint restore (reference: base_trainer.py:373-397).  Opt-in: without save_dir / resume* nothing touches the disk."""
        tp = self.config.training_parameters
        self.checkpoint = None
        if tp.get("save_dir", None) or tp.get("resume_file", None) or tp.get("resume", False):
            from antmmf.common.checkpoint import Checkpoint

            self.checkpoint = Checkpoint(self, load_only=not tp.get("save_dir", None))
            self.checkpoint.load_state_dict()
            synchronize()

    def _init_process_group(self):
        tp = self.config.training_parameters
        self.device = torch.device(tp.get("device", "cuda"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", tp.get("local_rank", 0) or 0))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            backend = "nccl" if self.device.type == "cuda" else "gloo"  # "nccl" is RCCL on ROCm
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        registry.register("current_device", self.device)
        seed = tp.get("seed", None)
        if seed is not None:
            torch.manual_seed(int(seed))

    def load_model(self):
        attrs = self.config.model_attributes
        assert len(attrs) == 1, "There should be only one model in model_attributes"
        key = list(attrs.keys())[0]
        mcfg = attrs[key]
        mcfg["model"] = key
        self.model = build_model(mcfg).to(self.device)
        if self.config.training_parameters.get("replace_speedup_op", False):
            from antmmf.utils.optim_utils import replace_speedup_op

            replace_speedup_op(self.model)
        escapes = self.config.get("amp_attributes", {}).get("amp_escapes", None) if hasattr(self.config, "get") else None
        if escapes:
            from antmmf.utils.register_fp32 import set_escapes_class_fp32

            set_escapes_class_fp32(self.model, escapes)
        if get_world_size() > 1:  # identical replicas: broadcast rank 0's initial weights
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)

    def load_optimizer(self):
        self.optimizer = build_optimizer(self.model, self.config)
        self.arena = getattr(self.optimizer, "arena", None)
        self.lr_scheduler = None

    def load_task(self, batches, val_batches=None):
        self.train_batches = batches
        self.val_batches = val_batches
        self.epoch_iterations = len(batches) if hasattr(batches, "__len__") else 0

    # ------------------------------------------------------------------ LR schedule
    def lr_lambda_update(self, i_iter):
        """Multiplier of every group's base lr at scheduler step i_iter (reference: antmmf/utils/general.py:27-44)."""
        tp = self.config.training_parameters
        if tp.get("use_warmup", False) and i_iter <= tp.get("warmup_iterations", 1000):
            alpha = float(i_iter) / float(tp.get("warmup_iterations", 1000))
            return tp.get("warmup_factor", 0.2) * (1.0 - alpha) + alpha
        steps = list(tp.get("lr_steps", []) or [])
        epochs = list(tp.get("lr_epochs", []) or [])
        if epochs:  # lr_epochs win over lr_steps
            steps = [self.epoch_iterations * e for e in epochs]
        return pow(tp.get("lr_ratio", 0.1), bisect(steps, i_iter))

    def setup_lr_scheduler(self):
        self.lr_scheduler = None
        target = self.model.module if hasattr(self.model, "module") and hasattr(self.model.module, "get_custom_scheduler") else self.model
        if hasattr(target, "get_custom_scheduler"):
            self.lr_scheduler = target.get_custom_scheduler(self)
        if self.lr_scheduler is None and self.config.training_parameters.get("lr_scheduler", False) is True:
            self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=self.lr_lambda_update)

    def _run_scheduler(self):
        if self.lr_scheduler is not None and self.current_iteration % self.gradient_accumulation_steps == 0:
            self.lr_scheduler.step()

    # ------------------------------------------------------------------ loop
    def train(self):
        self.model.train()
        self.optimizer.zero_grad()
        t0 = time.perf_counter()
        mcfg = self.config.model_attributes[list(self.config.model_attributes.keys())[0]]
        mining = bool(mcfg.get("hard_example_mining", False)) and mcfg.get("change_iter", None) is not None
        for batch in self.train_batches:
            if self.current_iteration >= self.max_iterations:
                break
            self.current_iteration += 1
            registry.register("current_iteration", self.current_iteration)
            if mining and batch:  # scheduled hard-negative ratio of the CN-VID configs (reference base_trainer.py:552-571)
                batch["incre_num"] = min(int(self.current_iteration / mcfg.change_iter) * mcfg.change_rate, 1.0)
            if self.train_step(batch) is None:
                continue
            if self.current_iteration % self.log_interval == 0:
                meters = self.read_meters()  # the one host sync of the interval
                if is_main_process():
                    dt = time.perf_counter() - t0
                    lr = self.optimizer.param_groups[0]["lr"]
                    print(f"iter {self.current_iteration}: " + ", ".join(f"{k}={v:.5f}" for k, v in meters.items()) + f", lr={lr:.3e} ({dt:.1f}s)", flush=True)
            if (self.checkpoint is not None and self.snapshot_interval and self.checkpoint.save_dir_enabled
                    and self.current_iteration % self.snapshot_interval == 0
                    and self.current_iteration % self.gradient_accumulation_steps == 0):
                self.checkpoint.save(self.current_iteration)
            if self._logistics():
                break
        synchronize()
        if self.checkpoint is not None and self.checkpoint.save_dir_enabled:
            if self.early_stopping.activated and self.should_early_stop:
                self.checkpoint.restore()   # <model>_final.pth holds the BEST weights after an early stop (reference early_stopping.py:79-83)
            self.checkpoint.finalize()
        return self.read_meters()

    # the pre-round-3 attribute names, kept readable
    @property
    def best_monitored(self):
        v = self.early_stopping.best_monitored_value
        return None if math.isinf(v) else v

    @property
    def best_iteration(self):
        return self.early_stopping.best_monitored_iteration

    def train_step(self, batch):
        """ONE iteration of the loop body: forward, meters, loss, backward (+ gradient all-reduce, clip, fused optimizer when the
        accumulation window closes), LR schedule.  bench.py times exactly this.  Returns the loss tensor (None for an empty batch)."""
        report, _, _ = self._forward_pass(batch)
        if report is None:
            return None
        self._update_meter(report)
        loss = self._extract_loss(report)
        self._backward(loss)
        self._run_scheduler()
        return loss

    # ------------------------------------------------------------------ evaluation / early stopping
    def _enter_eval_rows(self):
        """the ragged-batch padding maximum of the evaluation loader for the duration of an evaluation loop (see __init__)"""
        if getattr(self, "_max_rows_eval", None) is not None:
            from antmmf.hip import contrastive

            contrastive.set_max_rows_per_rank(self._max_rows_eval)

    def _leave_eval_rows(self):
        if getattr(self, "_max_rows_train", None) is not None:
            from antmmf.hip import contrastive

            contrastive.set_max_rows_per_rank(self._max_rows_train)

    def evaluate(self, batches):
        """Mean losses of `batches` in eval mode (no gradients, no optimizer); overridden by RetrievalTrainer for retrieval metrics."""
        was_training = self.model.training
        self.model.eval()
        sums, n = {}, 0
        self._enter_eval_rows()
        try:
            with torch.no_grad():
                for batch in batches:
                    report, _, _ = self._forward_pass(batch)
                    if report is None:
                        continue
                    n += 1
                    for k, v in report["losses"].items():
                        sums[k] = sums.get(k, 0) + v.detach().float().mean()
        finally:
            self._leave_eval_rows()
        if was_training:
            self.model.train()
        if sums and get_world_size() > 1 and not self.config.training_parameters.get("losses_are_global", True):
            # per-rank validation shards: the mean over ranks is the value every rank reports (one packed all-reduce)
            keys = sorted(sums)
            packed = torch.stack([sums[k].reshape(()) for k in keys] + [torch.tensor(float(n), device=sums[keys[0]].device)])
            dist.all_reduce(packed)
            n = float(packed[-1])
            sums = {k: packed[i] for i, k in enumerate(keys)}
        out = {k: float(v) / max(n, 1) for k, v in sums.items()}
        out["total_loss"] = sum(out.values())
        return out

    def _logistics(self):
        """Evaluation every `evaluation_interval` iterations + early stopping on `monitored_metric` (reference `_logistics` /
        EarlyStopping, base_trainer.py:473-530).  Returns True when training should stop."""
        if not self.evaluation_interval or getattr(self, "val_batches", None) is None:
            return False
        if self.current_iteration % self.evaluation_interval != 0:
            return False
        result = self.evaluate(self.val_batches)
        self.last_evaluation = result
        if is_main_process():
            print(f"iter {self.current_iteration} val: " + ", ".join(f"{k}={v:.5f}" for k, v in result.items() if isinstance(v, float)), flush=True)
        key = self.monitored_metric if self.monitored_metric in result else next((k for k in result if k.endswith(self.monitored_metric)), None)
        if key is None:
            return False
        value = float(result[key])
        if get_world_size() > 1:
            # ONE value for every rank (rank 0's): a rank that saw a different validation shard must not leave the loop alone --
            # the others would block in the next collective
            t = torch.tensor([value], dtype=torch.float64, device=self.device)
            dist.broadcast(t, src=0)
            value = float(t)
        better, stop = self.early_stopping.update(self.current_iteration, value)
        if better and self.checkpoint is not None and self.checkpoint.save_dir_enabled:
            self.checkpoint.save(self.current_iteration, update_best=True)
        return stop

    def _forward_pass(self, batch, enable_amp=False):
        if not batch:
            return None, None, None
        if (self.arena is not None and self.model.training and torch.is_grad_enabled() and get_world_size() > 1
                and self.current_iteration % self.gradient_accumulation_steps == 0
                and self.config.training_parameters.get("overlap_grad_allreduce", True)):
            rd = self.config.training_parameters.get("grad_allreduce_dtype", "fp32")
            self.arena.arm_overlap(reduce_dtype=torch.bfloat16 if str(rd) in ("bf16", "bfloat16") else None)
        prepared = batch.to(self.device) if isinstance(batch, SampleList) else SampleList(batch).to(self.device)
        model_output = self.model(prepared)
        return dict(losses=model_output["losses"], metrics=model_output.get("metrics", {}),
                    dataset_type=prepared.get("dataset_type", "train")), model_output, prepared

    def _extract_loss(self, report):
        return sum(l.mean() for l in report["losses"].values())

    def _backward(self, loss):
        (loss / self.gradient_accumulation_steps).backward()
        if self.current_iteration % self.gradient_accumulation_steps != 0:
            return
        world = 1
        if self.arena is not None:
            world = self.arena.allreduce_grads()
            scale = 1.0 / world
            if self.should_clip_gradients and self.max_grad_l2_norm:
                # clip coefficient min(1, max_norm / (norm + 1e-6)) stays on the device: it reaches the fused AdamW as a 1-element tensor
                norm = self.arena.grad_norm() * scale
                self._dev_meters["grad_norm"] = norm.detach().reshape(())
                scale = scale * torch.clamp(self.max_grad_l2_norm / (norm + 1e-6), max=1.0)
            self.optimizer.grad_scale = scale
        else:
            if get_world_size() > 1:
                for p in self.model.parameters():
                    if p.grad is not None:
                        dist.all_reduce(p.grad)
                        p.grad.div_(get_world_size())
            if self.should_clip_gradients and self.max_grad_l2_norm:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_l2_norm)
        self.optimizer.step()
        self.optimizer.zero_grad()

    def _update_meter(self, report, meter=None, sync=True):
        """Meters are device tensors (latest value per key); nothing is copied to the host here."""
        if not hasattr(self, "_dev_meters"):
            self._dev_meters = {}
        with torch.no_grad():
            losses = {k: v.detach().float().mean() for k, v in report["losses"].items()}
            if sync and get_world_size() > 1 and not self.config.training_parameters.get("losses_are_global", True):
                losses = reduce_dict(losses)
            total = None
            for k, v in losses.items():
                self._dev_meters[k] = v
                total = v if total is None else total + v
            if total is not None:
                self._dev_meters[f"{report['dataset_type']}/total_loss"] = total

    def read_meters(self):
        """ONE host sync: every meter of the last iteration as a Python float (also published to the registry, where the
        reference keeps `<dataset_type>/total_loss` and `grad_norm`)."""
        dm = getattr(self, "_dev_meters", {})
        if dm:
            keys = list(dm.keys())
            vals = torch.stack([dm[k].reshape(()).float() for k in keys]).tolist()
            for k, v in zip(keys, vals):
                self.meters[k] = v
                if k.endswith("/total_loss") or k == "grad_norm":
                    registry.register(k, v)
        return self.meters
