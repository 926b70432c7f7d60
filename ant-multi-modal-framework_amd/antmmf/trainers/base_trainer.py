"""BaseTrainer: step orchestration around the contrastive hot path (reference: antmmf/trainers/base_trainer.py:112-139,
184-218,274-371,473-717 -- load -> wrap -> loop{forward, extract loss, backward, clip, step, meter}).

What is kept: the registry name ("base_trainer"), `Trainer(config)`, `.load()`, `.train()`, the `training_parameters.*`
keys the loop reads, `_forward_pass / _extract_loss / _backward / _update_meter`, and the one-process-per-GPU contract
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher).  What is re-designed for MI355X:
  * data parallelism is NOT torch DDP: parameters live in a flat arena (antmmf.hip.arena) and the gradient reduction is a
    few large RCCL all-reduces over it after backward, the 1/world mean and the clip coefficient folded into the fused
    AdamW launch -- no per-bucket copies, no find_unused_parameters graph walk;
  * bf16 compute with fp32 masters instead of fp16 autocast + GradScaler (no scaler, no unscale pass);
  * `current_iteration` advances ONCE per batch (the reference increments it twice, base_trainer.py:551,589, so its
    max_iterations / lr steps count half-steps; documented deviation);
  * the per-iteration reduce_dict of losses is skipped when the model already returns globally reduced losses.
Checkpointing (antmmf.common.checkpoint.Checkpoint, same files and keys as the reference's) is active when
`training_parameters.save_dir` / `resume_file` / `resume` is set: `load()` restores, `train()` snapshots every
`snapshot_interval` iterations and writes `<model>_final.pth` at the end (reference: base_trainer.py:184-218,373-397,555-607).
Data loading (task_loader), validation and early stopping are outside the step path: `load_task()` takes a user-supplied
iterable of SampleLists (tests / bench feed synthetic ones).
"""
import math
import os
import time

import torch
import torch.distributed as dist

from antmmf.common.registry import registry
from antmmf.models.build import build_model
from antmmf.optimizer import build_optimizer
from antmmf.structures.sample import SampleList
from antmmf.utils.distributed_utils import get_rank, get_world_size, is_main_process, reduce_dict, synchronize


@registry.register_trainer("base_trainer")
class BaseTrainer:
    def __init__(self, config, train_batches=None):
        self.config = config
        self.train_batches = train_batches
        self.profiler = {}
        self.current_iteration = 0
        self.current_epoch = 0
        self.meters = {}
        self.checkpoint = None

    # ------------------------------------------------------------------ load
    def load(self):
        self._init_process_group()
        self.load_model()
        self.load_optimizer()
        tp = self.config.training_parameters
        self.max_iterations = tp.get("max_iterations", math.inf)
        self.log_interval = tp.get("log_interval", 100)
        self.gradient_accumulation_steps = max(1, int(tp.get("update_frequency", 1)))
        self.should_clip_gradients = bool(tp.get("clip_gradients", False))
        self.max_grad_l2_norm = tp.get("max_grad_l2_norm", None)
        self.snapshot_interval = tp.get("snapshot_interval", None)
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
        if get_world_size() > 1:  # identical replicas: broadcast rank 0's initial weights
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)

    def load_optimizer(self):
        self.optimizer = build_optimizer(self.model, self.config)
        self.arena = getattr(self.optimizer, "arena", None)
        self.lr_scheduler = None

    def load_task(self, batches):
        self.train_batches = batches

    # ------------------------------------------------------------------ loop
    def train(self):
        self.model.train()
        self.optimizer.zero_grad()
        t0 = time.perf_counter()
        for batch in self.train_batches:
            if self.current_iteration >= self.max_iterations:
                break
            self.current_iteration += 1
            report, model_output, _ = self._forward_pass(batch)
            if report is None:
                continue
            self._update_meter(report)
            loss = self._extract_loss(report)
            self._backward(loss)
            if self.current_iteration % self.log_interval == 0 and is_main_process():
                dt = time.perf_counter() - t0
                print(f"iter {self.current_iteration}: " + ", ".join(f"{k}={v:.5f}" for k, v in self.meters.items()) + f" ({dt:.1f}s)", flush=True)
            if (self.checkpoint is not None and self.snapshot_interval and self.checkpoint.save_dir_enabled
                    and self.current_iteration % self.snapshot_interval == 0
                    and self.current_iteration % self.gradient_accumulation_steps == 0):
                self.checkpoint.save(self.current_iteration)
        synchronize()
        if self.checkpoint is not None and self.checkpoint.save_dir_enabled:
            self.checkpoint.finalize()
        return self.meters

    def _forward_pass(self, batch, enable_amp=False):
        if not batch:
            return None, None, None
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
                norm = float(self.arena.grad_norm()) * scale
                if norm > self.max_grad_l2_norm:
                    scale *= self.max_grad_l2_norm / (norm + 1e-6)
                registry.register("grad_norm", norm)
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
        with torch.no_grad():
            losses = {k: v.detach().mean() for k, v in report["losses"].items()}
            if sync and get_world_size() > 1 and not self.config.training_parameters.get("losses_are_global", True):
                losses = reduce_dict(losses)
            total = 0.0
            for k, v in losses.items():
                self.meters[k] = float(v)
                total += float(v)
            self.meters[f"{report['dataset_type']}/total_loss"] = total
            registry.register(f"{report['dataset_type']}/total_loss", total)
