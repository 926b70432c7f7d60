"""Checkpoint: save / resume / load-pretrained around the training step (the surface of the reference's antmmf/common/checkpoint.py:79-356 and
`ckpt_name_from_core_args`, antmmf/utils/general.py:71-82).  SURVEY.md section 8(f1): released AntMMF checkpoints and this build's own checkpoints
are interchangeable because module / parameter names map 1:1.

What is kept from the reference is the CONTRACT -- class name, entry points, file layout, dictionary keys, `training_parameters` keys:
    <save_dir>/<tasks>_<models>[_<seed>]/config.yaml
    <save_dir>/<...>/models/model_<iteration>.ckpt        {"model", "optimizer", "current_iteration", "current_epoch",
    <save_dir>/<...>/<prefix>best.ckpt                      "best_iteration", "best_metric_value"}
    <save_dir>/<...>/<prefix><models>_final.pth            bare state_dict
    save_dir, seed, resume, resume_file, restart, load_pretrained, pretrained_mapping, max_ckpt_num
and the loader's tolerance: a bare state_dict or {"model": ...}; `module.` prefixes of a (Distributed)DataParallel run are dropped (this build never
wraps the model: data parallelism is the flat-arena all-reduce); `fa_history` is the old name of `fa_context`; a tensor the model does not have, or has in
another shape, is skipped with a warning; `pretrained_mapping` {source sub-tree: target sub-tree} additionally copies a sub-tree onto a differently
named module.

How it is done here: a weight file becomes a `_WeightPlan` -- (1) its keys go through a small rename pipeline, (2) ONE pass over the renamed keys builds
the list of (destination tensor, source tensor) copies against an index of the model's state_dict, sub-tree mappings being resolved by prefix lookup in
that index instead of scanning every (checkpoint key, model key) pair, (3) the copies run under no_grad, (4) the skipped keys are reported once.

MI355X specifics: parameters are fp32 views into the optimizer's flat arena with a bf16 compute shadow, so after any weight load the shadow is rebuilt
in one cast launch (`arena.sync_shadow()`); the optimizer state is the arena-shaped exp_avg / exp_avg_sq pair (HipAdamW.state_dict), saved and restored
as two flat tensors instead of per-parameter dicts.
"""
import os
import warnings

import torch

from antmmf.common.registry import registry
from antmmf.utils.distributed_utils import is_main_process, synchronize

_RENAMES = (
    lambda k: k[len("module."):] if k.startswith("module.") else k,     # written through a DataParallel / DDP wrapper
    lambda k: k.replace("fa_history", "fa_context"),                    # the attention-context module's earlier name
)


def ckpt_name_from_core_args(config):
    """`<tasks>_<models>[_<seed>]`: the run's folder under save_dir (reference: antmmf/utils/general.py:71-82)."""
    parts = ["-".join((config.get("task_attributes", None) or {}).keys()), "-".join(config.model_attributes.keys())]
    seed = config.training_parameters.get("seed", None)
    if seed is not None:
        parts.append("%d" % int(seed))
    return "_".join(parts)


def load_state_dict_mapping(model, ckpt_model, attr_mapping):
    """model.state_dict()[target] <- ckpt_model[source] for every {target: source} of `attr_mapping` (same helper name as the reference's, :16-47)."""
    own = model.state_dict()
    with torch.no_grad():
        for target, source in attr_mapping.items():
            own[target].copy_(ckpt_model[source])


class _WeightPlan:
    """The copies one weight file implies for one model (see the module docstring)."""

    def __init__(self, model, weights, subtree_map=None):
        self.own = model.state_dict()
        self.copies = []            # (destination name, source tensor)
        self.unknown, self.misshapen = [], []
        renamed = {}
        for key, value in weights.items():
            for fn in _RENAMES:
                key = fn(key)
            renamed[key] = value.data if isinstance(value, torch.nn.Parameter) else value
        for key, value in renamed.items():
            self._plan(key, value, report=True)
        # sub-tree mappings: "<source>.<rest>" in the file also lands on "<target>.<rest>" of the model -- wherever the sub-tree sits in the key
        # ("<parent>.<source>.<rest>" -> "<parent>.<target>.<rest>": nested encoders), as in the reference; absent targets are not an error here
        for source, target in (subtree_map or {}).items():
            needle = source + "."
            for key, value in renamed.items():
                at = key.find(needle)
                while at >= 0:
                    if at == 0 or key[at - 1] == ".":
                        self._plan(key[:at] + target + "." + key[at + len(needle):], value, report=False)
                    at = key.find(needle, at + 1)

    def _plan(self, name, value, report):
        dest = self.own.get(name)
        if dest is None:
            if report:
                self.unknown.append(name)
        elif tuple(dest.shape) != tuple(value.shape):
            if report:
                self.misshapen.append((name, tuple(value.shape), tuple(dest.shape)))
        else:
            self.copies.append((name, value))

    def apply(self):
        with torch.no_grad():
            for name, value in self.copies:
                self.own[name].copy_(value)
        for name in self.unknown:
            warnings.warn(f"checkpoint tensor '{name}' has no counterpart in the model: not loaded")
        for name, got, want in self.misshapen:
            warnings.warn(f"checkpoint tensor '{name}' is {got}, the model's is {want}: not loaded")
        return len(self.copies)


class Checkpoint:
    def __init__(self, trainer, load_only=False):
        self.trainer = trainer
        self.config = trainer.config
        tp = self.config.training_parameters
        self.save_dir = tp.get("save_dir", None) or "./save"
        self.save_dir_enabled = bool(tp.get("save_dir", None)) and not load_only
        self.max_ckpt_num = tp.get("max_ckpt_num", None)
        self.device = registry.get("current_device")
        self.model_name = "-".join(self.config.model_attributes.keys())
        self.ckpt_prefix = trainer.model.get_ckpt_name() + "_" if hasattr(trainer.model, "get_ckpt_name") else ""
        run = ckpt_name_from_core_args(self.config)
        _set_unfrozen(self.config, "log_foldername", run)
        self.ckpt_foldername = os.path.join(self.save_dir, run)
        self.models_foldername = os.path.join(self.ckpt_foldername, "models")
        self.pth_filepath = os.path.join(self.ckpt_foldername, self.ckpt_prefix + self.model_name + "_final.pth")
        if not load_only and is_main_process():
            os.makedirs(self.models_foldername, exist_ok=True)
            self.save_config()

    # ------------------------------------------------------------------ paths / small helpers
    @property
    def best_filepath(self):
        return os.path.join(self.ckpt_foldername, self.ckpt_prefix + "best.ckpt")

    def _snapshot_path(self, iteration):
        return os.path.join(self.models_foldername, "model_%d.ckpt" % iteration)

    def _log(self, msg):
        writer = getattr(self.trainer, "writer", None)
        if writer is not None:
            writer.write(msg)

    def save_config(self):
        with open(os.path.join(self.ckpt_foldername, "config.yaml"), "w") as f:
            f.write(str(self.config))

    def _read(self, file):
        on_gpu = self.device is not None and "cuda" in str(self.device)
        return torch.load(file, map_location=self.device if on_gpu else "cpu", weights_only=False)

    def _after_weight_load(self):
        """The loaded tensors went into the fp32 masters: rebuild the bf16 compute shadow (and invalidate cached packings)."""
        arena = getattr(self.trainer, "arena", None) or getattr(getattr(self.trainer, "optimizer", None), "arena", None)
        if arena is not None:
            arena.sync_shadow()
        else:
            from antmmf.hip.functional import bump_weight_version

            bump_weight_version()
        # modules that cache HOST copies of a parameter / buffer (e.g. DMAE's TokenImportanceSelector threshold: one device read per load, not per step) drop them
        # here: weights arrive through state_dict()[name].copy_(), which never runs a module's _load_from_state_dict
        model = getattr(self.trainer, "model", None)
        for m in (model.modules() if model is not None else ()):
            hook = getattr(m, "on_weights_loaded", None)
            if callable(hook):
                hook()

    # ------------------------------------------------------------------ load
    def load_state_dict(self):
        """Trainer.load() calls this once: an explicit `resume_file` wins (with the training state unless `restart`), otherwise `resume: true` picks up
        <prefix>best.ckpt of this run's folder if there is one (reference :136-162)."""
        tp = self.config.training_parameters
        explicit = tp.get("resume_file", None)
        if explicit is not None:
            if not os.path.exists(explicit):
                raise RuntimeError(f"resume_file {explicit} does not exist")
            self._load(explicit, resume_state=not tp.get("restart", False))
        elif tp.get("resume", False) is True:
            if os.path.exists(self.best_filepath):
                self._load(self.best_filepath)
            else:
                warnings.warn(f"resume requested but {self.best_filepath} is not there: starting from the initial weights")

    def load_model_weights(self, file, force=False):
        """Weights of `file` -> model (+ `pretrained_mapping` sub-trees when `load_pretrained` is on and `force` is off).  Returns the file's dictionary
        in the {"model": ...} form."""
        self._log(f"loading weights from {file}")
        blob = self._read(file)
        if "model" not in blob:
            blob = {"model": blob}
        tp = self.config.training_parameters
        use_map = bool(tp.get("load_pretrained", False)) and force is not True
        plan = _WeightPlan(self.trainer.model, blob["model"], (tp.get("pretrained_mapping", None) or {}) if use_map else None)
        n = plan.apply()
        self._log(f"{n} tensors loaded, {len(plan.unknown) + len(plan.misshapen)} skipped")
        self._after_weight_load()
        return blob

    def _load(self, file, force=False, resume_state=False):
        blob = self.load_model_weights(file, force=force)
        if not resume_state:
            return
        if "optimizer" in blob:
            self.trainer.optimizer.load_state_dict(blob["optimizer"])
        else:
            warnings.warn(f"{file} carries no optimizer state: the optimizer starts fresh")
        stopper = getattr(self.trainer, "early_stopping", None)
        if stopper is not None:
            stopper.init_from_checkpoint(blob)
        for key in ("current_iteration", "current_epoch"):
            if key in blob:
                setattr(self.trainer, key, blob[key])
                registry.register(key, blob[key])
        self._log(f"training state of {file} restored")

    # ------------------------------------------------------------------ save
    def remove_redundant_ckpts(self):
        """keep the newest `max_ckpt_num` snapshots"""
        if self.max_ckpt_num is None:
            return
        snaps = [os.path.join(self.models_foldername, f) for f in os.listdir(self.models_foldername) if f.startswith("model_") and f.endswith(".ckpt")]
        snaps.sort(key=os.path.getmtime, reverse=True)
        for stale in snaps[int(self.max_ckpt_num):]:
            os.remove(stale)

    def save(self, iteration, update_best=False):
        if not is_main_process():  # replicas are identical: rank 0 writes
            return
        model = self.trainer.model
        if isinstance(getattr(model, "module", None), torch.nn.Module):
            model = model.module
        stopper = getattr(self.trainer, "early_stopping", None)
        blob = dict(model=model.state_dict(), optimizer=self.trainer.optimizer.state_dict(),
                    current_iteration=self.trainer.current_iteration, current_epoch=self.trainer.current_epoch,
                    best_iteration=getattr(stopper, "best_monitored_iteration", 0), best_metric_value=getattr(stopper, "best_monitored_value", None))
        os.makedirs(self.models_foldername, exist_ok=True)
        torch.save(blob, self._snapshot_path(iteration))
        self.remove_redundant_ckpts()
        if update_best:
            torch.save(blob, self.best_filepath)

    def restore(self, with_sync=True):
        """back to the best weights seen (early stop / end of training), if a best.ckpt was written"""
        if with_sync:
            synchronize()
        if os.path.exists(self.best_filepath):
            self._log("restoring the best checkpoint")
            self._load(self.best_filepath, force=True)

    def finalize(self):
        if is_main_process():
            os.makedirs(self.ckpt_foldername, exist_ok=True)
            torch.save(self.trainer.model.state_dict(), self.pth_filepath)


def _set_unfrozen(config, key, value):
    frozen = getattr(config, "_frozen", False)
    if frozen and hasattr(config, "defrost"):
        config.defrost()
    config[key] = value
    if frozen and hasattr(config, "freeze"):
        config.freeze()
