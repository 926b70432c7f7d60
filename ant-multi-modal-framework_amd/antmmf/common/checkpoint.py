"""Checkpoint: save / resume / load-pretrained around the training step (reference: antmmf/common/checkpoint.py:79-356,
`ckpt_name_from_core_args` antmmf/utils/general.py:71-82).  SURVEY.md section 8(f1): released AntMMF checkpoints and this
build's own checkpoints are interchangeable because module / parameter names map 1:1.

Same surface and file layout as the reference:
    <save_dir>/<tasks>_<models>[_<seed>]/config.yaml
    <save_dir>/<...>/models/model_<iteration>.ckpt        {"model", "optimizer", "current_iteration", "current_epoch",
    <save_dir>/<...>/<prefix>best.ckpt                      "best_iteration", "best_metric_value"}
    <save_dir>/<...>/<prefix><models>_final.pth            bare state_dict
`training_parameters` keys read: save_dir, seed, resume, resume_file, restart, load_pretrained, pretrained_mapping,
max_ckpt_num.  Semantics kept from the reference's loader: a bare state_dict or {"model": ...}; `module.` prefixes written
by a DataParallel / DDP run are stripped (this build never wraps the model: data parallelism is the flat-arena all-reduce);
`fa_history` -> `fa_context`; tensors that are missing from the model or whose shape differs are skipped with a warning;
`pretrained_mapping` copies sub-trees across differently named modules.

MI355X specifics: parameters are fp32 views into the optimizer's flat arena with a bf16 compute shadow, so after any weight
load the shadow is rebuilt in one cast launch (`arena.sync_shadow()`); the optimizer state is the arena-shaped
exp_avg / exp_avg_sq pair (HipAdamW.state_dict), saved and restored as two flat tensors instead of per-parameter dicts.
"""
import glob
import os
import warnings

import torch

from antmmf.common.registry import registry
from antmmf.utils.distributed_utils import is_main_process, synchronize


def ckpt_name_from_core_args(config):
    """`<tasks>_<models>[_<seed>]` (reference: antmmf/utils/general.py:71-82)."""
    tasks = config.get("task_attributes", None) or {}
    name = "{}_{}".format("-".join(tasks.keys()), "-".join(config.model_attributes.keys()))
    seed = config.training_parameters.get("seed", None)
    if seed is not None:
        name += "_{:d}".format(int(seed))
    return name


def load_state_dict_mapping(model, ckpt_model, attr_mapping):
    """own_state[target_key] <- ckpt_model[source_key] for every pair of `attr_mapping` (reference :16-47)."""
    own_state = model.state_dict()
    for key in attr_mapping:
        own_state[key].copy_(ckpt_model[attr_mapping[key]])


class Checkpoint:
    def __init__(self, trainer, load_only=False):
        self.trainer = trainer
        self.config = trainer.config
        tp = self.config.training_parameters
        self.save_dir_enabled = bool(tp.get("save_dir", None)) and not load_only
        self.save_dir = tp.get("save_dir", None) or "./save"
        self.model_name = "-".join(self.config.model_attributes.keys())
        self.ckpt_foldername = ckpt_name_from_core_args(self.config)
        self.device = registry.get("current_device")
        self.ckpt_prefix = ""
        if hasattr(trainer.model, "get_ckpt_name"):
            self.ckpt_prefix = trainer.model.get_ckpt_name() + "_"
        _set_unfrozen(self.config, "log_foldername", self.ckpt_foldername)
        self.ckpt_foldername = os.path.join(self.save_dir, self.ckpt_foldername)
        self.pth_filepath = os.path.join(self.ckpt_foldername, self.ckpt_prefix + self.model_name + "_final.pth")
        self.models_foldername = os.path.join(self.ckpt_foldername, "models")
        if not load_only and is_main_process():
            os.makedirs(self.models_foldername, exist_ok=True)
            self.save_config()
        self.max_ckpt_num = tp.get("max_ckpt_num", None)

    # ------------------------------------------------------------------ small helpers
    def _write(self, msg):
        writer = getattr(self.trainer, "writer", None)
        if writer is not None:
            writer.write(msg)

    def save_config(self):
        with open(os.path.join(self.ckpt_foldername, "config.yaml"), "w") as f:
            f.write(str(self.config))

    def _torch_load(self, file):
        if self.device is not None and "cuda" in str(self.device):
            return torch.load(file, map_location=self.device, weights_only=False)
        return torch.load(file, map_location="cpu", weights_only=False)

    def _after_weight_load(self):
        """The loaded tensors went into the fp32 masters: rebuild the bf16 compute shadow (and invalidate cached packings)."""
        arena = getattr(self.trainer, "arena", None) or getattr(getattr(self.trainer, "optimizer", None), "arena", None)
        if arena is not None:
            arena.sync_shadow()
        else:
            from antmmf.hip.functional import bump_weight_version

            bump_weight_version()

    # ------------------------------------------------------------------ load
    def load_state_dict(self):
        """Called once by Trainer.load(): resume_file > <prefix>best.ckpt when `resume` (reference :136-162)."""
        tp = self.config.training_parameters
        resume_file = tp.get("resume_file", None)
        if resume_file is not None:
            if not os.path.exists(resume_file):
                raise RuntimeError("{} doesn't exist".format(resume_file))
            self._load(resume_file, resume_state=not tp.get("restart", False))
            return
        best = os.path.join(self.ckpt_foldername, self.ckpt_prefix + "best.ckpt")
        if tp.get("resume", False) is True:
            if os.path.exists(best):
                self._load(best)
            else:
                warnings.warn("Tried to resume but checkpoint filepath {} is not present. Skipping.".format(best))

    def load_model_weights(self, file, force=False):
        self._write("Loading checkpoint")
        ckpt = self._torch_load(file)
        if "model" in ckpt:
            ckpt_model = ckpt["model"]
        else:
            ckpt_model, ckpt = ckpt, {"model": ckpt}
        new_dict = {}
        for attr, value in ckpt_model.items():
            if "fa_history" in attr:
                new_dict[attr.replace("fa_history", "fa_context")] = value
            elif attr.startswith("module."):  # written by a (Distributed)DataParallel-wrapped run
                new_dict[attr.replace("module.", "", 1)] = value
            else:
                new_dict[attr] = value
        self._load_state_dict(new_dict)
        self._load_model_weights_with_mapping(new_dict, force=force)
        self._after_weight_load()
        return ckpt

    def _load_state_dict(self, state_dict):
        own_state = self.trainer.model.state_dict()
        with torch.no_grad():
            for name, param in state_dict.items():
                if name not in own_state:
                    warnings.warn(f"loading checkpoint warning: skip loading tensor:{name} in checkpoint, which does not exist in model")
                    continue
                if isinstance(param, torch.nn.Parameter):
                    param = param.data
                if own_state[name].shape != param.shape:
                    warnings.warn(f"loading checkpoint warning: skip loading tensor:{name} in checkpoint, whose shape does not "
                                  f"match model's tensor:{name}")
                    continue
                own_state[name].copy_(param)

    def _load_model_weights_with_mapping(self, weight_dict, force):
        tp = self.config.training_parameters
        mapping = tp.get("pretrained_mapping", None) or {}
        if not tp.get("load_pretrained", False) or force is True:
            mapping = {}
        if len(mapping) == 0:
            return
        own_state = self.trainer.model.state_dict()
        with torch.no_grad():
            for key, value in mapping.items():
                key, value = key + ".", value + "."
                for attr in weight_dict:
                    for own_attr in own_state:
                        if key in attr and value in own_attr and attr.replace(key, "") == own_attr.replace(value, ""):
                            self._write("Copying " + attr + " " + own_attr)
                            own_state[own_attr].copy_(weight_dict[attr])
        self._write("Pretrained model loaded")

    def _load(self, file, force=False, resume_state=False):
        ckpt = self.load_model_weights(file, force=force)
        if resume_state is False:
            return
        if "optimizer" in ckpt:
            self.trainer.optimizer.load_state_dict(ckpt["optimizer"])
        else:
            warnings.warn("'optimizer' key is not present in the checkpoint asked to be loaded. Skipping.")
        early_stopping = getattr(self.trainer, "early_stopping", None)
        if early_stopping is not None:
            early_stopping.init_from_checkpoint(ckpt)
        self._write("Checkpoint {} loaded".format(file))
        if "current_iteration" in ckpt:
            self.trainer.current_iteration = ckpt["current_iteration"]
            registry.register("current_iteration", self.trainer.current_iteration)
        if "current_epoch" in ckpt:
            self.trainer.current_epoch = ckpt["current_epoch"]
            registry.register("current_epoch", self.trainer.current_epoch)

    # ------------------------------------------------------------------ save
    def remove_redundant_ckpts(self):
        ckpts = glob.glob(os.path.join(self.models_foldername, "model_*.ckpt"))
        if self.max_ckpt_num is not None and len(ckpts) > self.max_ckpt_num:
            ckpts = sorted(ckpts, key=os.path.getmtime)
            for c in ckpts[: len(ckpts) - self.max_ckpt_num]:
                os.remove(c)

    def save(self, iteration, update_best=False):
        if not is_main_process():  # replicas are identical: rank 0 writes
            return
        early_stopping = getattr(self.trainer, "early_stopping", None)
        model = self.trainer.model
        model = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
        ckpt = {
            "model": model.state_dict(),
            "optimizer": self.trainer.optimizer.state_dict(),
            "current_iteration": self.trainer.current_iteration,
            "current_epoch": self.trainer.current_epoch,
            "best_iteration": getattr(early_stopping, "best_monitored_iteration", 0),
            "best_metric_value": getattr(early_stopping, "best_monitored_value", None),
        }
        os.makedirs(self.models_foldername, exist_ok=True)
        torch.save(ckpt, os.path.join(self.models_foldername, "model_%d.ckpt" % iteration))
        self.remove_redundant_ckpts()
        if update_best:
            torch.save(ckpt, os.path.join(self.ckpt_foldername, self.ckpt_prefix + "best.ckpt"))

    def restore(self, with_sync=True):
        if with_sync:
            synchronize()
        self._write("Restoring checkpoint")
        best = os.path.join(self.ckpt_foldername, self.ckpt_prefix + "best.ckpt")
        if os.path.exists(best):
            self._load(best, force=True)

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
