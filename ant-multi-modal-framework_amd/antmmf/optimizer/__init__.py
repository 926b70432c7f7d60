"""build_optimizer (reference: antmmf/optimizer/build.py + antmmf/optimizer/__init__.py): `optimizer_attributes.type`
names a torch.optim class or a registry optimizer; parameter groups come from the model's get_optimizer_parameters.
On a GPU, Adam / AdamW map onto the fused flat-arena AdamW (antmmf.hip.arena.HipAdamW)."""
import torch

from antmmf.common.registry import registry


def build_optimizer(model, config, use_hip_arena=None):
    attrs = config.optimizer_attributes
    name = attrs.type
    params = attrs.get("params", {})
    params = params.to_dict() if hasattr(params, "to_dict") else dict(params)
    target = model.module if hasattr(model, "module") and hasattr(model.module, "get_optimizer_parameters") else model
    groups = target.get_optimizer_parameters(config) if hasattr(target, "get_optimizer_parameters") else [{"params": list(model.parameters())}]
    groups = [g for g in groups if len(g["params"]) > 0]
    on_gpu = any(p.is_cuda for g in groups for p in g["params"])
    if use_hip_arena is None:
        use_hip_arena = on_gpu
    if name in ("Adam", "AdamW") and use_hip_arena:
        from antmmf.hip.arena import HipAdamW

        if name == "Adam":
            params.setdefault("weight_decay", 0.0)
        if "betas" in params:
            params["betas"] = tuple(params["betas"])
        return HipAdamW(groups, **params)
    cls = registry.get_optimizer_class(name) or getattr(torch.optim, name, None)
    if cls is None:
        raise ValueError(f"No optimizer found for type {name}")
    if "betas" in params:
        params["betas"] = tuple(params["betas"])
    return cls(groups, **params)
