"""fp32 escape list (reference: antmmf/utils/register_fp32.py:27-69, switch `amp_attributes.amp_escapes`, call site
antmmf/trainers/base_trainer.py:285-289).

The reference trains under fp16 autocast and forces the classes named in `amp_escapes` to run in fp32 (inputs cast up, autocast off).
This build computes in bf16 with fp32 masters; the same seam forces a listed class to fp32 the same way: floating-point tensor
arguments are cast to fp32 before its forward and the result keeps fp32.  Modules that ARE the fused kernels (marked
`_antmmf_hip_native`: they take bf16 activations by contract and already accumulate / normalise in fp32 inside the kernel) cannot be
escaped and are reported."""
import functools
import warnings
import weakref

import torch


def tensor2dtype(x, dtype):
    if torch.is_tensor(x):
        return x.to(dtype) if torch.is_floating_point(x) else x
    if isinstance(x, (list, tuple)):
        return type(x)(tensor2dtype(v, dtype) for v in x)
    if isinstance(x, dict):
        return {k: tensor2dtype(v, dtype) for k, v in x.items()}
    return x


def _fp32_forward(forward, ref_module, *args, **kwargs):
    return forward(ref_module(), *tensor2dtype(args, torch.float32), **tensor2dtype(kwargs, torch.float32))


def customed_forward(module):
    module.forward = functools.partial(_fp32_forward, type(module).forward, weakref.ref(module))


def get_amp_escapes_name(amp_escapes):
    if isinstance(amp_escapes, (list, tuple)):
        return [str(n).strip() for n in amp_escapes if str(n).strip()]
    return [n.strip() for n in str(amp_escapes).split(",") if n.strip()]


def set_escapes_class_fp32(model, amp_escapes):
    """Returns the names of the module instances now running in fp32."""
    names = get_amp_escapes_name(amp_escapes)
    done, native = [], []
    for name, layer in model.named_modules():
        if type(layer).__name__ not in names:
            continue
        if getattr(layer, "_antmmf_hip_native", False):
            native.append(name)
            continue
        customed_forward(layer)
        done.append(name)
    if native:
        warnings.warn(f"amp_escapes: {native} are fused bf16 kernels (fp32 statistics / accumulation inside the kernel); not escaped")
    if not done and not native:
        warnings.warn(f"Can't find any class in model. config amp_escapes:{amp_escapes}")
    return done
