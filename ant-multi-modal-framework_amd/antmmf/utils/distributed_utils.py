"""Communication helpers of the contrastive step, RCCL-native (torch.distributed backend "nccl" IS RCCL on ROCm).

Same public names and meaning as the reference's antmmf/utils/distributed_utils.py:13-273 (get_rank,
get_world_size, is_main_process, synchronize, gather_tensor, all_gather, reduce_dict, broadcast_scalar),
re-designed for xGMI: tensor collectives are single `all_gather_into_tensor` / `reduce_scatter_tensor` calls on
contiguous buffers (the reference issues W+1 list-of-tensor collectives, a host sync for the size exchange and W
serial `reduce`s in backward).  (The multi-process CPU unit tests run it on "gloo" with a reduce_scatter shim installed by the tests.)
"""
import pickle

import torch
from torch import distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def broadcast_tensor(tensor, src=0):
    if get_world_size() > 1:
        with torch.no_grad():
            dist.broadcast(tensor, src=src)
    return tensor


def broadcast_scalar(scalar, src=0, device="cpu"):
    t = torch.tensor(scalar).to(device)
    return broadcast_tensor(t, src).item()


def reduce_tensor(tensor):
    world = get_world_size()
    if world < 2:
        return tensor
    with torch.no_grad():
        dist.reduce(tensor, dst=0)
        if dist.get_rank() == 0:
            tensor = tensor.div(world)
    return tensor


class GradientAllGather(torch.autograd.Function):
    """all-gather with autograd: forward = one all_gather_into_tensor, backward = one reduce_scatter_tensor(sum)
    -- the same arithmetic as the reference's all_gather + W x dist.reduce (distributed_utils.py:92-116): rank r
    receives the SUM over ranks of the gradient w.r.t. its shard."""

    @staticmethod
    def forward(ctx, tensor):
        world = dist.get_world_size()
        out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(out, tensor.contiguous())
        return out

    @staticmethod
    def backward(ctx, grad):
        world = dist.get_world_size()
        grad = grad.contiguous()
        out = torch.empty((grad.shape[0] // world,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
        dist.reduce_scatter_tensor(out, grad, op=dist.ReduceOp.SUM)
        return out


gradient_all_gather = GradientAllGather.apply


def gather_tensor(tensor, method="stack", back_gradient=False, pad_tensors=False):
    """Gather `tensor` from every rank; "cat" concatenates along dim 0, "stack" adds a leading world dim.
    `pad_tensors=True` supports ragged first dims (one extra int64 all-gather, as the reference does on EVERY call;
    here only when asked)."""
    world = get_world_size()
    if world < 2:
        return tensor
    if tensor.ndim == 0 and (method != "stack" or pad_tensors):
        raise ValueError("gather_tensor: 0-dim tensors need method='stack' and pad_tensors=False")
    if tensor.ndim == 0:
        with torch.no_grad():
            out = torch.empty(world, dtype=tensor.dtype, device=tensor.device)
            dist.all_gather_into_tensor(out, tensor.reshape(1))
        return out
    sizes = None
    if pad_tensors:
        with torch.no_grad():
            n = torch.tensor([tensor.shape[0]], dtype=torch.int64, device=tensor.device)
            all_n = torch.empty(world, dtype=torch.int64, device=tensor.device)
            dist.all_gather_into_tensor(all_n, n)
            sizes = all_n.tolist()
        if len(set(sizes)) == 1:
            sizes = None
        else:
            pad = max(sizes) - tensor.shape[0]
            if pad:
                tensor = torch.cat([tensor, tensor.new_zeros((pad,) + tuple(tensor.shape[1:]))], dim=0)
    if back_gradient:
        full = gradient_all_gather(tensor)
    else:
        with torch.no_grad():
            full = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
            dist.all_gather_into_tensor(full, tensor.contiguous())
    chunks = list(full.chunk(world, dim=0))
    if sizes is not None:
        chunks = [c[:s] for c, s in zip(chunks, sizes)]
        return torch.cat(chunks, dim=0) if method != "stack" else torch.stack(chunks, dim=0)
    return full if method != "stack" else torch.stack(chunks, dim=0)


def reduce_dict(dictionary):
    """Average a dict of scalar tensors onto rank 0 (one packed reduce)."""
    world = get_world_size()
    if world < 2:
        return dictionary
    with torch.no_grad():
        keys = sorted(dictionary.keys())
        values = torch.stack([dictionary[k].detach().float().reshape(()) for k in keys], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0:
            values /= world
        return {k: v for k, v in zip(keys, values)}


def all_gather(data):
    """Gather arbitrary picklable objects (hard-mining batch sizes, eval results)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, pickle.loads(pickle.dumps(data)))
    return out
