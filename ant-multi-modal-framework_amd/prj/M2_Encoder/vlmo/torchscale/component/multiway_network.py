"""MultiwayNetwork: two copies (A = vision branch, B = text branch) of a module; the whole tensor is routed to one
of them (reference: prj/M2_Encoder/vlmo/torchscale/component/multiway_network.py:10-55; the split/cat form for fused
vision-language input is not used by ITC).  Here the copies are parameter holders: the fused layer reads
`.A` / `.B` directly, so no per-call module-tree walk (the reference re-applies set_split_position to the whole tree
on every layer call, encoder.py:122-124)."""
import copy

from torch import nn


def branch_of(split_position):
    if split_position == -1:
        return "A"
    if split_position == 0:
        return "B"
    raise NotImplementedError("fused vision-language input (split_position > 0) is outside the ITC path")


class MultiwayNetwork(nn.Module):
    def __init__(self, module, dim=1):
        super().__init__()
        self.dim = dim
        self.A = module
        self.B = copy.deepcopy(module)
        if hasattr(self.B, "reset_parameters"):
            self.B.reset_parameters()
        self.split_position = -1

    def pick(self, branch):
        return self.A if branch == "A" else self.B


class MutliwayEmbedding(MultiwayNetwork):
    def __init__(self, modules, dim=1):
        nn.Module.__init__(self)
        self.dim = dim
        assert len(modules) == 2
        self.A, self.B = modules
        self.split_position = -1


def MultiwayWrapper(args, module, dim=1):
    return MultiwayNetwork(module, dim=dim) if args.multiway else module
