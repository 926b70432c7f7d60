"""TEST INFRASTRUCTURE: W ranks of the row-sharded losses simulated one after the other in ONE process (the GPU boxes have a single GPU).

The product functions (antmmf.hip.contrastive.*_sharded) talk to their peers through four seams -- `_world`, `_all_gather_packed`,
`_reduce_scatter_packed`, `_all_reduce_sum`.  `SimulatedRanks` replaces them while rank r's forward + backward run:
  * the gather returns the FULL tensors the test registered (what the W ranks would have contributed),
  * the loss all-reduce records rank r's local part (the caller sums the W parts),
  * the reduce-scatter adds rank r's [B_g, ...] gradient contributions into accumulators (the caller reads the sums: exactly what the W-rank
    reduce-scatter would deliver, rank by rank) and hands the local slice back to autograd.
Everything else -- similarity GEMMs, loss kernels, the four GEMMs of the backward -- is the product code on the product device."""
import contextlib

import torch


class SimulatedRanks:
    def __init__(self, world):
        self.world = world
        self.rank = 0
        self.full = None            # list of full [B_g, ...] tensors, in the order of the gather call
        self.grad_sums = None       # accumulators in the order of the reduce-scatter call
        self.loss_parts = []

    @contextlib.contextmanager
    def as_rank(self, rank, full):
        from antmmf.hip import contrastive as C

        self.rank, self.full = rank, [f.detach().float().contiguous() for f in full]
        saved = (C._world, C._single, C._all_gather_packed, C._reduce_scatter_packed, C._all_reduce_sum)
        sim = self

        def all_gather_packed(tensors, group):
            assert len(tensors) == len(sim.full)
            B = tensors[0].shape[0]
            for t, f in zip(tensors, sim.full):   # the local share must be what the test said this rank owns
                assert torch.equal(t.reshape(B, -1), f.reshape(f.shape[0], -1)[sim.rank * B:(sim.rank + 1) * B])
            return [f.reshape((f.shape[0],) + tuple(t.shape[1:])) for t, f in zip(tensors, sim.full)]

        def reduce_scatter_packed(fulls, group):
            if sim.grad_sums is None:
                sim.grad_sums = [torch.zeros_like(f) for f in fulls]
            outs = []
            for acc, f in zip(sim.grad_sums, fulls):
                acc += f
                B = f.shape[0] // sim.world
                outs.append(f[sim.rank * B:(sim.rank + 1) * B].contiguous())
            return outs

        def all_reduce_sum(t, group):
            sim.loss_parts.append(t.detach().clone())
            return t

        C._world = lambda group: (sim.world, sim.rank)
        C._single = lambda w: False
        C._all_gather_packed, C._reduce_scatter_packed, C._all_reduce_sum = all_gather_packed, reduce_scatter_packed, all_reduce_sum
        try:
            yield self
        finally:
            C._world, C._single, C._all_gather_packed, C._reduce_scatter_packed, C._all_reduce_sum = saved
