"""Two (or more) RCCL ranks vs one rank on the whole tiny-M2 ITC training step -- the hardware twin of
tests/test_host_logic.py::test_m2_step_two_ranks_equals_single_rank (which runs the same step on gloo + the lane emulator).

Launched by tests/test_e2e_gpu.py::test_m2_step_two_rccl_ranks_equals_single_rank when the box has >= 2 GPUs:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dp_rccl_case.py [overlap|plain|bf16]

Every rank first runs the FULL batch alone (no process group: world-1 shortcuts), then its shard of the same batch inside the job:
same global loss on every rank, the averaged arena gradient equals the single-rank gradient, replicas bit-identical after the fused AdamW.
Reference semantics being checked: GradientAllGather's x W backward + DDP's 1 / W mean (antmmf/utils/distributed_utils.py:92-189,
antmmf/trainers/base_trainer.py:351-371)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ant-multi-modal-framework_amd")
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), PKG, os.path.join(PKG, "prj", "M2_Encoder"), ROOT]


def one_step(dev, rows, mode, in_job):
    import model_cases as mc
    import weightgen as W
    from antmmf.hip.arena import HipAdamW

    model = mc.build_tiny_m2(dev)
    opt = HipAdamW([{"params": list(model.parameters())}], lr=1e-2, weight_decay=0.01)
    total = 8
    img = (W.data_tensor("m2dp.image8", (total, 3, 32, 32)) * 0.25 + 0.5).clamp(0, 1).to(dev)
    ids = W.data_ints("m2dp.ids8", (total, 12), 1, 300).to(dev)
    lengths = torch.tensor([12, 5, 8, 3, 12, 7, 4, 9], device=dev)
    mask = (torch.arange(12, device=dev)[None, :] < lengths[:, None]).long()
    ids = ids * mask
    if in_job and mode != "plain":
        assert opt.arena.arm_overlap(bucket_bytes=64 << 10, reduce_dtype=torch.bfloat16 if mode == "bf16" else None)
    out = model({"image": [img[rows]], "text_ids": ids[rows], "text_masks": mask[rows]})
    loss = out["losses"]["itc_loss"] + out["losses"]["itc_vl_loss"]
    loss.backward()
    w = opt.arena.allreduce_grads() if in_job else 1
    grad = opt.arena.grad.clone() / w
    opt.grad_scale = 1.0 / w
    opt.step()
    torch.cuda.synchronize()
    return float(loss), grad, opt.arena.master.clone(), w, getattr(opt.arena, "overlapped_buckets", 0)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    assert 8 % world == 0
    l1, g1, m1, w1, _ = one_step(dev, slice(0, 8), mode, in_job=False)
    assert w1 == 1
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    per = 8 // world
    lw, gw, mw, ww, early = one_step(dev, slice(rank * per, (rank + 1) * per), mode, in_job=True)
    assert ww == world, (ww, world)
    # every rank reports the GLOBAL loss; equal to the one-rank loss on the concatenated batch
    t = torch.tensor([lw], dtype=torch.float64, device=dev)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float(hi - lo) < 1e-6, (float(lo), float(hi))
    assert abs(lw - l1) <= 2e-4 * abs(l1), (lw, l1)
    tol = dict(rtol=3e-2, atol=3e-3 * float(g1.abs().max())) if mode == "bf16" else dict(rtol=2e-2, atol=2e-3 * float(g1.abs().max()))
    torch.testing.assert_close(gw, g1, **tol)   # (bf16 activations: the 2-rank shards round differently from the full batch in a few layers)
    cosine = float(torch.dot(gw, g1) / (gw.norm() * g1.norm()))
    assert cosine >= (0.999 if mode != "bf16" else 0.995), cosine
    # replicas: bit-identical masters on every rank
    ref = mw.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, mw), "replicas diverged after the optimizer step"
    if mode == "overlap":
        assert early >= 1, early
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"okdp world={world} mode={mode} loss={lw:.6f} one_rank={l1:.6f} cos={cosine:.6f} early_buckets={early}", flush=True)


if __name__ == "__main__":
    main()
