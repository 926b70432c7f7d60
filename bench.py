#!/usr/bin/env python
"""bench.py -- image-text pairs/s of the M2-Encoder ViT-L/14 ITC training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = forward + backward + optimizer over one synthetic batch already resident in HBM: B = 1024 pairs per GPU
(224x224x3 frames, 77-token captions), M2 `large` towers with patch 14 (257 visual tokens, 21+3 layers, d = 1024; SURVEY.md
"Facts" 3), random-init weights, bf16 MFMA compute with fp32 masters, global negatives all-gathered over RCCL, row-sharded
symmetric InfoNCE on both ITC levels, bucketed RCCL all-reduce of the flat gradient arena, fused AdamW.  Weak scaling: the
per-GPU batch is fixed, so the global batch is 1024 x N (8192 at N = 8, the configuration the metric is quoted on).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the dominant kernel (the bf16 MFMA GEMM): algorithmic flops per launch / average launch duration, measured live
                with HIP events on the launch stream over the timed steps; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/, plain torch fp32 restatement of the reference; kind "port") timed on this host's cores
                on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd", "prj", "M2_Encoder"))

TRAIN_GFLOP_PER_PAIR = {"l14": 627.3, "b16": 145.3}  # BASELINE.md section 4 (algorithmic, train = 3 x forward, recompute not counted)
PEAK_TFLOPS = 2500.0

WORKLOADS = {
    # M2 `large`, patch 14: 21 + 3 layers, d = 1024, 16 heads, 257 image tokens, 77 text tokens, vocab 115244, D = 1024
    "l14": dict(beit_version="large", encoder_embed_dim=1024, out_embed_dim=1024, encoder_layers=21, beit3_vl_layers=3,
                image_size=224, patch_size=14, vocab_size=115244, max_text_len=77),
    # config[1]: M2 `base`, patch 16 (ViT-B/16 dims + 12-layer text stack)
    "b16": dict(beit_version="base", encoder_embed_dim=768, out_embed_dim=768, encoder_layers=9, beit3_vl_layers=3,
                image_size=224, patch_size=16, vocab_size=64010, max_text_len=77),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="l14", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1024, help="pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recompute-ffn-norm", action="store_true", help="do not keep ffn_layernorm(gelu(u)) for backward (saves ~65 GiB at 1024 pairs/GPU, costs ~3 %%)")
    ap.add_argument("--gemm-table", default=None, help="write per-shape GEMM timing (from the live HIP-event trace) to this file")
    ap.add_argument("--cpu-sample", type=int, default=8, help="pairs in the CPU-oracle sample (8 pairs ~ 10 s on 32 host threads)")
    return ap.parse_args()


def synthetic_batch(cfg, batch, device, rank):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    img = torch.rand(batch, 3, cfg["image_size"], cfg["image_size"], generator=g, device=device)
    seq = cfg["max_text_len"]
    ids = torch.randint(1, cfg["vocab_size"], (batch, seq), generator=g, device=device)
    lengths = torch.randint(8, seq + 1, (batch,), generator=g, device=device)
    mask = (torch.arange(seq, device=device)[None, :] < lengths[:, None]).long()
    ids = ids * mask
    return {"image": [img], "text_ids": ids, "text_masks": mask}


def _cpu_baseline_worker(cfg, pairs, q):
    """The CPU oracle's M2 ITC step (fwd + bwd, fp32) on `pairs` pairs of the same shapes; puts pairs/s on `q`."""
    from oracle import step as ostep
    from oracle.shapes import m2_shapes

    ncpu = os.cpu_count() or 1
    # torch's CPU kernels stop scaling (and on a 256-core host collapse) well before one thread per core for this
    # workload's GEMM shapes ([2*257, 1024] x [1024, 4096]); 32 threads is used, or every core on a smaller host
    cores = min(ncpu, 32)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    shapes = m2_shapes(d=cfg["encoder_embed_dim"], layers=cfg["encoder_layers"], vl_layers=cfg["beit3_vl_layers"],
                       patch=cfg["patch_size"], res=cfg["image_size"], vocab=cfg["vocab_size"], out=cfg["out_embed_dim"])
    P = {}
    for k, s in shapes.items():
        if len(s) == 0:
            t = torch.tensor(2.659)
        elif len(s) == 1:
            t = torch.ones(s) if ("layer_norm" in k or "layernorm" in k or "_ln" in k) and k.endswith("weight") else torch.zeros(s)
        else:
            t = torch.empty(s).uniform_(-0.03, 0.03, generator=g)
        P[k] = t.requires_grad_(True)
    heads = cfg["encoder_embed_dim"] // 64
    img = torch.rand(pairs, 3, cfg["image_size"], cfg["image_size"], generator=g)
    ids = torch.randint(1, cfg["vocab_size"], (pairs, cfg["max_text_len"]), generator=g)
    mask = torch.ones(pairs, cfg["max_text_len"], dtype=torch.long)
    t0 = time.perf_counter()
    out = ostep.m2_itc(P, img, ids, mask, heads=heads, patch=cfg["patch_size"])
    out["loss"].backward()
    dt = time.perf_counter() - t0
    q.put(dict(value=round(pairs / dt, 4), unit="pairs/s", cores=cores, kind="port",
               sample=f"oracle.step.m2_itc fwd+bwd (no optimizer), fp32, {pairs} pairs of the same shapes, one pass of {dt:.1f} s, {cores} threads"))


def cpu_baseline(cfg, pairs, limit_s=300):
    """Runs the CPU sample in a child process with a hard wall-clock limit so the bench line is always printed."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(cfg, pairs, q))
    p.start()
    p.join(limit_s)
    if p.is_alive():
        p.terminate()
        p.join()
        return dict(value=None, unit="pairs/s", cores=os.cpu_count() or 1, kind="port",
                    sample=f"oracle.step.m2_itc on {pairs} pairs did not finish within the {limit_s} s bound on this host")
    return q.get() if not q.empty() else None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs a torch.distributed.run launch with --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from antmmf.hip import _lib, functional, ops
    from antmmf.hip.arena import HipAdamW
    from vlmo.config import default_config
    from vlmo.modules.vlmo_module import VLMo

    assert _lib.backend() == 1, "bench.py must run on the gfx950 library"
    # activation-memory policy: keep the 4d-wide normalised FFN activation when the device has the HBM for it
    keep_ffn = (not a.recompute_ffn_norm) and torch.cuda.get_device_properties(device).total_memory >= 250 * 2 ** 30 and a.batch <= 1024
    functional.set_keep_ffn_norm(keep_ffn)
    cfg = default_config()
    cfg.update(WORKLOADS[a.workload])
    torch.manual_seed(1234)  # identical replicas on every rank
    model = VLMo(cfg).to(device).train()
    opt = HipAdamW([{"params": [p for p in model.parameters() if p.requires_grad]}], lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)
    batch = synthetic_batch(cfg, a.batch, device, rank)

    def step():
        out = model(batch)
        loss = out["losses"]["itc_loss"] + out["losses"]["itc_vl_loss"]
        loss.backward()
        w = opt.arena.allreduce_grads()
        opt.grad_scale = 1.0 / w
        opt.step()
        opt.zero_grad()
        return loss

    loss0 = None
    for _ in range(a.warmup):
        loss = step()
        if loss0 is None:
            loss0 = float(loss.detach())  # loss of the untouched random init: depends on the forward numerics only
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.GEMM_TRACE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        pairs_per_s = a.batch * world * a.steps / elapsed
        gemm_ms = sum(t[0].elapsed_time(t[1]) for t in trace)
        gemm_flops = sum(t[2] for t in trace)
        # algorithmic bytes of a launch: both operands once + the output once (bf16; fp32 for the wgrad accumulators)
        gemm_bytes = sum(2.0 * (t[4][0] * t[4][2] + t[4][1] * t[4][2]) + (4.0 if t[4][3] == "wgrad" else 2.0) * t[4][0] * t[4][1] for t in trace)
        n = max(1, len(trace))
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        per_layout = {}
        for e0, e1, f, tag, _shape in trace:
            d = per_layout.setdefault(tag, [0.0, 0.0, 0])
            d[0] += f; d[1] += e0.elapsed_time(e1); d[2] += 1
        if a.gemm_table:
            shapes = {}
            for e0, e1, f, tag, shp in trace:
                d = shapes.setdefault((tag,) + tuple(shp), [0.0, 0.0, 0])
                d[0] += f; d[1] += e0.elapsed_time(e1); d[2] += 1
            with open(a.gemm_table, "w") as fh:
                fh.write("layout I J R epilogue calls_per_step ms_per_step avg_us tflops\n")
                for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                    fh.write(" ".join(str(x) for x in k) + f" {v[2] / a.steps:.0f} {v[1] / a.steps:.3f} {v[1] / v[2] * 1e3:.1f} {v[0] / (v[1] * 1e-3) / 1e12:.1f}\n")
        # HBM-side traffic per GEMM launch: measured in separate rocprofv3 --pmc passes over this same command
        # (tools/gpu_traffic.sh) and committed under profiles/; null when no measurement matches the configuration
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"gemm_traffic_{a.workload}_b{a.batch}.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = int(json.load(fh)["traffic_bytes_per_launch"])  # bytes per launch, like `achieved`
        step_tflops = pairs_per_s / world * TRAIN_GFLOP_PER_PAIR[a.workload] / 1e3
        out = {
            "metric": ("image-text pairs/sec/node, M2_Encoder ViT-L/14 ITC, global batch 8192" if a.workload == "l14"
                       else "image-text pairs/sec/node, M2_Encoder ViT-B/16 ITC (secondary workload, not the BASELINE metric)"),
            "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"M2_Encoder {'ViT-L/14 (beit large, patch 14, 21+3 layers)' if a.workload == 'l14' else 'ViT-B/16 (beit base, 9+3 layers)'} ITC train step, 224x224x3 + 77 tokens",
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       "loss": round(final_loss, 5), "step_tflops_per_gpu": round(step_tflops, 1),
                       "step_frac_of_bf16_peak": round(step_tflops / PEAK_TFLOPS, 4),
                       "loss_step0": None if loss0 is None else round(loss0, 5), "keep_ffn_norm": keep_ffn,
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       "reserved_hbm_gib": round(torch.cuda.max_memory_reserved() / 2 ** 30, 1)},
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel (bf16 MFMA GEMM, all layouts)", "achieved": round(achieved, 1),
                         "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (L2-miss side: rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/gemm_traffic_*.json); algorithmic bytes per launch = 2(I R + J R + I J)",
                         "launches_per_step": n // a.steps, "avg_launch_ms": round(gemm_ms / n, 4),
                         "avg_launch_gflop": round(gemm_flops / n / 1e9, 2), "avg_launch_algorithmic_bytes": int(gemm_bytes / n), "gemm_ms_per_step": round(gemm_ms / a.steps, 2),
                         "by_layout_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in per_layout.items() if v[1] > 0}},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
