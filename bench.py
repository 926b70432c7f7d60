#!/usr/bin/env python
"""bench.py -- pairs/s of the contrastive image/video-text training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload l14|b16|vtp8|vtp8t|dmae12]      (N > 1: starts its own N ranks under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default workload `l14` (the BASELINE metric): one step = forward + backward + optimizer over one synthetic batch already resident in
HBM: B = 1024 pairs per GPU (224x224x3 frames, 77-token captions, ragged true lengths), M2 `large` towers with patch 14 (257 visual
tokens, 21+3 layers, d = 1024; SURVEY.md "Facts" 3), random-init weights, bf16 MFMA compute with fp32 masters, global negatives
all-gathered over RCCL (one packed message), row-sharded symmetric InfoNCE on both ITC levels, gradient all-reduce of the flat arena started
from inside the backward pass, fused AdamW.  Weak scaling: the per-GPU batch is fixed, so the global batch is 1024 x N (8192 at N = 8).
The step is BaseTrainer.train_step -- the trainer's own loop body (forward, device-side meters, backward, reduction, optimizer, LR
schedule), not a hand-rolled loop.

Other workloads (BASELINE.json configs 1, 3, 4; their own metric strings, never the BASELINE line):
  b16     M2 `base` (ViT-B/16 dims) ITC, 1024 pairs per GPU
  vtp8    prj/base_vtp `univl` (clip arch: ViT-B/16 + BERT-base, vocab 21128), 8 clips per video, stage1 (MIL-NCE over the 8 x B_g clips)
          + stage2 (cross-modal merged attention over [text ; clips ; SEP] through the text tower's layers for every text x video pair)
  vtp8t   config 3 with its temporal module: 8 frames, 77 tokens, stage1 + stage3 = the CLIP4Clip temporal transformer (seqTransf, 4 layers) over the frame tokens
          + WTI scores + CrossEn in both directions (TPM-CL off)
  dmae12  prj/dmae_vtp `univl`, 12 frames, 30-word captions, stage1 + stage3 (seqTransf temporal transformer, WTI similarity, NegNCE, TPM-CL type 4)

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the dominant kernel family (the bf16 MFMA GEMMs): algorithmic flops per launch / average launch duration, measured live
                with HIP events on the launch stream over the timed steps; peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/, plain torch fp32 restatement of the reference; kind "port") timed on this host's cores on a bounded
                sample of the same workload: 1 warm-up + 3 timed passes of forward + backward (NO optimizer step: stated), same ragged
                masks; for `l14` also the SURVEY 8(d) config-0 leg (clip-arch ViT-B/16 + BERT-base, B = 8).  Rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "ant-multi-modal-framework_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(PKG, "prj", "M2_Encoder"))

PEAK_TFLOPS = 2500.0

M2_WORKLOADS = {
    # M2 `large`, patch 14: 21 + 3 layers, d = 1024, 16 heads, 257 image tokens, 77 text tokens, vocab 115244, D = 1024
    "l14": dict(beit_version="large", encoder_embed_dim=1024, out_embed_dim=1024, encoder_layers=21, beit3_vl_layers=3,
                image_size=224, patch_size=14, vocab_size=115244, max_text_len=77),
    # config[1]: M2 `base`, patch 16 (ViT-B/16 dims + 12-layer text stack)
    "b16": dict(beit_version="base", encoder_embed_dim=768, out_embed_dim=768, encoder_layers=9, beit3_vl_layers=3,
                image_size=224, patch_size=16, vocab_size=64010, max_text_len=77),
}
M2_TRAIN_GFLOP_PER_PAIR = {"l14": 627.3, "b16": 145.3}  # BASELINE.md section 4 (algorithmic, train = 3 x forward, recompute not counted)

# DRY RUN (ANTMMF_BENCH_DRY_RUN=1; tests/test_host_logic.py::test_bench_two_ranks_dry_run only): the same main() on the host -- gloo instead of RCCL, the
# lane-emulated kernels, a toy M2 -- so that everything the first real N > 1 run depends on besides the GPUs themselves is exercised where there are none:
# self-launch under torch.distributed.run, rendezvous, the probe step + agreed activation policy, barriers, max-over-ranks timing, rank 0's JSON line.
# Its numbers mean nothing and its line says so (`"data": "dry run"`).
DRY_RUN = os.environ.get("ANTMMF_BENCH_DRY_RUN") == "1"
if DRY_RUN:
    M2_WORKLOADS["tiny"] = dict(beit_version="base", encoder_embed_dim=64, out_embed_dim=64, encoder_layers=1, beit3_vl_layers=1,
                                image_size=16, patch_size=8, vocab_size=300, max_text_len=8, encoder_attention_heads=1)
    M2_TRAIN_GFLOP_PER_PAIR["tiny"] = 0.01


class _Hw:
    """The device plumbing main() needs.  Real: the MI355X through torch.cuda.  Dry run: the host, with a pretend 288-GiB device whose memory numbers send an
    N > 1 run through choose_keep_ffn's probe arithmetic."""

    def __init__(self, local_rank):
        self.dry = DRY_RUN
        if self.dry:
            self.device, self.backend = torch.device("cpu"), "gloo"
        else:
            torch.cuda.set_device(local_rank)
            self.device, self.backend = torch.device("cuda", local_rank), "nccl"

    @staticmethod
    def device_count():
        return 64 if DRY_RUN else torch.cuda.device_count()

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def total_memory(self):
        return 288 * 2 ** 30 if self.dry else torch.cuda.get_device_properties(self.device).total_memory

    def reset_peak(self):
        if not self.dry:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()

    def max_reserved(self):
        return 2 ** 30 if self.dry else torch.cuda.max_memory_reserved()

    def max_allocated(self):
        return 2 ** 30 if self.dry else torch.cuda.max_memory_allocated()

    def free_and_reserved(self):
        if self.dry:   # pretend device: 1 GiB in torch's pool and ANTMMF_BENCH_DRY_OUTSIDE_GIB (default 7) outside it -- RCCL's buffers in a real N > 1 run
            outside = float(os.environ.get("ANTMMF_BENCH_DRY_OUTSIDE_GIB", "7"))
            return int((288 - 1 - outside) * 2 ** 30), 2 ** 30
        return torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()

    def init_process_group(self, rank, world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not self.dry:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=self.device)
            return
        dist.init_process_group("gloo", rank=rank, world_size=world)
        real = dist.reduce_scatter_tensor   # gloo has no reduce_scatter_tensor (RCCL has): all_reduce + slice, dry run only

        def reduce_scatter_tensor(output, input, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if dist.get_backend(group) != "gloo":
                return real(output, input, op=op, group=group, async_op=async_op)
            full = input.clone()
            dist.all_reduce(full, op=op, group=group)
            n = output.shape[0]
            output.copy_(full[dist.get_rank(group) * n:(dist.get_rank(group) + 1) * n])

        dist.reduce_scatter_tensor = reduce_scatter_tensor

CLIP_B16 = dict(image_encoder=dict(type="VitImageEncoder", params=dict(model_name="ViT-B-16", input_resolution=224, patch_size=16, width=768,
                                                                       layers=12, out_dim=768, pretrained=False)),
                text_encoder=dict(type="RobertBertEncoder", params=dict(pretrained=False, vocab_size=21128, hidden_size=768, intermediate_size=3072,
                                                                        num_hidden_layers=12, num_attention_heads=12, max_position_embeddings=512,
                                                                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, out_dim=768, is_proj=True)))
VTP_WORKLOADS = {
    "vtp8": dict(prj="base_vtp", n_clips=8, seq=77, default_batch=64,
                 model=dict(training_head_type="video_text_retrieval", arch_type="clip", training_stage="stage1+stage2", with_moco=False,
                            with_cross_encoder=True, hidden_size=768, **CLIP_B16)),
    # config 3 WITH its temporal module ("n = 8, temporal encoder"; VERDICT r4 missing 4): the CLIP4Clip temporal transformer over the 8 frame tokens
    # (seqTransf header, 4 layers, prj/dmae_vtp/.../dmae_utils.py:186-227) feeding the token-wise retrieval scores, next to the stage-1 MIL-NCE
    "vtp8t": dict(prj="dmae_vtp", n_clips=8, seq=77, default_batch=128,
                  model=dict(training_head_type="video_text_retrieval", arch_type="clip", training_stage="stage1+stage3", with_moco=False,
                             with_cross_encoder=False, hidden_size=768, l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="seqTransf",
                             l3_sim_header_hidden_layer=4, l3_partial_type=-1, l3_max_frames=8, l3_max_words=77, l3_loss_type="cross_entropy", **CLIP_B16)),
    "dmae12": dict(prj="dmae_vtp", n_clips=12, seq=30, default_batch=128,
                   model=dict(training_head_type="video_text_retrieval", arch_type="clip", training_stage="stage1+stage3", with_moco=False,
                              with_cross_encoder=False, hidden_size=768, l3_interaction="wti", l3_with_nfc=True, l3_wti_arch=1, l3_sim_header="seqTransf",
                              l3_sim_header_hidden_layer=4, l3_partial_type=4, l3_max_frames=12, l3_max_words=30, l3_loss_type="negNCE", **CLIP_B16)),
}


def tower_gflop(L, N, d, extra=0.0):
    """forward GFLOP of a transformer tower over one sequence: L N (24 d^2 + 4 N d)."""
    return (L * N * (24.0 * d * d + 4.0 * N * d) + extra) / 1e9


def vtp_train_gflop_per_pair(name, batch_global, batch_local):
    """Algorithmic training GFLOP (3 x forward) per video-text pair of the video workloads."""
    w = VTP_WORKLOADS[name]
    n, seq, d = w["n_clips"], w["seq"], 768
    vit = tower_gflop(12, 197, d, extra=2.0 * 196 * 3 * 256 * d + 2.0 * d * d)   # per frame
    bert = tower_gflop(12, seq, d, extra=2.0 * d * d)                             # per caption
    fwd = n * vit + bert
    if name == "vtp8":   # stage 2: every local caption against every (gathered) video: B_g sequences of (seq + n + 1) tokens per caption
        fwd += batch_global * tower_gflop(12, seq + n + 1, d, extra=2.0 * d * 2 * d)
    else:                # stage 3: 4-layer temporal transformer over n + 1 tokens per video + the token-wise similarity against the global batch
        fwd += tower_gflop(4, n + 1, d) + 2.0 * batch_global * seq * (n + 1) * d / 1e9
    return 3.0 * fwd


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="l14", choices=sorted(list(M2_WORKLOADS) + list(VTP_WORKLOADS)))
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default: 1024 for the M2 workloads, 64 / 128 videos for vtp8 / dmae12)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recompute-ffn-norm", action="store_true", help="do not keep ffn_layernorm(gelu(u)) for backward (saves ~65 GiB at 1024 pairs/GPU, costs ~3 %%)")
    ap.add_argument("--gemm-table", default=None, help="write per-shape GEMM timing (from the live HIP-event trace) to this file")
    ap.add_argument("--cpu-sample", type=int, default=2, help="pairs per pass of the CPU-oracle sample")
    return ap.parse_args()


def ragged_captions(batch, seq, vocab, g, device, cls_id=None):
    ids = torch.randint(1, vocab, (batch, seq), generator=g, device=device)
    lengths = torch.randint(min(8, seq), seq + 1, (batch,), generator=g, device=device)
    mask = (torch.arange(seq, device=device)[None, :] < lengths[:, None]).long()
    ids = ids * mask
    if cls_id is not None:
        ids[:, 0] = cls_id
    return ids, mask


def synthetic_m2_batch(cfg, batch, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    img = torch.rand(batch, 3, cfg["image_size"], cfg["image_size"], generator=g, device=device)
    ids, mask = ragged_captions(batch, cfg["max_text_len"], cfg["vocab_size"], g, device)
    return {"image": [img], "text_ids": ids, "text_masks": mask}


def synthetic_vtp_batch(name, batch, device, seed):
    from antmmf.structures.sample import SampleList

    w = VTP_WORKLOADS[name]
    g = torch.Generator(device=device).manual_seed(seed)
    n = w["n_clips"]
    frames = torch.randn(batch, n, 3, 224, 224, generator=g, device=device)   # already-normalised frames
    ids, mask = ragged_captions(batch, w["seq"], 21128, g, device, cls_id=101)
    if name == "dmae12":   # (vtp8t keeps ragged captions: TPM-CL, the part that needs full-length captions, is off there)
        mask = torch.ones_like(mask)   # DMAE's token predictors are built for exactly l3_max_words tokens (tpmcl_utils.py:21-24)
        ids = torch.where(ids == 0, torch.ones_like(ids), ids)
    return SampleList(image_data=frames, image_pad_mask=torch.zeros(batch, n, 224, 224, dtype=torch.bool, device=device), image_n_clips=[n] * batch,
                      image_num_frames=[1] * batch, caption_input_ids=ids, caption_raw_input_ids=ids, caption_input_mask=mask, dataset_type="train")


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle; separate process)
def _time_passes(fn, warm=1, timed=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def _cpu_baseline_worker(workload, pairs, q):
    """The CPU oracle's training step (forward + backward, fp32, no optimizer) on `pairs` pairs of the same shapes and the same kind of
    ragged masks; 1 warm-up + 3 timed passes; puts a dict on `q`."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import step as ostep

    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)  # torch's CPU GEMMs stop scaling well before one thread per core on these shapes; 32 threads, or every core of a smaller host
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    dev = torch.device("cpu")

    def rand_params(shapes):
        P = {}
        for k, s in shapes.items():
            if len(s) == 0:
                t = torch.tensor(2.659)
            elif len(s) == 1:
                t = torch.ones(s) if ("layer_norm" in k or "layernorm" in k or "_ln" in k or "LayerNorm" in k or "ln_" in k) and k.endswith("weight") else torch.zeros(s)
            else:
                t = torch.empty(s).uniform_(-0.03, 0.03, generator=g)
            P[k] = t.requires_grad_(True)
        return P

    def clip_leg(n_pairs, n_clips, seq, stage):
        import tiny_models

        c = dict(width=768, layers=12, heads=12, patch=16, res=224, out_dim=768, vocab=21128, hidden=768, inter=3072, bert_layers=12, bert_heads=12, max_pos=512)
        shapes = tiny_models.clip_arch_shapes(c)
        if stage == "stage2":
            shapes.update({"similarity_dense.0.weight": (1536, 768), "similarity_dense.0.bias": (1536,), "similarity_dense.2.weight": (1, 1536), "similarity_dense.2.bias": (1,)})
        P = rand_params(shapes)
        frames = torch.randn(n_pairs, n_clips, 3, 224, 224, generator=g)
        ids, mask = ragged_captions(n_pairs, seq, 21128, g, dev, cls_id=101)

        def one():
            for p in P.values():
                p.grad = None
            if stage == "stage2":
                out = ostep.univl_stage2(P, frames, ids, mask, n_clips, 12, 16, 12)
                loss = out["loss1"] + out["loss2"] if "loss1" in out else out["loss"]
            else:
                loss = ostep.univl_stage1(P, frames, ids, mask, n_clips, 12, 16, 12)["loss"]
            loss.backward()

        return one

    if workload in M2_WORKLOADS:
        from oracle.shapes import m2_shapes

        cfg = M2_WORKLOADS[workload]
        P = rand_params(m2_shapes(d=cfg["encoder_embed_dim"], layers=cfg["encoder_layers"], vl_layers=cfg["beit3_vl_layers"], patch=cfg["patch_size"],
                                  res=cfg["image_size"], vocab=cfg["vocab_size"], out=cfg["out_embed_dim"]))
        heads = cfg["encoder_embed_dim"] // 64
        img = torch.rand(pairs, 3, cfg["image_size"], cfg["image_size"], generator=g)
        ids, mask = ragged_captions(pairs, cfg["max_text_len"], cfg["vocab_size"], g, dev)

        def one():
            for p in P.values():
                p.grad = None
            ostep.m2_itc(P, img, ids, mask, heads=heads, patch=cfg["patch_size"])["loss"].backward()

        ts = _time_passes(one)
        res = dict(value=round(pairs / (sum(ts) / len(ts)), 4), unit="pairs/s", cores=cores, kind="port",
                   sample=f"oracle.step.m2_itc forward + backward (no optimizer step), fp32, {pairs} pairs of the same shapes with ragged caption masks, "
                          f"1 warm-up + {len(ts)} timed passes ({', '.join(f'{t:.1f}' for t in ts)} s), {cores} threads")
        if workload == "l14":   # clip-arch ViT-B/16 + BERT-base, stage 1, B = 8
            t0 = _time_passes(clip_leg(8, 1, 77, "stage1"))
            res["config0"] = dict(value=round(8 / (sum(t0) / len(t0)), 4), unit="pairs/s", cores=cores, kind="port",
                                  sample=f"oracle.step.univl_stage1 (clip arch ViT-B/16 + BERT-base, B = 8, 77 tokens) forward + backward, fp32, 1 warm-up + {len(t0)} timed passes "
                                         f"({', '.join(f'{t:.1f}' for t in t0)} s), {cores} threads")
        q.put(res)
        return
    w = VTP_WORKLOADS[workload]
    if workload == "vtp8":
        one = clip_leg(pairs, w["n_clips"], w["seq"], "stage2")
        what = "oracle.step.univl_stage2 (stage1 + stage2 cross-encoder)"
    else:   # the oracle restates stage 3 without TPM-CL (l3_partial_type -1): the towers dominate the CPU time either way
        one = clip_leg(pairs, w["n_clips"], w["seq"], "stage1")
        what = f"oracle.step.univl_stage1 on {w['n_clips']} frames x {w['seq']} words (towers + MIL-NCE; the stage-3 head is < 1 % of the CPU time)"
    ts = _time_passes(one, warm=1, timed=2)
    q.put(dict(value=round(pairs / (sum(ts) / len(ts)), 4), unit="pairs/s", cores=cores, kind="port",
               sample=f"{what} forward + backward (no optimizer step), fp32, {pairs} video-text pairs, 1 warm-up + {len(ts)} timed passes "
                      f"({', '.join(f'{t:.1f}' for t in ts)} s), {cores} threads"))


def cpu_baseline(workload, pairs, limit_s=300):
    """Runs the CPU sample in a child process with a hard wall-clock limit so the bench line is always printed."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(workload, pairs, q))
    p.start()
    p.join(limit_s)
    if p.is_alive():
        p.terminate()
        p.join()
        return dict(value=None, unit="pairs/s", cores=os.cpu_count() or 1, kind="port",
                    sample=f"the oracle sample of {pairs} pairs did not finish within the {limit_s} s bound on this host")
    return q.get() if not q.empty() else None


# ------------------------------------------------------------------------------------------------ trainer around the bench model
def make_trainer(a, device, world):
    """BaseTrainer with the bench's model / optimizer: its train_step is the timed unit."""
    from antmmf.common.configuration import Configuration
    from antmmf.hip.arena import HipAdamW
    from antmmf.trainers.base_trainer import BaseTrainer

    tp = {"trainer": "base_trainer", "device": device.type, "log_interval": 10 ** 9, "max_iterations": 10 ** 9, "seed": 1234, "lr_scheduler": True,
          "use_warmup": True, "warmup_iterations": 1000, "warmup_factor": 0.2, "clip_gradients": False}
    if a.workload in M2_WORKLOADS:
        from vlmo.config import default_config
        from vlmo.modules.vlmo_module import VLMo

        mcfg = default_config()
        mcfg.update(M2_WORKLOADS[a.workload])

        class BenchTrainer(BaseTrainer):
            def load_model(self):
                torch.manual_seed(1234)  # identical replicas on every rank
                self.model = VLMo(mcfg).to(self.device).train()

            def load_optimizer(self):
                self.optimizer = HipAdamW([{"params": [p for p in self.model.parameters() if p.requires_grad]}], lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)
                self.arena = self.optimizer.arena

        cfg = Configuration({"training_parameters": tp, "optimizer_attributes": {"type": "AdamW", "params": {"lr": 1e-4, "weight_decay": 0.05}},
                             "model_attributes": {"m2_encoder": {}}})
        return BenchTrainer(cfg)
    w = VTP_WORKLOADS[a.workload]
    sys.path.insert(0, os.path.join(PKG, "prj", w["prj"]))
    import roi_univl  # noqa: F401  (registers `univl` + the encoders of the chosen project)

    class VtpTrainer(BaseTrainer):
        def load_model(self):
            torch.manual_seed(1234)
            super().load_model()   # build_model("univl") from the config: the registry path of prj/*_vtp

    mattrs = dict(w["model"])
    if a.workload == "vtp8":   # all caption rows of the rank in ONE cross-encoder call (the reference chunks by 5 rows to bound its memory): 256-aligned token counts
        mattrs["cross_chunk_rows"] = a.batch if a.batch is not None else w["default_batch"]
    cfg = Configuration({"training_parameters": tp, "optimizer_attributes": {"type": "AdamW", "params": {"lr": 1e-4, "weight_decay": 0.05, "betas": [0.9, 0.98], "eps": 1e-6}},
                         "model_attributes": {"univl": mattrs}})
    return VtpTrainer(cfg)


def _self_launch(a):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this file under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1, a free port) and hand its exit code back -- the reference's own launcher does the same with one Popen per local
    rank (antmmf/utils/launch.py:220-282).  The ranks print the JSON line; this parent prints nothing."""
    import socket
    import subprocess

    have = _Hw.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: this node exposes {have} GPU(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver
    env["ANTMMF_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def choose_keep_ffn(a, is_m2, batch_size, hw, world, probe_step):
    """Activation policy: keep fc2's 4d-wide input for backward when HBM allows.  N = 1: the l14 step peaks at 244 GiB
    reserved of 268 (measured, profiles/).  N > 1: RCCL's channel buffers and the bucket staging live OUTSIDE torch's allocator, so the
    decision is made from a measurement on this very process -- one step in recompute mode (`probe_step`), the bytes the kept activation adds
    on top of its peak (analytic), and what the device still has free next to torch's pool -- and agreed over all ranks (MIN)."""
    from antmmf.hip import functional

    device = hw.device
    total = hw.total_memory()
    if a.recompute_ffn_norm:
        return False, "recompute (--recompute-ffn-norm)"
    if total < 250 * 2 ** 30 or (is_m2 and batch_size > 1024):
        return False, "recompute (HBM)"
    if world == 1:
        return True, "kept"
    functional.set_keep_ffn_norm(False)
    hw.reset_peak()
    probe_step()
    hw.sync()
    reserved = hw.max_reserved()
    free, reserved_now = hw.free_and_reserved()
    outside = total - free - reserved_now                         # RCCL buffers, HIP runtime, code objects
    if is_m2:
        m = M2_WORKLOADS[a.workload]
        tokens = batch_size * ((m["image_size"] // m["patch_size"]) ** 2 + 1 + m["max_text_len"])
        extra = tokens * 4 * m["encoder_embed_dim"] * 2 * (m["encoder_layers"] + m["beit3_vl_layers"])
    else:
        w = VTP_WORKLOADS[a.workload]
        extra = batch_size * (w["n_clips"] * 197 + w["seq"]) * 3072 * 2 * 12
    margin = 6 * 2 ** 30
    keep = reserved + extra + outside + margin <= total
    flag = torch.tensor([1 if keep else 0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    keep = bool(flag.item())
    note = (f"{'kept' if keep else 'recompute'} (probe: {reserved / 2 ** 30:.1f} GiB reserved in recompute mode + {extra / 2 ** 30:.1f} GiB kept activations + "
            f"{outside / 2 ** 30:.1f} GiB outside torch's pool (RCCL, runtime) + {margin / 2 ** 30:.0f} GiB margin vs {total / 2 ** 30:.1f} GiB)")
    return keep, note


def _gemm_clock_mhz(device):
    """Effective shader clock while the persistent NT GEMM runs (outside the timed region): workgroup 0 of gemm_nt_k64r_kernel reads the shader cycle counter and the
    constant 100-MHz counter around its whole run on every launch (two scalar reads); antmmf_debug_gemm_clock returns the last launch's pair."""
    import ctypes

    from antmmf.hip import _lib, ops

    try:
        lib = _lib.load()
        X = torch.randn(257 * 1024, 1024, device=device).bfloat16()
        W = (torch.randn(1024, 1024, device=device) * 0.03).bfloat16()
        for _ in range(4):
            ops.gemm(X, W)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 2)()
        rc = lib.antmmf_debug_gemm_clock(buf)
        if rc != 0 or buf[1] == 0:
            return None
        return round(buf[0] / (buf[1] / 100.0), 0)   # cycles per microsecond = MHz
    except Exception:   # a probe must never cost the bench line
        return None


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _self_launch(a)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    hw = _Hw(local_rank)
    device = hw.device
    if world > 1:
        hw.init_process_group(rank, world)

    from antmmf.hip import _lib, functional, ops

    assert hw.dry or _lib.backend() == 1, "bench.py must run on the gfx950 library"
    is_m2 = a.workload in M2_WORKLOADS
    batch_size = a.batch if a.batch is not None else (1024 if is_m2 else VTP_WORKLOADS[a.workload]["default_batch"])
    trainer = make_trainer(a, device, world)
    trainer.load()
    trainer.model.train()
    batch = synthetic_m2_batch(M2_WORKLOADS[a.workload], batch_size, device, 1234 + rank) if is_m2 else synthetic_vtp_batch(a.workload, batch_size, device, 1234 + rank)
    # every rank is really there: an RCCL all-reduce of ones
    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        ranks_seen = int(t.item())

    def step():
        trainer.current_iteration += 1
        return trainer.train_step(batch)

    # activation-memory policy: fc2's 4d-wide input kept for backward instead of recomputed when the device has the HBM for it (the video
    # workloads peak at ~90 GiB without it, the M2 ones fit up to 1024 pairs); at N > 1 decided from a probe step (RCCL's buffers count)
    probe_loss = []
    keep_ffn, keep_note = choose_keep_ffn(a, is_m2, batch_size, hw, world, lambda: probe_loss.append(float(step().detach())))
    functional.set_keep_ffn_norm(keep_ffn)
    hw.reset_peak()

    loss0 = probe_loss[0] if probe_loss else None   # (N > 1: the probe step was the first step of the run)
    for _ in range(a.warmup):
        loss = step()
        if loss0 is None:
            loss0 = float(loss.detach())  # loss of the untouched random init: depends on the forward numerics only
    if world > 1:
        dist.barrier()
    hw.sync()
    ops.GEMM_TRACE = None if hw.dry else []   # (HIP events on the launch stream: none on the host)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    hw.sync()
    if world > 1:
        dist.barrier()
    hw.sync()
    elapsed = time.perf_counter() - t0
    trace, ops.GEMM_TRACE = (ops.GEMM_TRACE or []), None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    meters = trainer.read_meters()
    gemm_clock_mhz = _gemm_clock_mhz(device) if rank == 0 and not hw.dry else None

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        pairs_per_s = batch_size * world * a.steps / elapsed
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
        tpath = os.path.join(ROOT, "profiles", f"gemm_traffic_{a.workload}_b{batch_size}.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = int(json.load(fh)["traffic_bytes_per_launch"])  # bytes per launch, like `achieved`
        gflop_pair = M2_TRAIN_GFLOP_PER_PAIR[a.workload] if is_m2 else vtp_train_gflop_per_pair(a.workload, batch_size * world, batch_size)
        step_tflops = pairs_per_s / world * gflop_pair / 1e3
        metric = {"l14": "image-text pairs/sec/node, M2_Encoder ViT-L/14 ITC, global batch 8192",
                  "b16": "image-text pairs/sec/node, M2_Encoder ViT-B/16 ITC (BASELINE config 1; not the BASELINE metric)",
                  "vtp8": "video-text pairs/sec/node, base_vtp univl clip-arch ViT-B/16 + BERT-base, 8 clips, stage1 + stage2 cross-encoder (BASELINE config 3; not the BASELINE metric)",
                  "vtp8t": "video-text pairs/sec/node, univl clip-arch ViT-B/16 + BERT-base, 8 frames, stage1 + temporal transformer (seqTransf) + WTI retrieval scores (BASELINE config 3 with its temporal module; not the BASELINE metric)",
                  "dmae12": "video-text pairs/sec/node, dmae_vtp univl, 12 frames x 30 words, stage1 + stage3 NegNCE + TPM-CL (BASELINE config 4; not the BASELINE metric)",
                  "tiny": "dry run of the control flow (toy M2; not a metric)"}[a.workload]
        workload = {"l14": "M2_Encoder ViT-L/14 (beit large, patch 14, 21+3 layers) ITC train step, 224x224x3 + 77 tokens",
                    "b16": "M2_Encoder ViT-B/16 (beit base, 9+3 layers) ITC train step, 224x224x3 + 77 tokens",
                    "vtp8": "univl (clip arch) video-text train step: 8 clips x 224x224x3 per video + 77 tokens, MIL-NCE over all clips + cross-encoder scores of every text x video pair",
                    "vtp8t": "univl video-text train step: 8 frames x 224x224x3 per video + 77 tokens, MIL-NCE + 4-layer temporal transformer over the frame tokens / WTI / CrossEn",
                    "dmae12": "univl (DMAE) video-text train step: 12 frames x 224x224x3 per video + 30 words, MIL-NCE + seqTransf / WTI / NegNCE / TPM-CL",
                    "tiny": "toy M2 (d = 64, 1 + 1 layers, 16 x 16 images, 8 tokens)"}[a.workload]
        out = {
            "metric": metric, "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "dry run (host, gloo, emulated kernels): not a measurement" if hw.dry else "synthetic",
            "config": {"workload": workload, "per_gpu_batch": batch_size, "global_batch": batch_size * world, "parallelism": f"dp{world}", "ranks_seen": ranks_seen,
                       "step": "BaseTrainer.train_step (forward, device-side meters, backward, arena all-reduce, fused AdamW, LR schedule)",
                       "loss": round(final_loss, 5), "meters": {k: round(v, 5) for k, v in meters.items()},
                       "train_gflop_per_pair": round(gflop_pair, 1), "step_tflops_per_gpu": round(step_tflops, 1),
                       "step_frac_of_bf16_peak": round(step_tflops / PEAK_TFLOPS, 4),
                       "loss_step0": None if loss0 is None else round(loss0, 5), "keep_ffn_norm": keep_ffn, "ffn_activation_policy": keep_note,
                       # N > 1: the gradient buckets of the last step in launch order (which went to RCCL from inside backward, which at the end, and when)
                       "grad_buckets": getattr(trainer.arena, "last_bucket_log", None) if world > 1 else None,
                       "peak_hbm_gib": round(hw.max_allocated() / 2 ** 30, 1),
                       "reserved_hbm_gib": round(hw.max_reserved() / 2 ** 30, 1),
                       # WHICH library served the step (ANTMMF_HIP_LIB can point at another build of the ABI): path relative to the repo, lab or product, size in bytes
                       "library": {"path": os.path.relpath(_lib.lib_path(), os.path.dirname(os.path.abspath(__file__))), "lab": bool(_lib.is_lab()),
                                   "bytes": os.path.getsize(_lib.lib_path()) if os.path.isfile(_lib.lib_path()) else None}},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_k64r_kernel / gemm_tn_k64_kernel (bf16 MFMA GEMM family, all layouts)", "achieved": round(achieved, 1),
                         "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes per launch (L2-miss side: rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/gemm_traffic_*.json); algorithmic bytes per launch = 2(I R + J R + I J)",
                         "launches_per_step": n // a.steps, "avg_launch_ms": round(gemm_ms / n, 4),
                         "avg_launch_gflop": round(gemm_flops / n / 1e9, 2), "avg_launch_algorithmic_bytes": int(gemm_bytes / n), "gemm_ms_per_step": round(gemm_ms / a.steps, 2),
                         "by_layout_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in per_layout.items() if v[1] > 0},
                         # shader clock under the MFMA loop on THIS box (the pool's boxes differ by +-4 % at identical code: compare runs at equal clock)
                         "gemm_clock_mhz": gemm_clock_mhz},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
