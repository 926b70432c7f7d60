"""Trainer surface of the contrastive path: LR schedule, gradient accumulation under the reference key,
device-side gradient clipping, evaluation + early stopping, fp32 escape list, scheduled hard-mining ratio, `Univl` through
`build_model` with the four parameter groups, and RetrievalTrainer's evaluation against a direct full-matrix computation.
Runs the kernels on the CPU lane emulator."""
import os
import subprocess
from bisect import bisect

import pytest
import torch

import model_cases as mc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "build", "libantmmf_emu.so")


@pytest.fixture(scope="module", autouse=True)
def emu():
    from test_kernels_emu import _stale

    if _stale():
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    from antmmf.hip import _lib

    old = os.environ.get("ANTMMF_HIP_LIB")
    os.environ["ANTMMF_HIP_LIB"] = EMU_LIB
    _lib.reset_for_tests()
    yield
    if old is None:
        os.environ.pop("ANTMMF_HIP_LIB", None)
    else:
        os.environ["ANTMMF_HIP_LIB"] = old
    _lib.reset_for_tests()


def _toy_trainer(arena=True):
    from antmmf.common.registry import registry
    from antmmf.models.base_model import BaseModel
    from antmmf.optimizer import build_optimizer
    from antmmf.trainers.base_trainer import BaseTrainer

    class Scale(torch.nn.Module):  # a class the fp32 escape list can name
        def forward(self, x):
            self.seen_dtype = x.dtype
            return x * 1.5

    @registry.register_model("toy_trainer_model")
    class Toy(BaseModel):
        def build(self):
            self.enc_a = torch.nn.Linear(8, 8)
            self.scale = Scale()
            self.enc_b = torch.nn.Linear(8, 8)

        def forward(self, sample_list):
            self.last_incre = sample_list.get("incre_num", None)
            h = self.scale(torch.tanh(self.enc_a(sample_list["image_data"])).to(torch.bfloat16)).float()
            y = self.enc_b(h)
            return {"losses": {"toy_loss": ((y - sample_list["caption_target"]) ** 2).mean()}}

    class ArenaTrainer(BaseTrainer):
        def load_optimizer(self):
            self.optimizer = build_optimizer(self.model, self.config, use_hip_arena=arena)
            self.arena = getattr(self.optimizer, "arena", None)

    return ArenaTrainer


def _cfg(model_attrs=None, **tp):
    from antmmf.common.configuration import Configuration

    base = {"trainer": "base_trainer", "device": "cpu", "max_iterations": 8, "log_interval": 100, "seed": 7}
    base.update(tp)
    return Configuration({"training_parameters": base, "optimizer_attributes": {"type": "AdamW", "params": {"lr": 0.05, "weight_decay": 0.01}},
                          "model_attributes": {"toy_trainer_model": model_attrs or {}}, "amp_attributes": {"amp_escapes": tp.pop("amp_escapes", [])}})


def _batches(n=8, seed=11):
    from antmmf.structures.sample import SampleList

    g = torch.Generator().manual_seed(seed)
    return [SampleList(image_data=torch.randn(6, 8, generator=g), caption_target=torch.randn(6, 8, generator=g)) for _ in range(n)]


def test_lr_schedule_and_gradient_accumulation():
    """`lr_scheduler: true` = LambdaLR over the reference's lr_lambda_update (antmmf/utils/general.py:27-44: linear warm-up from
    warmup_factor, then lr_ratio ** bisect(lr_steps)); the scheduler and the optimizer advance once per `gradient_accumulation_steps`
    iterations (reference key, base.yml:174; base_trainer.py:604-607, 700-717)."""
    Trainer = _toy_trainer()
    tp = dict(lr_scheduler=True, use_warmup=True, warmup_iterations=2, warmup_factor=0.2, lr_steps=[3], lr_ratio=0.1, gradient_accumulation_steps=2)
    tr = Trainer(_cfg(**tp), _batches())
    tr.load()
    assert tr.gradient_accumulation_steps == 2 and tr.lr_scheduler is not None

    def ref_lambda(i):
        if i <= 2:
            a = i / 2.0
            return 0.2 * (1 - a) + a
        return pow(0.1, bisect([3], i))

    lrs, weights = [], []
    orig = tr._run_scheduler

    def spy():
        orig()
        lrs.append(tr.optimizer.param_groups[0]["lr"])
        weights.append(tr.model.enc_a.weight.detach().clone())

    tr._run_scheduler = spy
    tr.train()
    assert tr.current_iteration == 8
    steps = [0, 1, 1, 2, 2, 3, 3, 4]  # scheduler steps taken after iteration 1..8
    for got, k in zip(lrs, steps):
        assert abs(got - 0.05 * ref_lambda(k)) < 1e-12, (lrs, k)
    for i in range(0, 8, 2):  # weights move only on the second iteration of each accumulation pair
        if i:
            assert torch.equal(weights[i], weights[i - 1])
        assert not torch.equal(weights[i + 1], weights[i])
    # `update_frequency` (this build's round-1 name) still works as an alias
    tr2 = Trainer(_cfg(update_frequency=4), _batches())
    tr2.load()
    assert tr2.gradient_accumulation_steps == 4 and tr2.lr_scheduler is None


def test_device_side_clipping_matches_torch():
    """clip_gradients through the flat arena: norm and coefficient stay on the device and reach the fused AdamW as a tensor; one step
    equals torch.optim.AdamW after clip_grad_norm_ on the same model."""
    Trainer = _toy_trainer()
    batches = _batches(1)
    tr = Trainer(_cfg(clip_gradients=True, max_grad_l2_norm=0.05, max_iterations=1), batches)
    tr.load()
    ref = type(tr.model)(tr.model.config)
    ref.build()
    ref.load_state_dict({k: v.clone() for k, v in tr.model.state_dict().items()})
    opt = torch.optim.AdamW(ref.parameters(), lr=0.05, weight_decay=0.01)
    ref(batches[0])["losses"]["toy_loss"].backward()
    norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
    assert float(norm) > 0.05  # the clip is active
    opt.step()
    meters = tr.train()
    assert abs(meters["grad_norm"] - float(norm)) < 1e-4 * float(norm)
    for (n, a), (_, b) in zip(tr.model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-3, atol=2e-4, msg=n)


def test_evaluation_and_early_stopping():
    """evaluation_interval + should_early_stop + patience on `monitored_metric` (reference `_logistics`, base_trainer.py:473-530): with a
    zero learning rate the validation loss never improves, so training stops `patience` iterations after the first evaluation."""
    Trainer = _toy_trainer()
    cfg = _cfg(evaluation_interval=1, should_early_stop=True, patience=2, monitored_metric="total_loss", max_iterations=8)
    cfg.optimizer_attributes.params.lr = 0.0
    tr = Trainer(cfg, _batches())
    tr.load()
    tr.load_task(_batches(), _batches(2, seed=5))
    tr.train()
    assert tr.best_iteration == 1 and tr.current_iteration == 4, (tr.best_iteration, tr.current_iteration)
    assert "total_loss" in tr.last_evaluation and tr.last_evaluation["total_loss"] > 0


def test_ragged_padding_maximum_is_per_phase():
    """`pad_ragged_batches`: training steps pad to batch_size / W, an evaluation loop to test_batch_size / W and back -- a bigger test_batch_size must not make every
    TRAINING step pad to the eval size."""
    from antmmf.hip import contrastive

    Trainer = _toy_trainer()
    seen = []
    try:
        tr = Trainer(_cfg(batch_size=4, test_batch_size=16, pad_ragged_batches=True, evaluation_interval=2, max_iterations=4, monitored_metric="total_loss"), _batches(4))
        tr.load()
        assert contrastive._DEFAULT_MAX_ROWS == 4
        tr.load_task(_batches(4), _batches(2, seed=5))
        inner = tr.model.forward

        def spy(*a, **k):
            seen.append((tr.model.training, contrastive._DEFAULT_MAX_ROWS))
            return inner(*a, **k)

        tr.model.forward = spy
        tr.train()
        assert {m for t, m in seen if t} == {4} and {m for t, m in seen if not t} == {16}, seen
        assert contrastive._DEFAULT_MAX_ROWS == 4     # back to the training maximum after the last evaluation
    finally:
        contrastive.set_max_rows_per_rank(None)


def test_fp32_escape_list_and_hard_mining_ratio():
    """amp_attributes.amp_escapes (reference register_fp32.py:42-69): the named class receives fp32 inputs; unknown names warn.
    hard_example_mining + change_iter / change_rate put `incre_num` into the batch (reference base_trainer.py:552-571)."""
    from antmmf.common.configuration import Configuration

    Trainer = _toy_trainer()
    cfg = _cfg(model_attrs={"hard_example_mining": True, "change_iter": 2, "change_rate": 0.25}, max_iterations=5)
    cfg = Configuration({**cfg.to_dict(), "amp_attributes": {"amp_escapes": "Scale, NoSuchClass"}}) if hasattr(cfg, "to_dict") else cfg
    tr = Trainer(cfg, _batches())
    tr.load()
    tr.train()
    assert tr.model.scale.seen_dtype == torch.float32
    assert tr.model.last_incre == min(int(5 / 2) * 0.25, 1.0)
    tr2 = Trainer(_cfg(max_iterations=1), _batches())
    tr2.load()
    tr2.train()
    assert tr2.model.scale.seen_dtype == torch.bfloat16 and tr2.model.last_incre is None


def test_univl_through_build_model_and_four_param_groups(golden):
    """SURVEY 8a R2 on the emulator (the same case runs on the MI355X in tests/test_e2e_gpu.py)."""
    print(mc.case_univl_registry(torch.device("cpu"), golden))


def test_retrieval_trainer_evaluate_set():
    """8(f2): RetrievalTrainer._evaluate_set (reference retrieval_trainer.py:86-293) -- caption batches, video batches de-duplicated
    by video id, blocks scored from the cached stage-1 outputs -- equals GlobalRetrievalRecall on the full similarity matrix
    computed directly (all captions x distinct videos) with the same ground-truth lists."""
    import roi_univl  # noqa: F401
    import weightgen as W
    from antmmf.common.configuration import Configuration
    from antmmf.modules.metrics import global_retrieval_recall as grr
    from antmmf.structures.sample import SampleList
    from antmmf.trainers.build import build_trainer

    cfg = Configuration({"training_parameters": {"trainer": "retrieval_trainer", "device": "cpu", "max_iterations": 0, "log_interval": 100, "seed": 1},
                         "optimizer_attributes": {"type": "SGD", "params": {"lr": 0.0, "weight_decay": 0.0}},
                         "model_attributes": {"univl": dict(mc.TINY_CLIP_CFG)}})
    tr = build_trainer(cfg, [])
    tr.load()
    W.fill_module_(tr.model.model)
    # 6 captions over 4 distinct videos (video 1 has two captions, video 2 shows up in both batches), batches of 3 captions
    cap_vid = [0, 1, 1, 2, 2, 3]
    vids_t2v = [[v] for v in cap_vid]
    v2t = [[t for t, v in enumerate(cap_vid) if v == vid] for vid in range(4)]
    gen = torch.Generator().manual_seed(9)
    frames = torch.randn(4, 1, 3, 32, 32, generator=gen)
    ids = torch.randint(1, 300, (6, 12), generator=gen)
    ids[:, 0] = 101
    mask = torch.ones(6, 12, dtype=torch.long)
    mask[2, 8:] = 0
    mask[5, 5:] = 0
    batches = []
    for b in range(2):
        sl_ = slice(3 * b, 3 * b + 3)
        vv = cap_vid[sl_]
        batches.append(SampleList(
            caption_raw_input_ids=ids[sl_], caption_input_ids=ids[sl_], caption_input_mask=mask[sl_], caption_tid=torch.arange(3 * b, 3 * b + 3),
            caption_vid_list=vids_t2v[sl_], image_data=frames[vv], image_pad_mask=torch.zeros(3, 1, 32, 32, dtype=torch.bool),
            image_n_clips=[1] * 3, image_num_frames=[1] * 3, image_vid=torch.tensor(vv), image_tid_list=[v2t[v] for v in vv],
            dataset_type="val", dataset_name="toy_ret"))
    name, result = tr.evaluate_set(batches)
    assert name == "toy_ret"
    base = tr.model.model.module
    tr.model.eval()
    with torch.no_grad():
        text = base.forward_text_encoder(ids, mask)["pooled_output"]
        video = base.forward_img_encoder(frames, torch.zeros(4, 1, 32, 32, dtype=torch.bool), [1] * 4, [1] * 4)["clip_feature"]
        full = tr.model.model.reduce_clips(tr.model.model.get_l1_simi_matrix(text, video, 1), "l1")
    want = grr._cal_sym_recall(full, vids_t2v, v2t)
    assert result, result
    for k, v in want.items():
        assert abs(result["l1_simi_" + k] - v) < 1e-6, (k, result["l1_simi_" + k], v)
