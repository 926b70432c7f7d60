"""antmmf.common.checkpoint (reference antmmf/common/checkpoint.py:79-356): file layout, reference-style key
handling, pretrained_mapping, and bit-exact resume of the flat-arena optimizer.  Runs the fused AdamW / cast kernels on the CPU lane
emulator."""
import os
import subprocess
import warnings

import pytest
import torch

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


def _toy():
    from antmmf.common.registry import registry
    from antmmf.models.base_model import BaseModel
    from antmmf.optimizer import build_optimizer
    from antmmf.trainers.base_trainer import BaseTrainer

    @registry.register_model("toy_ckpt")
    class Toy(BaseModel):
        def build(self):
            self.enc_a = torch.nn.Linear(8, 8)
            self.enc_b = torch.nn.Linear(8, 8)
            self.norm = torch.nn.LayerNorm(8)

        def forward(self, sample_list):
            y = self.norm(self.enc_b(torch.tanh(self.enc_a(sample_list["image_data"]))))
            return {"losses": {"toy_loss": ((y - sample_list["caption_target"]) ** 2).mean()}}

    class ArenaTrainer(BaseTrainer):  # CPU tensors, but the MI355X optimizer path: flat arena + fused AdamW (emulated)
        def load_optimizer(self):
            self.optimizer = build_optimizer(self.model, self.config, use_hip_arena=True)
            self.arena = self.optimizer.arena
            self.lr_scheduler = None

    return ArenaTrainer


def _cfg(tmp, **tp):
    from antmmf.common.configuration import Configuration

    base = {"trainer": "base_trainer", "device": "cpu", "max_iterations": 4, "log_interval": 100, "seed": 7, "save_dir": str(tmp),
            "snapshot_interval": 2}
    base.update(tp)
    return Configuration({"training_parameters": base, "task_attributes": {"toy_task": {}},
                          "optimizer_attributes": {"type": "AdamW", "params": {"lr": 0.05, "weight_decay": 0.01}},
                          "model_attributes": {"toy_ckpt": {}}})


def _batches():
    from antmmf.structures.sample import SampleList

    g = torch.Generator().manual_seed(11)
    return [SampleList(image_data=torch.randn(6, 8, generator=g), caption_target=torch.randn(6, 8, generator=g)) for _ in range(4)]


def test_layout_and_bit_exact_resume(tmp_path):
    Trainer = _toy()
    batches = _batches()
    full = Trainer(_cfg(tmp_path / "a"), batches)
    full.load()
    full.train()
    folder = tmp_path / "a" / "toy_task_toy_ckpt_7"
    assert (folder / "config.yaml").is_file() and (folder / "toy_ckpt_final.pth").is_file()
    assert sorted(os.listdir(folder / "models")) == ["model_2.ckpt", "model_4.ckpt"]
    ck = torch.load(folder / "models" / "model_2.ckpt", weights_only=False)
    assert set(ck) >= {"model", "optimizer", "current_iteration", "current_epoch", "best_iteration", "best_metric_value"}
    assert ck["current_iteration"] == 2 and set(ck["model"]) == set(full.model.state_dict())
    final = torch.load(folder / "toy_ckpt_final.pth", weights_only=False)
    for k, v in full.model.state_dict().items():
        assert torch.equal(final[k], v)

    # resume from iteration 2 with the optimizer state, run the remaining batches: identical weights and moments, bit for bit
    resumed = Trainer(_cfg(tmp_path / "b", resume_file=str(folder / "models" / "model_2.ckpt")), batches[2:])
    resumed.load()
    assert resumed.current_iteration == 2 and resumed.optimizer._step == 2
    for p in resumed.model.parameters():  # the bf16 compute shadow follows the loaded masters
        assert torch.equal(p._antmmf_bf16, p.data.to(torch.bfloat16))
    resumed.train()
    assert resumed.current_iteration == 4
    assert torch.equal(resumed.arena.master, full.arena.master)
    assert torch.equal(resumed.optimizer.exp_avg, full.optimizer.exp_avg) and torch.equal(resumed.optimizer.exp_avg_sq, full.optimizer.exp_avg_sq)

    # restart: weights only, iteration counter and moments start over
    fresh = Trainer(_cfg(tmp_path / "c", resume_file=str(folder / "models" / "model_2.ckpt"), restart=True, max_ckpt_num=1), batches)
    fresh.load()
    assert fresh.current_iteration == 0 and fresh.optimizer._step == 0 and float(fresh.optimizer.exp_avg.abs().sum()) == 0.0
    assert torch.equal(fresh.model.enc_a.weight, ck["model"]["enc_a.weight"])
    fresh.train()
    assert os.listdir(tmp_path / "c" / "toy_task_toy_ckpt_7" / "models") == ["model_4.ckpt"]  # max_ckpt_num prunes the older snapshot


def test_reference_style_keys_and_pretrained_mapping(tmp_path):
    Trainer = _toy()
    tr = Trainer(_cfg(tmp_path / "m", load_pretrained=True, pretrained_mapping={"enc_a": "enc_b"}), _batches())
    tr.load()
    g = torch.Generator().manual_seed(5)
    src = {
        "module.enc_a.weight": torch.randn(8, 8, generator=g),   # written by a DDP-wrapped reference run
        "module.enc_a.bias": torch.randn(8, generator=g),
        "module.norm.weight": torch.randn(9, generator=g),       # shape mismatch: skipped
        "module.head.weight": torch.randn(2, 2, generator=g),    # not in the model: skipped
    }
    path = tmp_path / "ref_style.ckpt"
    torch.save({"model": src}, path)
    before_norm = tr.model.norm.weight.detach().clone()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tr.checkpoint.load_model_weights(str(path))
    msgs = " ".join(str(x.message) for x in w)
    assert "module.head.weight" not in msgs and "head.weight" in msgs and "norm.weight" in msgs
    assert torch.equal(tr.model.enc_a.weight, src["module.enc_a.weight"]) and torch.equal(tr.model.enc_a.bias, src["module.enc_a.bias"])
    assert torch.equal(tr.model.norm.weight, before_norm)
    # pretrained_mapping {"enc_a": "enc_b"}: the enc_a sub-tree of the checkpoint is also copied onto enc_b
    assert torch.equal(tr.model.enc_b.weight, src["module.enc_a.weight"]) and torch.equal(tr.model.enc_b.bias, src["module.enc_a.bias"])
    assert torch.equal(tr.model.enc_b.weight._antmmf_bf16, src["module.enc_a.weight"].to(torch.bfloat16))
    # a bare state_dict (the *_final.pth form) loads the same way; force=True ignores the mapping
    torch.save({"enc_a.weight": torch.zeros(8, 8)}, tmp_path / "bare.pth")
    tr.checkpoint.load_model_weights(str(tmp_path / "bare.pth"), force=True)
    assert float(tr.model.enc_a.weight.abs().sum()) == 0.0 and torch.equal(tr.model.enc_b.weight, src["module.enc_a.weight"])
    with pytest.raises(RuntimeError):
        bad = Trainer(_cfg(tmp_path / "x", resume_file=str(tmp_path / "missing.ckpt")), _batches())
        bad.load()


def test_reference_parameter_names_exist_in_this_build(golden):
    """State-dict compatibility (8(f1)): every trainable parameter name the REFERENCE models expose (recorded by
    tests/golden/make_golden.py next to the gradient norms) is a key of this build's state_dict for the same configuration."""
    import model_cases as mc

    dev = torch.device("cpu")
    g = golden("e2e_clip_arch.pt")
    ref = {k.split(".gnorm.", 1)[1] for k in g if ".gnorm." in k}
    own = set(mc.build_tiny_univl(dev).state_dict())
    assert len(ref) > 50 and ref <= own, sorted(ref - own)[:5]
    g = golden("e2e_clip_moco.pt")
    ref = {k.split(".gnorm1.", 1)[1] for k in g if ".gnorm1." in k}
    assert len(ref) > 50 and ref <= own, sorted(ref - own)[:5]
    g = golden("e2e_m2.pt")
    ref = {k.split("gnorm.", 1)[1] for k in g if k.startswith("gnorm.")}
    own = set(mc.build_tiny_m2(dev).state_dict())
    assert len(ref) > 50 and ref <= own, sorted(ref - own)[:5]


def test_shadow_follows_in_place_weight_writes(tmp_path):
    """the bf16 compute shadow used by the forward pass must follow writes to the fp32 masters that do not come from the
    fused optimizer -- model.load_state_dict, an initialiser, an in-place clamp (detected through torch's version counter) -- and
    `sync_shadow()` covers `.data` edits."""
    from antmmf.hip import functional as F

    tr = _toy()(_cfg(tmp_path, save_dir=None, snapshot_interval=None))
    tr.load()
    p = tr.model.enc_a.weight
    torch.testing.assert_close(F.compute_copy(p).float(), p.detach().bfloat16().float())
    sd = {k: v.clone() + 1.0 for k, v in tr.model.state_dict().items()}
    tr.model.load_state_dict(sd)
    torch.testing.assert_close(p.detach(), sd["enc_a.weight"])
    torch.testing.assert_close(F.compute_copy(p).float(), sd["enc_a.weight"].bfloat16().float())
    v0 = F._T_CACHE_VERSION[0]
    with torch.no_grad():
        p.clamp_(-0.25, 0.25)
    assert float(F.compute_copy(p).float().abs().max()) <= 0.25 and F._T_CACHE_VERSION[0] > v0
    p.data.mul_(2.0)                       # invisible to the version counter ...
    tr.arena.sync_shadow()                 # ... documented remedy
    torch.testing.assert_close(F.compute_copy(p).float(), p.detach().bfloat16().float())


def test_hip_adamw_loads_reference_torch_adamw_state(tmp_path):
    """resuming from a reference-format optimizer state (torch.optim.AdamW: per-parameter exp_avg / exp_avg_sq / step) restores
    the flat moments and the step count; the next fused step then equals torch's next step."""
    Trainer = _toy()
    tr = Trainer(_cfg(tmp_path, save_dir=None, snapshot_interval=None))
    tr.load()
    batches = _batches()
    ref_model = type(tr.model)(tr.model.config)
    ref_model.build()
    ref_model.load_state_dict({k: v.clone() for k, v in tr.model.state_dict().items()})
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=0.05, weight_decay=0.01)
    for b in batches[:2]:
        ref_opt.zero_grad()
        ref_model(b)["losses"]["toy_loss"].backward()
        ref_opt.step()
    state = ref_opt.state_dict()
    snapshot = {k: (v if not isinstance(v, dict) else dict(v)) for k, v in state.items()}
    tr.model.load_state_dict({k: v.clone() for k, v in ref_model.state_dict().items()})
    tr.optimizer.load_state_dict(state)
    assert set(state.keys()) == set(snapshot.keys()) and "state" in state and len(state["state"]) == len(snapshot["state"])  # caller's dict untouched
    assert tr.optimizer._step == 2 and float(tr.optimizer.exp_avg.abs().sum()) > 0
    ref_opt.zero_grad()
    ref_model(batches[2])["losses"]["toy_loss"].backward()
    ref_opt.step()
    tr.optimizer.zero_grad()
    tr.model(batches[2])["losses"]["toy_loss"].backward()
    tr.optimizer.step()
    for (n, a), (_, b) in zip(tr.model.named_parameters(), ref_model.named_parameters()):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=2e-3, atol=2e-4, msg=n)
    with pytest.raises(ValueError):
        tr.optimizer.load_state_dict({"state": {0: dict(step=torch.tensor(1.0), exp_avg=torch.zeros(3), exp_avg_sq=torch.zeros(3))},
                                      "param_groups": tr.optimizer.state_dict()["param_groups"]})


def test_optimizer_step_refreshes_transposed_weight_copies_in_one_launch():
    """The transposed bf16 weight copies dgrad uses (compute_copy_t) are re-made by HipAdamW.step() for every arena parameter that asked for one --
    a single batched launch out of the fresh shadow -- and are exactly shadow.t(); an out-of-band write still invalidates them."""
    from antmmf.hip import functional as F
    from antmmf.hip.arena import HipAdamW

    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(70, 130, generator=g)), torch.nn.Parameter(torch.randn(64, 64, generator=g)),
          torch.nn.Parameter(torch.randn(130, generator=g)), torch.nn.Parameter(torch.randn(9, 200, generator=g))]
    opt = HipAdamW([{"params": ps}], lr=1e-2, weight_decay=0.01)
    for p in (ps[0], ps[1], ps[3]):
        t = F.compute_copy_t(p)                                   # first use: per-parameter launch, registers the parameter
        assert torch.equal(t, F.compute_copy(p).t())
    for step in range(2):
        for p in ps:
            p.grad.copy_(torch.randn(p.shape, generator=g))
        opt.step()
        for p in (ps[0], ps[1], ps[3]):
            ver, t = p._antmmf_bf16_t
            assert ver == F._T_CACHE_VERSION[0] and t.shape == (p.shape[1], p.shape[0])
            assert torch.equal(t, F.compute_copy(p).t()), step     # the batched launch wrote shadow.t()
            assert F.compute_copy_t(p) is t                        # and the backward pass finds it: no launch of its own
    with torch.no_grad():
        ps[0].clamp_(-0.1, 0.1)                                    # a write the optimizer did not make
    t = F.compute_copy_t(ps[0])
    assert float(t.float().abs().max()) <= 0.1 + 1e-3 and torch.equal(t, F.compute_copy(ps[0]).t())


@pytest.mark.parametrize("prefix", ["module.", ""])
def test_cn_clip_load_pretrained_round_trip(tmp_path, prefix):
    """8(f1): the CN-CLIP tower loaders (reference clip_visual_encoder.py:46-71, clip_text_encoder.py:194-227).  A released checkpoint is
    {"state_dict": {"module.visual.<k>", "module.bert.<k>", "module.text_projection", "module.logit_scale", ...}}: the image tower keeps the
    `visual.*` entries, the text tower the `bert.*` entries + `text_projection`; everything else is ignored.  Both spellings are accepted
    here ("module."-prefixed as released, and bare -- the reference's own bare-key branch mangles `text_projection`, :213, so only the
    prefixed form is pinned by it).  After the load every parameter equals the checkpoint tensor, the bf16 compute shadow follows (the
    forward changes), and a wrong-shaped projection is skipped as the reference does (:222-227)."""
    import model_cases as mc   # (puts prj/base_vtp on sys.path)
    import roi_univl  # noqa: F401
    from roi_univl.univl.model.clip_text_encoder import RobertBertEncoder
    from roi_univl.univl.model.clip_visual_encoder import VitImageEncoder

    vp = dict(mc.TINY_CLIP_CFG["image_encoder"]["params"])
    tparams = dict(mc.TINY_CLIP_CFG["text_encoder"]["params"])
    src_v, src_t = VitImageEncoder(**vp), RobertBertEncoder(**tparams)
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, v in src_v.visual.state_dict().items():
        sd[prefix + "visual." + k] = torch.randn(v.shape, generator=g) * 0.05 if v.is_floating_point() else v.clone()
    for k, v in src_t.module.state_dict().items():
        sd[prefix + "bert." + k] = torch.randn(v.shape, generator=g) * 0.05 if v.is_floating_point() else v.clone()
    sd[prefix + "text_projection"] = torch.randn(src_t.text_projection.shape, generator=g) * 0.05
    sd[prefix + "logit_scale"] = torch.tensor(2.5)          # not a tower key: ignored by both loaders
    path = str(tmp_path / "cn_clip.pt")
    torch.save({"state_dict": sd, "epoch": 3}, path)

    vis = VitImageEncoder(**dict(vp, pretrained=True, model_name=path))
    for k, v in vis.visual.state_dict().items():
        torch.testing.assert_close(v, sd[prefix + "visual." + k], rtol=0, atol=0, msg=k)
    txt = RobertBertEncoder(**dict(tparams, pretrained=True, model_name=path))
    for k, v in txt.module.state_dict().items():
        torch.testing.assert_close(v, sd[prefix + "bert." + k], rtol=0, atol=0, msg=k)
    torch.testing.assert_close(txt.text_projection.data, sd[prefix + "text_projection"], rtol=0, atol=0)

    # a projection of another width is left alone (reference :222-227), a missing file is an error, not a download attempt
    sd_bad = dict(sd)
    sd_bad[prefix + "text_projection"] = torch.zeros(src_t.text_projection.shape[0], 7)
    torch.save({"state_dict": sd_bad}, path)
    keep = RobertBertEncoder(**dict(tparams, pretrained=True, model_name=path))
    assert keep.text_projection.shape == src_t.text_projection.shape and float(keep.text_projection.abs().sum()) > 0
    with pytest.raises(RuntimeError):
        VitImageEncoder(**dict(vp, pretrained=True, model_name=str(tmp_path / "absent.pt")))

    # the loaded weights are the ones the kernels compute with: forward of the loaded tower == forward of a tower filled by hand
    ref = VitImageEncoder(**vp)
    ref.visual.load_state_dict({k: sd[prefix + "visual." + k] for k in ref.visual.state_dict()})
    img = torch.randn(2, 1, 3, 32, 32, generator=g)
    m = torch.zeros(2, 1, 32, 32, dtype=torch.bool)
    a, b = vis(img, m)["grid_feature"], ref(img, m)["grid_feature"]
    torch.testing.assert_close(a.float(), b.float(), rtol=0, atol=0)


def test_early_stop_state_is_persisted_and_best_weights_are_final(tmp_path):
    """best.ckpt records the early-stopping state (`best_iteration`, `best_metric_value`; reference checkpoint.py:320-356 reads it
    from trainer.early_stopping), a resumed run gets it back (init_from_checkpoint), and after an early stop `<model>_final.pth` holds the
    BEST weights (checkpoint.restore() before finalize(), reference early_stopping.py:79-83).  The learning rate is negative-ish large so the
    validation loss gets worse after the first evaluation."""
    Trainer = _toy()
    cfg = _cfg(tmp_path / "es", evaluation_interval=1, should_early_stop=True, patience=1, monitored_metric="total_loss", max_iterations=8,
               snapshot_interval=100)
    cfg.optimizer_attributes.params.lr = 5.0   # diverges: the first evaluation is the best one
    batches = _batches() + _batches()
    tr = Trainer(cfg, batches)
    tr.load()
    tr.load_task(batches, _batches()[:2])
    tr.train()
    assert tr.early_stopping.activated and tr.best_iteration >= 1 and tr.current_iteration == tr.best_iteration + 2, (tr.best_iteration, tr.current_iteration)
    folder = tmp_path / "es" / "toy_task_toy_ckpt_7"
    best = torch.load(folder / "toy_ckpt_best.ckpt" if (folder / "toy_ckpt_best.ckpt").is_file() else next(folder.glob("*best.ckpt")), weights_only=False)
    assert best["best_iteration"] == tr.best_iteration and abs(best["best_metric_value"] - tr.best_monitored) < 1e-12
    final = torch.load(next(folder.glob("*_final.pth")), weights_only=False)
    for k, v in best["model"].items():
        assert torch.equal(final[k], v), k      # the restored best weights, not the diverged last ones
    # resume: patience keeps counting from the recorded best
    cfg2 = _cfg(tmp_path / "es2", resume_file=str(next(folder.glob("*best.ckpt"))), evaluation_interval=1, should_early_stop=True, patience=1)
    tr2 = Trainer(cfg2, batches)
    tr2.load()
    assert tr2.best_iteration == tr.best_iteration and abs(tr2.best_monitored - tr.best_monitored) < 1e-12


def test_arena_pack_adjacency_views_and_legacy_optimizer_state():
    """Round 6: parameters tagged with arena.tag_pack (the separate q / k / v projections) are laid out adjacently, in tag order, whatever their order in the module -- the
    packed [3d, d] operand and [3d] bias of the fused layer are then VIEWS of the arena (equal to the concatenation, no copy); everything else keeps its place.  The flat
    optimizer moments carry their layout: a state written by a build WITHOUT pack reordering (no layout entry) is remapped parameter by parameter."""
    from antmmf.hip import functional as HF
    from antmmf.hip.arena import HipAdamW, tag_pack

    torch.manual_seed(3)
    d = 64
    k, v, q, o = (torch.nn.Linear(d, d) for _ in range(4))       # torchscale's order: k, v, q (multihead_attention.py:66-71)
    tag_pack(q.weight, k.weight, v.weight)
    tag_pack(q.bias, k.bias, v.bias)
    decay, no_decay = [k.weight, v.weight, o.weight, q.weight], [k.bias, v.bias, o.bias, q.bias]
    opt = HipAdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-2)
    P = dict(wq=q.weight, wk=k.weight, wv=v.weight, bq=q.bias, bk=k.bias, bv=v.bias)
    assert HF._arena_run([q.weight, k.weight, v.weight]) is not None and HF._arena_run([q.bias, k.bias, v.bias]) is not None
    assert o.weight._antmmf_offset == 3 * d * d                      # the pack sits where its first member (k) was, the untagged parameter follows
    spec = HF.LayerSpec(kind="m2", heads=1, eps=1e-5, act="gelu", packed_qkv=False)
    w, b = HF._packed_qkv_weight(P, spec)
    assert w.data_ptr() == opt.arena.shadow.data_ptr() + 2 * q.weight._antmmf_offset and b.data_ptr() == opt.arena.master.data_ptr() + 4 * q.bias._antmmf_offset
    torch.testing.assert_close(w.float(), torch.cat([q.weight, k.weight, v.weight]).detach().bfloat16().float(), rtol=0, atol=0)
    torch.testing.assert_close(b, torch.cat([q.bias, k.bias, v.bias]).detach(), rtol=0, atol=0)
    torch.testing.assert_close(HF._packed_qkv_weight_t(P, spec).float(), w.float().t(), rtol=0, atol=0)
    # a legacy flat state: moments laid out in plain parameter order (k, v, o, q | k, v, o, q), recognisable values per parameter
    legacy = torch.zeros_like(opt.exp_avg)
    off = 0
    for i, p in enumerate(decay + no_decay):
        legacy[off:off + p.numel()] = float(i + 1)
        off += (p.numel() + 63) // 64 * 64
    sd = opt.state_dict()
    sd["antmmf_arena"] = dict(step=5, exp_avg=legacy, exp_avg_sq=legacy * 2)      # no "layout": written before round 6
    opt.load_state_dict(sd)
    for i, p in enumerate(decay + no_decay):
        sl = opt.exp_avg[p._antmmf_offset:p._antmmf_offset + p.numel()]
        assert float(sl.min()) == float(sl.max()) == float(i + 1), (i, float(sl.min()), float(sl.max()))
    assert opt._step == 5
    # its own state (with the layout) round-trips unchanged
    import copy

    sd2 = copy.deepcopy(opt.state_dict())      # (the state dict holds the live moment tensors)
    before = opt.exp_avg.clone()
    opt.exp_avg.zero_()
    opt.load_state_dict(sd2)
    torch.testing.assert_close(opt.exp_avg, before, rtol=0, atol=0)
