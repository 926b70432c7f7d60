"""SURVEY.md 8(f1) on the device: Checkpoint save / resume of the flat-arena optimizer on a real MI355X (the emulator-side twin is
tests/test_checkpoint.py): snapshot at iteration 2, resume with the optimizer state, finish -- masters, moments and the bf16 compute shadow
equal the uninterrupted run bit for bit; a reference-format torch AdamW state loads into the flat moments."""
import os

import pytest
import torch

import test_checkpoint as tc

pytestmark = pytest.mark.gpu


def test_bit_exact_resume_on_gpu(tmp_path):
    from antmmf.hip import _lib

    os.environ.pop("ANTMMF_HIP_LIB", None)
    _lib.reset_for_tests()
    assert _lib.backend() == 1
    Trainer = tc._toy()
    batches = [b.to(torch.device("cuda:0")) for b in tc._batches()]
    full = Trainer(tc._cfg(tmp_path / "a", device="cuda"), batches)
    full.load()
    full.train()
    folder = tmp_path / "a" / "toy_task_toy_ckpt_7"
    assert sorted(os.listdir(folder / "models")) == ["model_2.ckpt", "model_4.ckpt"]
    resumed = Trainer(tc._cfg(tmp_path / "b", device="cuda", resume_file=str(folder / "models" / "model_2.ckpt")), batches[2:])
    resumed.load()
    assert resumed.current_iteration == 2 and resumed.optimizer._step == 2 and resumed.arena.master.is_cuda
    for p in resumed.model.parameters():
        assert torch.equal(p._antmmf_bf16, p.data.to(torch.bfloat16))
    resumed.train()
    assert torch.equal(resumed.arena.master, full.arena.master) and torch.equal(resumed.arena.shadow, full.arena.shadow)
    assert torch.equal(resumed.optimizer.exp_avg, full.optimizer.exp_avg) and torch.equal(resumed.optimizer.exp_avg_sq, full.optimizer.exp_avg_sq)
    # a reference-format optimizer state (torch.optim.AdamW) scatters into the flat moments
    ref_opt = torch.optim.AdamW(full.model.parameters(), lr=0.05, weight_decay=0.01)
    full.model(batches[0])["losses"]["toy_loss"].backward()
    ref_opt.step()
    fresh = Trainer(tc._cfg(tmp_path / "c", device="cuda", save_dir=None, snapshot_interval=None), batches)
    fresh.load()
    fresh.optimizer.load_state_dict(ref_opt.state_dict())
    assert fresh.optimizer._step == 1 and float(fresh.optimizer.exp_avg.abs().sum()) > 0
