"""End-to-end product model on the CPU lane emulator vs the reference's golden outputs (host logic + kernels)."""
import os
import subprocess

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
    # TEST SHORTCUT (lane emulator only): the TPM-CL head makes dozens of small fp32-accurate GEMMs and an emulated MFMA GEMM costs ~10 s a launch, so this
    # module swaps the head's GEMM for a host product; the split GEMM itself is exercised by tests/test_kernels_emu.py::test_tpmcl_ops_and_linear_f32
    # and by every -m gpu run.  The product module carries no such switch.
    import antmmf.hip.tpmcl as tpm

    real = tpm.matmul_f32

    def host_matmul(A, B, a_rmajor=False, b_rmajor=False):
        a = A.float().t() if a_rmajor else A.float()
        b = B.float() if b_rmajor else B.float().t()
        return a @ b

    tpm.matmul_f32 = host_matmul
    yield
    tpm.matmul_f32 = real
    if old is None:
        os.environ.pop("ANTMMF_HIP_LIB", None)
    else:
        os.environ["ANTMMF_HIP_LIB"] = old
    _lib.reset_for_tests()


def test_univl_stage1_vs_reference(golden):
    r = mc.case_univl_stage1(torch.device("cpu"), golden, "b4n1", 1)
    print(r)


SLOW = pytest.mark.skipif(not os.environ.get("ANTMMF_SLOW_TESTS"), reason="set ANTMMF_SLOW_TESTS=1 (each emulated e2e case takes 1-2 min)")


@SLOW
def test_univl_stage1_two_clips(golden):
    r = mc.case_univl_stage1(torch.device("cpu"), golden, "b3n2", 2)
    print(r)


@SLOW
def test_univl_stage1_activation_output_kept(golden):
    from antmmf.hip import functional

    functional.set_keep_ffn_norm(True)
    try:
        print(mc.case_univl_stage1(torch.device("cpu"), golden, "b4n1", 1))
    finally:
        functional.set_keep_ffn_norm(False)


@SLOW
def test_m2_towers_vs_reference(golden):
    print(mc.case_m2_towers(torch.device("cpu"), golden))


@SLOW
def test_univl_moco_vs_reference(golden):
    print(mc.case_univl_moco(torch.device("cpu"), golden))


@SLOW
def test_univl_moco_arena_ema(golden):
    print(mc.case_univl_moco(torch.device("cpu"), golden, with_optimizer=True))


@SLOW
def test_univl_stage2_vs_reference(golden):
    print(mc.case_univl_stage2(torch.device("cpu"), golden))


@SLOW
def test_univl_stage2_hard_mining_vs_oracle(golden):
    print(mc.case_univl_stage2(torch.device("cpu"), golden, mining=True))


@SLOW
def test_loss_contracts_over_seeded_batches_moco_and_dmae():
    """(2 batches each on the emulator: plumbing; the GPU suite runs 6)"""
    import subprocess
    import sys

    print(mc.case_univl_moco_loss_contract(torch.device("cpu"), k=2))
    env = dict(os.environ, ANTMMF_HIP_LIB=EMU_LIB, ANTMMF_ALLOW_EMULATOR="1")
    out = subprocess.run([sys.executable, "-c", mc.case_dmae_stage3_loss_contract("cpu", k=2)], capture_output=True, text=True, timeout=2400, env=env)
    assert "okdmaek" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


@SLOW
def test_univl_stage2_cnvid_scheduled_gate_vs_reference(golden):
    print(mc.case_univl_stage2_cnvid_gate(torch.device("cpu"), golden))


@SLOW
@pytest.mark.parametrize("loss_type", ["negNCE", "cross_entropy"])
def test_dmae_stage3_vs_reference(loss_type):
    import subprocess
    import sys

    code = mc.case_dmae_stage3() % (mc.ROOT, "cpu", loss_type, mc.TINY_CLIP_CFG, mc.DMAE_E2E)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, env=dict(os.environ))
    assert "okdmae" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    print(out.stdout[-400:])


@SLOW
def test_dmae_stage3_with_tpmcl_vs_reference():
    import subprocess
    import sys

    out = subprocess.run([sys.executable, "-c", mc.case_dmae_stage3_tpm("cpu")], capture_output=True, text=True, timeout=1500, env=dict(os.environ))
    assert "okdmae" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    print(out.stdout[-300:])


def test_dmae_tpmcl_vs_reference(golden):
    """Loss type 4 (all three margin terms) by default; 2 and 3 with ANTMMF_SLOW_TESTS (about 80 s of emulated row kernels each -- the
    hardware suite runs all three)."""
    ptypes = (2, 3, 4) if os.environ.get("ANTMMF_SLOW_TESTS") else (4,)
    print(mc.case_dmae_tpmcl(torch.device("cpu"), golden, ptypes))


@SLOW
def test_dmae_tpmcl_batched_blocks_equal_block_loop():
    print(mc.case_dmae_tpmcl_blocks(torch.device("cpu")))


def test_dmae_wti_vs_reference(golden):
    print(mc.case_dmae_wti(torch.device("cpu"), golden))


def test_bert_layer_dropout_vs_oracle_same_masks():
    print(mc.case_bert_layer_dropout(torch.device("cpu")))


@pytest.mark.parametrize("head_size", [64, 128])
def test_vilbert_biattention_vs_reference(golden, head_size):
    print(mc.case_vilbert_biattention(torch.device("cpu"), golden, head_size))


def test_temporal_head_vs_oracle_and_reference(golden):
    print(mc.case_temporal_head(torch.device("cpu"), golden))


def test_dmae_seqtransf_vs_reference(golden):
    print(mc.case_dmae_seqtransf(torch.device("cpu"), golden))


@SLOW
def test_m2_itc_step_vs_oracle():
    print(mc.case_m2_itc_vs_oracle(torch.device("cpu")))


@SLOW
def test_m2_sub_ln_fold(golden):
    """The optional sub-LN fold of the M2 layers (gelu -> ffn_layernorm inside the GEMM epilogues; functional.set_ffn_fold): reference goldens for the
    towers at the usual gates."""
    from antmmf.hip import functional

    functional.set_keep_ffn_norm(True)
    functional.set_ffn_fold(True)
    try:
        print(mc.case_m2_towers(torch.device("cpu"), golden))
    finally:
        functional.set_ffn_fold(False)
        functional.set_keep_ffn_norm(False)
