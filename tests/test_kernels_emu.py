"""Run the kernel parity cases against the CPU lane emulation of the HIP kernels (tests/emu).

This validates the kernels' index arithmetic (LDS layouts, MFMA fragment bookkeeping, guards,
epilogues) in the GPU-less build container; hardware parity is tests/test_kernels_gpu.py."""
import os
import subprocess

import pytest
import torch

import kernel_cases as kc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "build", "libantmmf_emu.so")
CSRC = os.path.join(ROOT, "ant-multi-modal-framework_amd", "csrc")


def _stale():
    if not os.path.isfile(EMU_LIB):
        return True
    t = os.path.getmtime(EMU_LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "tests", "emu", "hip_emu.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


@pytest.fixture(scope="module")
def ops():
    if not os.path.isfile("/opt/rocm/lib/llvm/bin/clang++") and not os.environ.get("EMU_CXX"):
        pytest.skip("no clang++ to build the emulator")
    if _stale():
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    from antmmf.hip import _lib

    old = os.environ.get("ANTMMF_HIP_LIB")
    os.environ["ANTMMF_HIP_LIB"] = EMU_LIB
    _lib.reset_for_tests()
    from antmmf.hip import ops as hops

    assert _lib.backend() == 0
    yield hops
    if old is None:
        os.environ.pop("ANTMMF_HIP_LIB", None)
    else:
        os.environ["ANTMMF_HIP_LIB"] = old
    _lib.reset_for_tests()


DEV = torch.device("cpu")


def test_layernorm(ops):
    kc.case_layernorm(ops, DEV, torch.float32)
    kc.case_layernorm(ops, DEV, torch.bfloat16)
    kc.case_layernorm(ops, DEV, torch.float32, rows=5, cols=1024 + 512, eps=1e-12)
    kc.case_layernorm(ops, DEV, torch.bfloat16, rows=261, cols=1024)
    os.environ["ANTMMF_LN_FWD_ADJ"] = "1"    # lab-only variant (two adjacent rows per wave; measured slower on MI355X, kept for the A/B): odd row count, the last wave holds one row
    try:
        kc.case_layernorm(ops, DEV, torch.bfloat16, rows=261, cols=1024)
        kc.case_layernorm(ops, DEV, torch.float32, rows=258, cols=768)
    finally:
        os.environ.pop("ANTMMF_LN_FWD_ADJ", None)


def test_act_layernorm(ops):
    kc.case_act_layernorm(ops, DEV, torch.float32, rows=9, cols=512, act="gelu")
    kc.case_act_layernorm(ops, DEV, torch.bfloat16, rows=5, cols=2048 + 512, act="gelu")   # workgroup-per-row path, 2 vectors / thread
    kc.case_act_layernorm(ops, DEV, torch.float32, rows=6, cols=1536, act="quick_gelu")  # workgroup-per-row path, 1 vector / thread
    kc.case_layernorm(ops, DEV, torch.float32, rows=7, cols=4096)


def test_ffn_fold_elementwise(ops):
    """Sub-LN fold through the generic kernel's epilogues + the row-statistics / column-sum passes (shapes the persistent kernel does not take)."""
    kc.case_ffn_fold(ops, DEV, tokens=40, d=64, ff=192)
    kc.case_ffn_fold(ops, DEV, tokens=9, d=128, ff=512, res_scale=4.0, seed=340)
    kc.case_ffn_fold(ops, DEV, tokens=17, d=64, ff=128, act="quick_gelu", seed=380)


def test_ffn_fold_persistent_kernel():
    """The same operator on the BK = 64 persistent kernel's register-level epilogues (row partials of fc1, row-affine fc2, LayerNorm-backward dgrad with the
    bias-gradient partials): 256-aligned shapes, an 8-workgroup grid so that workgroups walk several tiles."""
    import subprocess
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'k';"
            "os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '8';"
            "import ctypes; import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cpu');"
            "kc.case_ffn_fold(ops, dev, tokens=512, d=256, ff=512, seed=420);"
            "lib = ctypes.CDLL(os.environ['ANTMMF_HIP_LIB']); lib.antmmf_debug_gemm_k64_launches.restype = ctypes.c_long;"
            "assert lib.antmmf_debug_gemm_k64_launches() >= 3, lib.antmmf_debug_gemm_k64_launches(); print('okffn')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert "okffn" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_activations(ops):
    kc.case_activations(ops, DEV)


def test_l2norm(ops):
    kc.case_l2norm(ops, DEV)


def test_movers(ops):
    kc.case_movers(ops, DEV)


def test_adamw(ops):
    kc.case_adamw(ops, DEV)


def test_gemm(ops):
    kc.case_gemm(ops, DEV)


def test_gemm_multitile(ops):
    kc.case_gemm_multitile(ops, DEV)


@pytest.mark.parametrize("variant", ["256", "ring"])
def test_gemm_large_tile_variants(ops, variant):
    """The 256 x 256 kernels (2-buffer DMA; 4-stage DMA ring -- normally chosen for >= 512 tiles) forced onto a
    2 x 2-tile problem with ragged edges."""
    import subprocess
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = %r;"
            "import kernel_cases as kc; from antmmf.hip import ops; kc.case_gemm_multitile(ops, torch.device('cpu')); print('okbig')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB, variant))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert "okbig" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("cont", ["1", pytest.param("0", marks=pytest.mark.skipif(not os.environ.get("ANTMMF_SLOW_TESTS"), reason="A/B variant: set ANTMMF_SLOW_TESTS=1"))])
def test_gemm_persistent_ring(cont):
    """The persistent 256 x 256 ring kernel on 3 x 3- and 3 x 5-tile problems with an 8-workgroup grid: workgroups walk two tiles
    (cont = 1: the DMA ring runs on across the tile boundary; 0: next-tile prologue issued as a burst before the epilogue; the
    epilogue is staged through the free ring slot either way)."""
    import subprocess
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'p';"
            "os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '8';"
            "os.environ['ANTMMF_GEMM_CONT'] = %r;"
            "import kernel_cases as kc; from antmmf.hip import ops; slow = bool(os.environ.get('ANTMMF_SLOW_TESTS'));"
            "kc.case_gemm_persistent(ops, torch.device('cpu'), I=520, J=600, R=128, quick=not slow);"
            "slow and kc.case_gemm_persistent(ops, torch.device('cpu'), I=700, J=1100, R=192); print('okpersist')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB, cont))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert "okpersist" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("variant", ["4", "134217732", pytest.param("134217748", marks=pytest.mark.skipif(not os.environ.get("ANTMMF_SLOW_TESTS"), reason="A/B variant: set ANTMMF_SLOW_TESTS=1"))])
def test_gemm_k64_persistent(variant):
    """The BK = 64 quarter-phase persistent kernel (default for the large GEMMs) on 2 - 16-tile problems with an 8-workgroup grid, so that
    workgroups walk two tiles and the DMA ring crosses the tile boundary: nk = 2 (no steady state), 3 and 5; every epilogue (staged for
    plain / bias; register-level lane swap for residual / generic).  variant 4 = the dispatch of both libraries (the generic run-time epilogue and R = 128 go to the other
    kernel families: 10 of the 13 calls are k64 launches); + bit 27: the burst-epilogue kernel's own forms for those shapes (lab A/B only: all 13); + bit 4: the 1 / 3 / 3 / 1 DMA distribution."""
    import subprocess
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'k';"
            "os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '8'; os.environ['ANTMMF_GEMM_VARIANT'] = %r;"
            "import ctypes; import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cpu');"
            "kc.case_gemm_k64(ops, dev, I=512, J=256, R=192);"
            "kc.case_gemm_k64(ops, dev, I=1024, J=1024, R=128, quick=True);"
            "kc.case_gemm_k64(ops, dev, I=768, J=768, R=320, quick=True);"
            "lib = ctypes.CDLL(os.environ['ANTMMF_HIP_LIB']); lib.antmmf_debug_gemm_k64_launches.restype = ctypes.c_long;"
            "assert lib.antmmf_debug_gemm_k64_launches() >= %d, lib.antmmf_debug_gemm_k64_launches(); print('okk64')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB, variant, 10 if variant == "4" else 13))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert "okk64" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gemm_k64_rolling_epilogue():
    """gemm_nt_k64r_kernel (rolling epilogue: bias / residual start the accumulators, row quarters stored inside the last K-tile) on 2 - 16-tile problems
    with an 8-workgroup grid (workgroups walk up to three tiles: quarter 3 of a tile leaves in the next tile's first K-tile, the next tile's residual is
    requested during the last K-tile of the previous one): nk = 3 (one K-tile per role), 4, 5; plain / bias / residual / bias + residual.  The
    kernel must really have run: for bias + residual a few results differ from the burst-epilogue kernel in the last bf16 bit (order of the fp32 sum)."""
    import subprocess
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'k';"
            "os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '8'; os.environ['ANTMMF_GEMM_VARIANT'] = '8196';"
            "import ctypes; import kernel_cases as kc; from antmmf.hip import ops, _lib; dev = torch.device('cpu');"
            "kc.case_gemm_k64(ops, dev, I=512, J=256, R=192, quick=True);"
            "kc.case_gemm_k64(ops, dev, I=2560, J=256, R=256, quick=True);"
            "g = torch.Generator().manual_seed(3); X = torch.randn(768, 320, generator=g).bfloat16(); W = (torch.randn(512, 320, generator=g) * 0.05).bfloat16();"
            "b = torch.randn(512, generator=g); r = (torch.randn(768, 512, generator=g) * 30).bfloat16(); ref = X.float() @ W.float().t();"
            "lib = _lib.load(); y1 = ops.gemm(X, W, bias=b, residual=r); p1 = ops.gemm(X, W);"
            "kc.check('k64r.bias', ops.gemm(X, W, bias=b), ref + b, 2e-2, 1e-2); kc.check('k64r.res', ops.gemm(X, W, residual=r), ref + r.float(), 2e-2, 1e-2);"
            "lib.antmmf_debug_set_gemm_variant(4 | 16384); y0 = ops.gemm(X, W, bias=b, residual=r); p0 = ops.gemm(X, W);"
            "assert torch.equal(p0, p1); d = (y0.float() - y1.float()).abs(); assert 0 < int((d > 0).sum()) < 400 and float(d.max()) <= 0.5, (int((d > 0).sum()), float(d.max()));"
            "print('okk64r')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert "okk64r" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gemm_k64_tail_round_cells():
    """The tail round of the persistent NT kernel as CELLS inside the same launch (gemm_nt_k64r_kernel): 40 tiles on a 16-workgroup grid = two full rounds of the tile walk
    + ONE leftover tile per XCD chunk, whose 16 cells (one phase x one Q fragment per wave, full K range) go to the chunk's two workgroups, eight cells each (variant bit
    26 lifts the product's one-cell-per-workgroup rule for this small grid).  K = 256 (4 K-tiles: shorter than the cell ring's depth) and K = 768 (12: the 8-slot ring
    wraps).  Against the fp32 product on every tile, and BIT-IDENTICAL to the same call with the cells disabled (variant bit 25: the leftover tiles go through the walk):
    the cells start their accumulators from bias + residual and add the K-tiles in the walk's order."""
    import sys

    code = ("import os, sys, ctypes, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'k';"
            "os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '16'; os.environ['ANTMMF_GEMM_VARIANT'] = str(4 | (1 << 26));"
            "import kernel_cases as kc; from antmmf.hip import ops, _lib; lib = _lib.load(); lib.antmmf_debug_gemm_cell_launches.restype = ctypes.c_long;"
            "g = torch.Generator().manual_seed(11);"
            "cases = [(2560, 1024, 256, True, True), (2560, 1024, 768, False, False)] + ([(4608, 1024, 768, True, False)] if os.environ.get('ANTMMF_SLOW_TESTS') else []);"
            "\nfor I, J, R, wb, wr in cases:\n"
            "    X = torch.randn(I, R, generator=g).bfloat16(); W = (torch.randn(J, R, generator=g) * 0.06).bfloat16()\n"
            "    b = torch.randn(J, generator=g) if wb else None; r = (torch.randn(I, J, generator=g) * 3).bfloat16() if wr else None\n"
            "    ref = X.float() @ W.float().t() + (b if wb else 0) + (r.float() if wr else 0)\n"
            "    lib.antmmf_debug_set_gemm_variant(4 | (1 << 26)); n0 = lib.antmmf_debug_gemm_cell_launches()\n"
            "    y1 = ops.gemm(X, W, bias=b, residual=r); kc.check('k64r.cells', y1, ref, 2e-2, 1e-2)\n"
            "    assert lib.antmmf_debug_gemm_cell_launches() == n0 + 1, 'the cell tail did not run'\n"
            "    lib.antmmf_debug_set_gemm_variant(4 | (1 << 25)); y0 = ops.gemm(X, W, bias=b, residual=r)\n"
            "    assert lib.antmmf_debug_gemm_cell_launches() == n0 + 1\n"
            "    assert torch.equal(y0, y1), (R, int((y0 != y1).sum()))\n"
            "print('okcells')"
            % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=2400)
    assert "okcells" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gemm_wgrad_ring(ops):
    kc.case_gemm_wgrad_ring(ops, DEV, quick=not os.environ.get("ANTMMF_SLOW_TESTS"))


def test_split_hi_lo(ops):
    kc.case_split_hi_lo(ops, DEV)


def test_gemm_wgrad_segments(ops):
    """the q / k / v wgrad as ONE launch with a segmented destination (round 6)"""
    kc.case_gemm_wgrad_seg(ops, DEV)


def test_attention_self(ops):
    kc.case_attention(ops, DEV, B=2, heads=2, Nq=17, Nk=17, bias_kind="none")


def test_attention_masks(ops):
    kc.case_attention(ops, DEV, B=2, heads=1, Nq=12, Nk=12, bias_kind="bert")
    kc.case_attention(ops, DEV, B=2, heads=1, Nq=12, Nk=12, bias_kind="inf")


@pytest.mark.parametrize("variant", [4, 6])
def test_attention_fwd32_opt_in(variant):
    """attn_fwd32_kernel (32 x 32 x 16 MFMA tiles, online softmax over 32- / 64-key blocks; opt-in, ANTMMF_ATTN_VARIANT bits 2 / 1 -- measured slower than the
    whole-row kernel): forward against the oracle on the key counts that select it (193 - 224 and 257 - 288 keys: 7 or 9 tiles of 32), with and without a key bias, ragged query
    count (a last 32-query tile with ONE valid row, as the 257-token tower has); the backward of the same case consumes its output and lse."""
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_ATTN_VARIANT'] = '%d';"
            "import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cpu');"
            "kc.case_attention(ops, dev, B=1, heads=1, Nq=257, Nk=257, bias_kind='none');"
            "kc.case_attention(ops, dev, B=2, heads=1, Nq=200, Nk=200, bias_kind='inf');"
            "kc.case_attention(ops, dev, B=1, heads=2, Nq=33, Nk=270, bias_kind='bert', packed=False);"
            "print('okfwd32')" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB, variant))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
    assert "okfwd32" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_attention_backward_one_kernel(ops):
    """attn_bwd_fused64_kernel (head size 64, 32 < keys <= 272, no dropout): the 257-token towers (a 17th key tile with ONE valid key, a last chunk with one valid query),
    even / odd numbers of resident key tiles (a missing half slice), masked keys incl. -inf, cross lengths either way, the largest size; the shorter cases above keep
    exercising the two-kernel backward."""
    import ctypes
    from antmmf.hip import _lib
    lib = _lib.load()
    lib.antmmf_debug_attn_fused_launches.restype = ctypes.c_long
    n0 = lib.antmmf_debug_attn_fused_launches()
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=257, Nk=257, bias_kind="none")
    kc.case_attention(ops, DEV, B=2, heads=1, Nq=200, Nk=200, bias_kind="inf")
    kc.case_attention(ops, DEV, B=1, heads=2, Nq=33, Nk=270, bias_kind="bert", packed=False)
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=150, Nk=130, bias_kind="bert", packed=False)
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=20, Nk=256, bias_kind="none", packed=False)
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=288, Nk=272, bias_kind="inf", packed=False)
    assert lib.antmmf_debug_attn_fused_launches() == n0 + 18, "the one-kernel backward did not run"   # (each case three times: without the token sums, with them, with them but dV's)
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=40, Nk=32, bias_kind="none", packed=False)   # (one or two key tiles: two kernels)
    assert lib.antmmf_debug_attn_fused_launches() == n0 + 18
    kc.case_attention(ops, DEV, B=2, heads=1, Nq=230, Nk=230, bias_kind="bert")               # (fifteen key tiles)
    assert lib.antmmf_debug_attn_fused_launches() == n0 + 21
    assert not ops.attention_bwd_sums_ok(64, 257, 257, 0.1) and not ops.attention_bwd_sums_ok(128, 257, 257) and not ops.attention_bwd_sums_ok(64, 40, 32) and ops.attention_bwd_sums_ok(64, 77, 77) and ops.attention_bwd_sums_ok(64, 257, 257)


def test_attention_backward_one_kernel_persistent_walk():
    """The persistent form of attn_bwd_fused64_kernel (all sixteen resident key tiles present) on a grid of TWO workgroups (lab knob), so each walks several (b, h) items:
    the next item's Q / dO pieces land in the rows the current item has finished with, its row statistics go to the second buffer, its K tile is requested after the last
    dQ contraction -- five items of the 257-token shape with a key mask (3 + 2), cross lengths with a 17th key tile (not persistent: run-time chunk count), exactly sixteen key tiles, nine
    and thirteen key tiles (a wave without a second tile; a missing half slice), and the short towers: five, three and eight key tiles (waves without any tile, which only
    contract their dQ tile)."""
    import sys

    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_HIP_LIB'] = %r; os.environ['ANTMMF_ATTN_PERSIST_WGS'] = '2';"
            "import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cpu');"
            "kc.case_attention(ops, dev, B=5, heads=1, Nq=257, Nk=257, bias_kind='bert');"
            "kc.case_attention(ops, dev, B=2, heads=2, Nq=33, Nk=270, bias_kind='bert', packed=False);"
            "kc.case_attention(ops, dev, B=3, heads=1, Nq=20, Nk=256, bias_kind='none', packed=False);"
            "kc.case_attention(ops, dev, B=3, heads=1, Nq=150, Nk=130, bias_kind='bert', packed=False);"
            "kc.case_attention(ops, dev, B=3, heads=1, Nq=197, Nk=197, bias_kind='inf');"
            "kc.case_attention(ops, dev, B=5, heads=1, Nq=77, Nk=77, bias_kind='bert');"
            "kc.case_attention(ops, dev, B=3, heads=1, Nq=288, Nk=33, bias_kind='inf', packed=False);"
            "kc.case_attention(ops, dev, B=3, heads=1, Nq=40, Nk=128, bias_kind='none', packed=False);"
            "print('okwalk')" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "ant-multi-modal-framework_amd"), ROOT, EMU_LIB))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=2400)
    assert "okwalk" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_attention_backward_rejects_misaligned_operands(ops):
    """16-byte operand vectors / LDS-DMA pieces: a q view that starts 8 bytes into an allocation is ANTMMF_EINVAL, not a fault."""
    B, N, h = 1, 130, 1
    buf = torch.zeros(B * N * 64 + 8, dtype=torch.bfloat16)
    q_bad = buf[4:4 + B * N * 64].view(B, N, 64)
    k = torch.zeros(B, N, 64, dtype=torch.bfloat16); v = torch.zeros_like(k); o = torch.zeros_like(k); d_o = torch.zeros_like(k)
    lse = torch.zeros(B, h, N)
    with pytest.raises(RuntimeError, match="code -22"):
        ops.attention_bwd(q_bad, k, v, o, lse, d_o, h, 0.125)


def test_attention_cross_multichunk(ops):
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=21, Nk=77, bias_kind="bert", packed=False)


def test_attention_head_size_128(ops):
    """ViLBERT's co-attention head size (bi_hidden_size 1024 / 8 heads): 256-B token rows, its own LDS swizzle, 4 MFMAs per score tile."""
    kc.case_attention(ops, DEV, B=1, heads=2, Nq=19, Nk=19, bias_kind="none", head_dim=128)
    kc.case_attention(ops, DEV, B=1, heads=1, Nq=13, Nk=37, bias_kind="bert", packed=False, head_dim=128)


def test_moco_and_ema(ops):
    kc.case_moco(ops, DEV)
    kc.case_moco(ops, DEV, R=5, Np=1, K=64)


def test_dmae_losses(ops):
    from antmmf.hip import contrastive

    kc.case_dmae_losses(contrastive, DEV)


def test_retrieval_metrics(ops, golden):
    kc.case_retrieval_metrics(DEV, golden)


def test_m2_eval_retrieval(ops, golden, tmp_path):
    kc.case_m2_eval_retrieval(DEV, golden, str(tmp_path))


def test_dropout_masks(ops):
    kc.case_dropout(ops, DEV)


def test_milnce(ops):
    kc.case_milnce(ops, DEV, Bg=6, n=2, world=2)
    kc.case_milnce(ops, DEV, Bg=4, n=1, world=1)
    kc.case_milnce(ops, DEV, Bg=3, n=3, world=1)


def test_softmax_ce(ops):
    kc.case_softmax_ce(ops, DEV)


def test_tpmcl_shape_guards(ops):
    """Beyond the row budget of the fused TPM-CL kernels (token sets > 128, selector rows > 64: e.g. frames x patches video tokens -- configurations the shipped ymls do not use)
    the wrappers take the same arithmetic as device tensor ops instead of failing: values and gradients against plain torch."""
    from antmmf.hip import tpmcl

    g = torch.Generator().manual_seed(21)
    feat = torch.randn(3, 150, 24, generator=g, requires_grad=True)
    w = torch.randn(1, 24, generator=g, requires_grad=True)
    b = torch.randn(1, generator=g, requires_grad=True)
    mask = (torch.rand(3, 150, generator=g) > 0.2).float(); mask[:, 0] = 1
    p = tpmcl.token_weights(feat, w, b, mask)
    ref = torch.softmax((feat @ w.t()).squeeze(-1).add(b).masked_fill(mask < 0.5, float("-inf")), -1)
    torch.testing.assert_close(p, ref, rtol=1e-5, atol=1e-6)
    p.square().sum().backward()
    assert feat.grad is not None and w.grad is not None and b.grad is not None
    x = torch.randn(4, 16, generator=g, requires_grad=True)
    y = torch.randn(4, 140, 16, generator=g, requires_grad=True)
    ww = torch.randn(4, 140, generator=g)
    torch.testing.assert_close(tpmcl.pair_dots(x, y), torch.einsum("cd,cvd->cv", x, y), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(tpmcl.pair_wsum(ww, y), torch.einsum("cv,cvd->cd", ww, y), rtol=1e-5, atol=1e-5)
    tpmcl.pair_dots(x, y).sum().backward()
    assert x.grad is not None and y.grad is not None
    wt = torch.softmax(torch.randn(5, 90, generator=g), -1)
    keep = tpmcl.tis_keep(wt, 0.3)
    order = torch.argsort(wt, dim=-1, descending=True)
    drop = torch.zeros_like(wt).scatter_(-1, order, (torch.cumsum(torch.gather(wt, -1, order), -1) < 0.3).float())
    assert torch.equal(keep, 1.0 - drop) and 0 < float(drop.sum()) < drop.numel()


def test_tpmcl_ops_and_linear_f32(ops):
    """csrc/tpmcl.hip + antmmf.hip.tpmcl on the lane emulator, the split GEMM of linear_f32 on the emulated MFMA GEMM."""
    kc.case_tpmcl_ops(DEV, C=9, V=5, T=7, D=40)
