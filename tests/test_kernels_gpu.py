"""Hardware parity: every C-ABI kernel family on a real MI355X against the CPU oracle (-m gpu)."""
import os

import pytest
import torch

import kernel_cases as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from antmmf.hip import _lib

    os.environ.pop("ANTMMF_HIP_LIB", None)
    _lib.reset_for_tests()
    from antmmf.hip import ops as hops

    assert _lib.backend() == 1, "the GPU tests must run against the gfx950 library, not the emulator"
    assert torch.cuda.is_available()
    return hops


DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def lab(ops):
    """The LAB library (libantmmf_hip_lab.so: the same kernels + the A/B switches, forced tile sizes, experiment kernels and the sub-LN fold that the product library
    does not have) for the duration of one test; the product library is back afterwards."""
    from antmmf.hip import _lib

    if not os.path.isfile(_lib.LAB_LIB):
        pytest.skip("lab library not built (make -C ant-multi-modal-framework_amd/csrc lab)")
    os.environ["ANTMMF_HIP_LIB"] = _lib.LAB_LIB
    _lib.reset_for_tests()
    lib = _lib.load()
    assert _lib.is_lab() and _lib.backend() == 1
    try:
        yield lib
    finally:
        os.environ.pop("ANTMMF_HIP_LIB", None)
        _lib.reset_for_tests()
        assert not _lib.is_lab()


def lab_env():
    from antmmf.hip import _lib

    return dict(os.environ, ANTMMF_HIP_LIB=_lib.LAB_LIB)


def test_product_library_has_no_switches(ops):
    """libantmmf_hip.so: no A/B setter, no fold entry points, no ANTMMF_* string (it reads no environment variable)."""
    import ctypes
    from antmmf.hip import _lib

    assert not _lib.is_lab()
    lib = ctypes.CDLL(_lib.DEFAULT_LIB)
    for name in ("antmmf_debug_set_gemm_variant", "antmmf_ffn_fc1_fwd"):
        assert not hasattr(lib, name), name
    assert b"ANTMMF_" not in open(_lib.DEFAULT_LIB, "rb").read()


def test_layernorm(ops):
    for dtype in (torch.float32, torch.bfloat16):
        kc.case_layernorm(ops, DEV, dtype)
        kc.case_layernorm(ops, DEV, dtype, rows=1031, cols=768, eps=1e-12)
        kc.case_layernorm(ops, DEV, dtype, rows=517, cols=1024)
        kc.case_layernorm(ops, DEV, dtype, rows=9001, cols=1024)  # partials path of the wave-per-row backward
        kc.case_layernorm(ops, DEV, dtype, rows=130, cols=4096)
        kc.case_layernorm(ops, DEV, dtype, rows=77, cols=3072)


def test_act_layernorm(ops):
    for dtype in (torch.float32, torch.bfloat16):
        kc.case_act_layernorm(ops, DEV, dtype, rows=300, cols=4096, act="gelu")
        kc.case_act_layernorm(ops, DEV, dtype, rows=1500, cols=3072, act="gelu")
        kc.case_act_layernorm(ops, DEV, dtype, rows=77, cols=512, act="quick_gelu")


def test_ffn_fold(ops, lab):
    """(lab library) Sub-LN fold: element-wise epilogues + row / column passes (unaligned shapes), then ViT-L/14 widths where all three GEMMs run on the persistent
    kernel (>= 512 tiles each: fc1 / dgrad 128 x 16, fc2 128 x 4) with their per-tile partial sums."""
    kc.case_ffn_fold(ops, DEV, tokens=40, d=64, ff=192)
    kc.case_ffn_fold(ops, DEV, tokens=515, d=768, ff=3072, res_scale=4.0, seed=340)
    kc.case_ffn_fold(ops, DEV, tokens=77, d=128, ff=512, act="quick_gelu", seed=380)
    import ctypes
    from antmmf.hip import _lib

    lib = lab
    lib.antmmf_debug_gemm_k64_launches.restype = ctypes.c_long
    before = lib.antmmf_debug_gemm_k64_launches()
    kc.case_ffn_fold(ops, DEV, tokens=32768, d=1024, ff=4096, seed=500)
    assert lib.antmmf_debug_gemm_k64_launches() - before >= 4   # fc1, fc2, dgrad (+ the wgrad)


def test_ffn_fold_forced_persistent():
    """The same operator with every GEMM forced onto the persistent kernel at a size where workgroups walk several tiles of a short grid."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_GEMM_FORCE_TILE'] = 'k'; os.environ['ANTMMF_GEMM_PERSIST_WGS'] = '24';"
            "import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cuda:0');"
            "kc.case_ffn_fold(ops, dev, tokens=2048, d=1024, ff=4096, seed=520); kc.case_ffn_fold(ops, dev, tokens=768, d=768, ff=3072, seed=540); print('okffn')"
            % (os.path.join(root, "tests"), os.path.join(root, "ant-multi-modal-framework_amd"), root))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=lab_env())
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
    kc.case_gemm(ops, DEV, I=513, J=264, R=328)
    kc.case_gemm_multitile(ops, DEV)


def test_gemm_persistent_ring(ops):
    """530 tiles of 256 x 256 on 256 persistent workgroups: every workgroup walks 2-3 tiles (ragged edges, all epilogues)."""
    kc.case_gemm_persistent(ops, DEV, I=13412, J=2560, R=256)


def test_gemm_k64_persistent(ops):
    """The BK = 64 quarter-phase persistent kernel (256-aligned shapes, >= 512 tiles): 520 tiles on 256 workgroups (every workgroup walks
    2-3 tiles, ring continuous across them), nk = 4 and nk = 2; all epilogues against the fp32 oracle GEMM."""
    from antmmf.hip import _lib

    lib = _lib.load()
    lib.antmmf_debug_gemm_k64_launches.restype = __import__("ctypes").c_long
    before = lib.antmmf_debug_gemm_k64_launches()
    kc.case_gemm_k64(ops, DEV, I=13312, J=2560, R=256)
    kc.case_gemm_k64(ops, DEV, I=33024, J=1024, R=128, quick=True)
    # 8 of the first case's 9 calls run on the BK = 64 kernels (plain / bias / residual / both / two-output activation / gated dgrad with the stored or the recomputed
    # derivative / strided output); the generic run-time epilogue (activation + residual + aux + alpha) and R = 128 are served by the BK = 32 ring kernels since round 6
    assert lib.antmmf_debug_gemm_k64_launches() - before >= 8, "the dispatcher did not pick the k64 kernel"


@pytest.mark.parametrize("which", ["product", "lab"])
def test_gemm_k64_full_size_vs_fp32(ops, which, request):
    """BASELINE sizes (ViT-L/14, 256 pairs): fc2-shaped GEMM with bias + residual, fc1-shaped with bias, a plain dgrad shape -- on the rolling-epilogue kernel
    (the default for these shapes) against an fp32 matmul of the same bf16 operands on sampled rows; every run of it bit-identical to the first (a DMA piece
    consumed before it landed would show up as a sporadically different tile).  [product]: the product library, which has exactly this path.  [lab]: the lab library
    -- its default equals the product's output bit for bit (same kernels), the burst-epilogue kernel (variant bit 14) is bit-exact against the BK = 32 ring kernel
    (same K order), and the rolling kernel is bit-exact for the plain epilogue and within one bf16 ulp on a small fraction of the elements where bias / residual
    enter the fp32 sum first instead of last."""
    from antmmf.hip import _lib

    prod = []
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = ((1024, 4096, True, True), (4096, 1024, True, False), (1024, 3072, False, False))
    I = 257 * 256
    if which == "lab":   # the product library's results first, then the same inputs on the lab library
        gp = torch.Generator(device="cuda").manual_seed(5)
        for (J, R, with_bias, with_res) in shapes:
            X = torch.randn(I, R, generator=gp, device=DEV).bfloat16()
            W = (torch.randn(J, R, generator=gp, device=DEV) * R ** -0.5).bfloat16()
            bias = torch.randn(J, generator=gp, device=DEV) if with_bias else None
            res = torch.randn(I, J, generator=gp, device=DEV).bfloat16() if with_res else None
            prod.append(ops.gemm(X, W, bias=bias, residual=res))
        lib = request.getfixturevalue("lab")
    else:
        lib = None
    for si, (J, R, with_bias, with_res) in enumerate(shapes):
        X = torch.randn(I, R, generator=g, device=DEV).bfloat16()
        W = (torch.randn(J, R, generator=g, device=DEV) * R ** -0.5).bfloat16()
        bias = torch.randn(J, generator=g, device=DEV) if with_bias else None
        res = torch.randn(I, J, generator=g, device=DEV).bfloat16() if with_res else None
        y = ops.gemm(X, W, bias=bias, residual=res)
        rows = torch.randint(0, I, (96,), device=DEV)
        ref = X[rows].float() @ W.float().t() + (bias if with_bias else 0) + (res[rows].float() if with_res else 0)
        torch.testing.assert_close(y[rows].float(), ref, rtol=2e-2, atol=2e-2)
        for _ in range(3):
            assert torch.equal(ops.gemm(X, W, bias=bias, residual=res), y)
        if lib is None:
            continue
        assert torch.equal(y, prod[si]), "lab default != product library"
        try:
            lib.antmmf_debug_set_gemm_variant(0)
            y0 = ops.gemm(X, W, bias=bias, residual=res)
            lib.antmmf_debug_set_gemm_variant(4 | 16384)
            yb = ops.gemm(X, W, bias=bias, residual=res)
        finally:
            lib.antmmf_debug_set_gemm_variant(4)
        assert torch.equal(yb, y0)
        if not (with_bias or with_res):
            assert torch.equal(y, y0)
        else:
            diff = (y.float() - y0.float()).abs()
            big = torch.maximum(y.float().abs(), y0.float().abs()).clamp_min(2.0 ** -100)
            ulp = big.log2().floor().exp2() * 2.0 ** -7      # one bf16 ulp at the larger magnitude; results near zero may differ by the fp32 reordering error itself
            assert int((diff > 0).sum()) <= y.numel() // 200 and bool((diff <= ulp * 1.001 + 2e-5).all()), (int((diff > 0).sum()), float((diff - ulp).max()))


def test_gemm_wgrad_ring(ops):
    kc.case_gemm_wgrad_ring(ops, DEV)
    kc.case_gemm_wgrad_ring(ops, DEV, tokens=257 * 64, n_out=1024, k_in=512)


def test_split_hi_lo(ops):
    kc.case_split_hi_lo(ops, DEV)


def test_gemm_wgrad_segments(ops):
    """the q / k / v wgrad as ONE launch with a segmented destination (round 6): toy size and the flagship's (3 x 1024 x 1024 over 65792 tokens)"""
    kc.case_gemm_wgrad_seg(ops, DEV)
    kc.case_gemm_wgrad_seg(ops, DEV, tokens=257 * 256, rows=1024, k_in=1024)


def test_gemm_large_linearity(ops):
    """Full-size property check (BASELINE sizes; the oracle would take minutes): the GEMM is linear in P,
    and agrees with an fp32 matmul of the same bf16 operands on a random sample of rows."""
    I, J, R = 257 * 64, 4096, 1024
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn(I, R, generator=g, device=DEV).bfloat16()
    W = (torch.randn(J, R, generator=g, device=DEV) * R ** -0.5).bfloat16()
    y = ops.gemm(X, W, out_dtype=torch.float32)
    y2 = ops.gemm((X.float() * 2).bfloat16(), W, out_dtype=torch.float32)
    torch.testing.assert_close(y2, 2 * y, rtol=1e-5, atol=1e-5)
    rows = torch.randint(0, I, (64,), device=DEV)
    ref = X[rows].float() @ W.float().t()
    torch.testing.assert_close(y[rows], ref, rtol=2e-3, atol=2e-3)


def test_gemm_persistent_uneven_rounds(ops):
    """bf16 persistent path (532 tiles on 256 workgroups: 2 rounds + a nearly empty third, ragged last row tile), every compile-time
    epilogue; rows sampled across the whole range, around row 128 * 256 and at the ragged end against an fp32 matmul of the same bf16
    operands.  (Peeling the last row tiles off to a 128 x 128-tile launch was measured at the bench sizes: +0.1 % on the step, not kept.)"""
    I, J, R = 132 * 256 + 100, 1024, 512
    g = torch.Generator(device="cuda").manual_seed(11)
    X = torch.randn(I, R, generator=g, device=DEV).bfloat16()
    W = (torch.randn(J, R, generator=g, device=DEV) * R ** -0.5).bfloat16()
    bias = torch.randn(J, generator=g, device=DEV)
    res = torch.randn(I, J, generator=g, device=DEV).bfloat16()
    rows = torch.cat([torch.randint(0, 128 * 256, (48,), device=DEV), torch.arange(128 * 256 - 3, 128 * 256 + 3, device=DEV),
                      torch.randint(128 * 256, I, (48,), device=DEV), torch.tensor([I - 1], device=DEV)])
    ref = X[rows].float() @ W.float().t()
    for kw, extra in ((dict(), 0), (dict(bias=bias), bias), (dict(residual=res), res[rows].float()), (dict(bias=bias, residual=res), bias + res[rows].float())):
        y = ops.gemm(X, W, **kw)
        assert y.dtype == torch.bfloat16 and torch.isfinite(y.float()).all()
        torch.testing.assert_close(y[rows].float(), ref + extra, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("cfg", [
    dict(B=2, heads=2, Nq=17, Nk=17, bias_kind="none"),
    dict(B=3, heads=2, Nq=12, Nk=12, bias_kind="bert"),
    dict(B=2, heads=2, Nq=12, Nk=12, bias_kind="inf"),
    dict(B=2, heads=12, Nq=77, Nk=77, bias_kind="bert"),
    dict(B=2, heads=12, Nq=197, Nk=197, bias_kind="none"),
    dict(B=2, heads=16, Nq=257, Nk=257, bias_kind="none"),
    dict(B=2, heads=2, Nq=21, Nk=77, bias_kind="bert", packed=False),
    dict(B=2, heads=2, Nq=288, Nk=33, bias_kind="inf", packed=False),
    dict(B=2, heads=8, Nq=100, Nk=100, bias_kind="none", head_dim=128),               # ViLBERT: 8 heads x 128, 100 regions
    dict(B=3, heads=8, Nq=37, Nk=100, bias_kind="bert", packed=False, head_dim=128),  # text queries over image regions, additive mask
    dict(B=2, heads=2, Nq=288, Nk=257, bias_kind="inf", packed=False, head_dim=128),  # the largest tiles (147 KB of LDS)
    # the one-kernel backward (head size 64, 32 < keys <= 272; the 77-, 197- and 257-token cases above run on it too): masks on the 17-tile case, a missing half slice,
    # cross lengths either way, the tile boundary, the largest size
    dict(B=2, heads=16, Nq=257, Nk=257, bias_kind="bert"),
    dict(B=3, heads=2, Nq=200, Nk=200, bias_kind="inf"),
    dict(B=2, heads=2, Nq=33, Nk=270, bias_kind="bert", packed=False),
    dict(B=2, heads=3, Nq=150, Nk=130, bias_kind="bert", packed=False),
    dict(B=2, heads=2, Nq=20, Nk=256, bias_kind="none", packed=False),
    dict(B=2, heads=2, Nq=288, Nk=272, bias_kind="inf", packed=False),
    dict(B=2, heads=2, Nq=64, Nk=129, bias_kind="none", packed=False),
])
def test_attention(ops, cfg):
    kc.case_attention(ops, DEV, **cfg)


def test_attention_backward_one_kernel_runs_and_repeats():
    """attn_bwd_fused64_kernel at the ViT-L/14 tower's shape (16 heads x 257 tokens, packed qkv rows; 640 (b, h) items on the 256 persistent workgroups: two or three items
    each, the next one's operands prefetched into the rows the current one has finished with): the lab library's counter says that kernel served the call, two calls
    give BIT-IDENTICAL gradients (dQ is contracted over all keys by one wave in a fixed order: no atomics), and they agree with the two-kernel backward (lab variant bit 3,
    separate process) to bf16 rounding of the outputs."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, ctypes, torch; sys.path[:0] = [%r, %r, %r];"
            "from antmmf.hip import ops, _lib; lib = _lib.load(); lib.antmmf_debug_attn_fused_launches.restype = ctypes.c_long; dev = torch.device('cuda:0');"
            "g = torch.Generator(device='cuda').manual_seed(3); B, N, h = 40, 257, 16;"
            "qkv = torch.randn(B, N, 3 * h * 64, generator=g, device=dev).bfloat16(); q, k, v = qkv[..., :h * 64], qkv[..., h * 64:2 * h * 64], qkv[..., 2 * h * 64:];"
            "o, lse = ops.attention_fwd(q, k, v, h, 0.125); do = torch.randn(B, N, h * 64, generator=g, device=dev).bfloat16();"
            "n0 = lib.antmmf_debug_attn_fused_launches(); r1 = ops.attention_bwd(q, k, v, o, lse, do, h, 0.125); r2 = ops.attention_bwd(q, k, v, o, lse, do, h, 0.125);"
            "n1 = lib.antmmf_debug_attn_fused_launches(); torch.cuda.synchronize();"
            "assert all(torch.equal(a, b) for a, b in zip(r1, r2));"
            "torch.save([t.cpu() for t in r1], sys.argv[1]); print('fused_launches', n1 - n0)"
            % (os.path.join(root, "tests"), os.path.join(root, "ant-multi-modal-framework_amd"), root))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        outs = {}
        for variant, want in (("0", 2), ("8", 0)):
            f = os.path.join(td, "g%s.pt" % variant)
            out = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=900, env=dict(lab_env(), ANTMMF_ATTN_VARIANT=variant))
            assert "fused_launches %d" % want in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
            outs[variant] = torch.load(f)
        for a, b, name in zip(outs["0"], outs["8"], ("dq", "dk", "dv")):
            a, b = a.float(), b.float()
            assert torch.isfinite(a).all()
            err = (a - b).abs().max().item()
            assert err <= 2.0 ** -6 * b.abs().max().item(), (name, err, b.abs().max().item())   # both round fp32 sums of the same products to bf16 (different summation orders)


@pytest.mark.parametrize("B,h,N,masked", [(48, 16, 257, False), (70, 16, 77, True), (40, 12, 197, False), (300, 12, 86, True), (70, 16, 129, True)])
def test_attention_backward_token_sums_at_tower_shapes(ops, B, h, N, masked):
    """antmmf_attention_bwd_sums at the towers' shapes, more (b, h) items than the 256 persistent workgroups (every workgroup walks several items: the per-wave partials in
    LDS are rewritten per item): gradients bit-identical to the plain backward, the [B, 3 D] token sums equal to the sums of the gradients it stored (to their bf16
    rounding), every element written; the q / k / v bias gradient = their column sums against the column sums of dQ | dK | dV."""
    g = torch.Generator(device="cuda").manual_seed(11)
    D = h * 64
    qkv = torch.randn(B, N, 3 * D, generator=g, device=DEV).bfloat16()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    kb = None
    if masked:
        lengths = torch.randint(3, N + 1, (B,), generator=g, device=DEV)
        kb = torch.zeros(B, N, device=DEV).masked_fill(torch.arange(N, device=DEV)[None, :] >= lengths[:, None], float("-inf"))
    o, lse = ops.attention_fwd(q, k, v, h, 0.125, kb)
    do = torch.randn(B, N, D, generator=g, device=DEV).bfloat16()
    assert ops.attention_bwd_sums_ok(64, N, N)
    r1 = ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, kb)
    sums = torch.full((B, 3 * D), float("nan"), device=DEV)
    r2 = ops.attention_bwd(q, k, v, o, lse, do, h, 0.125, kb, sums=sums)
    assert all(torch.equal(a, b) for a, b in zip(r1, r2))
    assert torch.isfinite(sums).all()
    own = torch.cat([t.float().sum(1) for t in r1], dim=1)
    scale = own.abs().max().item()
    err = (sums - own).abs().max().item()
    assert err <= 2e-2 * scale, (err, scale)   # N bf16 roundings of 2^-9 relative each, random signs
    bias = torch.zeros(3 * D, device=DEV)
    ops.colsum_(bias, sums)
    ref = torch.zeros(3 * D, device=DEV)
    ops.colsum_(ref, torch.cat(r1, dim=2).view(B * N, 3 * D))
    assert (bias - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("variant", [4, 6])
def test_attention_fwd32_opt_in(variant):
    """The opt-in forward on 32 x 32 x 16 MFMA tiles (ANTMMF_ATTN_VARIANT bit 2; bit 1: 64-key softmax blocks) at the towers' sizes: 257 tokens x 16 heads (ViT-L/14),
    197 x 12 (ViT-B/16), a masked and a cross case -- forward vs the oracle, its output + lse through the (unchanged) backward kernels."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, torch; sys.path[:0] = [%r, %r, %r]; os.environ['ANTMMF_ATTN_VARIANT'] = '%d';"
            "import kernel_cases as kc; from antmmf.hip import ops; dev = torch.device('cuda:0');"
            "kc.case_attention(ops, dev, B=2, heads=16, Nq=257, Nk=257, bias_kind='none');"
            "kc.case_attention(ops, dev, B=2, heads=12, Nq=197, Nk=197, bias_kind='none');"
            "kc.case_attention(ops, dev, B=3, heads=2, Nq=200, Nk=200, bias_kind='inf');"
            "kc.case_attention(ops, dev, B=2, heads=2, Nq=33, Nk=270, bias_kind='bert', packed=False);"
            "print('okfwd32')" % (os.path.join(root, "tests"), os.path.join(root, "ant-multi-modal-framework_amd"), root, variant))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=lab_env())
    assert "okfwd32" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gemm_k64_tail_round_cells(ops, lab):
    """The tail round of the persistent NT kernel as cells inside the same launch, at the sizes where the product uses it (263168 x 1024: 4112 tiles on 256 workgroups = 16
    rounds + 2 leftover tiles per XCD chunk -> 32 cells, one per workgroup; K = 4096 and K = 1024, with and without bias + residual): the product library's output is
    BIT-IDENTICAL to the lab library's with the cells on (same code) and with the cells off (variant bit 25: leftover tiles through the walk) -- the cells start from
    bias + residual and add the K-tiles in the walk's order -- and rows of the leftover tiles agree with an fp32 product."""
    import ctypes
    from antmmf.hip import _lib

    lib = lab
    lib.antmmf_debug_gemm_cell_launches.restype = ctypes.c_long
    g = torch.Generator(device="cuda").manual_seed(9)
    I, J = 257 * 1024, 1024
    for R, with_br in ((4096, True), (1024, False), (1024, True)):
        X = torch.randn(I, R, generator=g, device=DEV).bfloat16()
        W = (torch.randn(J, R, generator=g, device=DEV) * R ** -0.5).bfloat16()
        b = torch.randn(J, generator=g, device=DEV) if with_br else None
        r = torch.randn(I, J, generator=g, device=DEV).bfloat16() if with_br else None
        try:
            lib.antmmf_debug_set_gemm_variant(4)
            n0 = lib.antmmf_debug_gemm_cell_launches()
            y1 = ops.gemm(X, W, bias=b, residual=r)
            assert lib.antmmf_debug_gemm_cell_launches() == n0 + 1, "the dispatcher did not pick the cell tail"
            for _ in range(3):
                assert torch.equal(ops.gemm(X, W, bias=b, residual=r), y1)    # (a DMA piece consumed before it landed would show up as a sporadically different cell)
            lib.antmmf_debug_set_gemm_variant(4 | (1 << 25))
            y0 = ops.gemm(X, W, bias=b, residual=r)
            assert lib.antmmf_debug_gemm_cell_launches() == n0 + 4
        finally:
            lib.antmmf_debug_set_gemm_variant(4)
        assert torch.equal(y0, y1), (R, with_br, int((y0 != y1).sum()))
        # the product library on the same inputs
        os.environ.pop("ANTMMF_HIP_LIB", None)
        _lib.reset_for_tests()
        try:
            assert not _lib.is_lab()
            yp = ops.gemm(X, W, bias=b, residual=r)
        finally:
            os.environ["ANTMMF_HIP_LIB"] = _lib.LAB_LIB
            _lib.reset_for_tests()
            _lib.load()
        assert torch.equal(yp, y1)
        # leftover tiles of XCD chunk x: local ids 512, 513 of its 514 -> global tile ids 514 x + 512 + {0, 1}; rows of those tiles (4 x 8 patches: band = id // 16)
        rows = []
        for x in range(8):
            for u in (512, 513):
                wg = 514 * x + u
                band, inb = wg // 16, wg % 16
                rows_here = min(4, 1028 - 4 * band)
                rows += [(band * 4 + inb % rows_here) * 256 + 7, (band * 4 + inb % rows_here) * 256 + 250]
        rows = torch.unique(torch.tensor(rows + [0, 70000, I - 1], device=DEV))
        ref = X[rows].float() @ W.float().t() + (b if with_br else 0) + (r[rows].float() if with_br else 0)
        err = (y1[rows].float() - ref).abs()
        ulp = torch.maximum(ref.abs(), y1[rows].float().abs()) * 2.0 ** -8
        assert bool((err <= ulp * 1.001 + 2e-5).all()), float((err / (ulp + 1e-9)).max())


def test_moco_and_ema(ops):
    kc.case_moco(ops, DEV)
    kc.case_moco(ops, DEV, R=5, Np=1, K=64)


def test_dmae_losses(ops):
    from antmmf.hip import contrastive

    kc.case_dmae_losses(contrastive, DEV)


def test_tpmcl_ops(ops):
    kc.case_tpmcl_ops(DEV)
    kc.case_tpmcl_ops(DEV, C=2048, V=13, T=30, D=768)   # the dmae12 bench's pair count and widths


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
    kc.case_milnce(ops, DEV, Bg=512, n=1, world=4)
    kc.case_milnce(ops, DEV, Bg=96, n=4, world=2)


def test_softmax_ce(ops):
    kc.case_softmax_ce(ops, DEV)
    kc.case_softmax_ce(ops, DEV, Bg=640, world=4)


def test_resize_bicubic_vs_pillow_goldens(ops, golden):
    import resize_cases as rc

    print(rc.case_goldens(DEV, golden))


def test_resize_bicubic_full_size_ragged_batch_vs_oracle(ops):
    """Photo-sized, ragged batch to the 224 x 224 tower input: every byte equal to the oracle (= Pillow); includes extents that skip a
    pass, an upscale, a 1-pixel-high strip and a row longer than 64 KB."""
    import resize_cases as rc

    sizes = [(1080, 1920), (1920, 1080), (480, 640), (224, 224), (224, 600), (700, 224), (100, 150), (1, 300), (300, 1), (64, 23000), (2160, 3840)]
    print(rc.case_vs_oracle(DEV, sizes, 224, seed=5))
    print(rc.case_vs_oracle(DEV, [(300, 400), (224, 500)], 224, seed=6, channels=4))
    print(rc.case_vs_oracle(DEV, [(300, 400), (90, 64)], 224, seed=7, channels=1))
