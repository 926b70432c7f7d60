"""BASELINE config 2's loss at its TRUE size: global batch 8192 = 8 ranks x 1024 pairs, D = 1024, through the PRODUCT functions
(packed gather -> hi/lo split MFMA similarity slabs -> row kernels -> four GEMMs back -> packed reduce-scatter) with the 8 ranks simulated one after the
other on the one GPU (tests/sim_ranks.py), against the oracle's full [8192 x 8192] loss on the CPU: loss, every embedding gradient, the logit-scale
gradients.  Reference: logits prj/M2_Encoder/m2_encoder.py:92-95 (the symmetric cross-entropy is this build's, oracle/losses.py::clip_itc);
MIL-NCE prj/base_vtp/roi_univl/univl/model/univl_video_ret.py:146-197."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def contrastive():
    from antmmf.hip import _lib

    os.environ.pop("ANTMMF_HIP_LIB", None)
    _lib.reset_for_tests()
    from antmmf.hip import contrastive as C

    assert _lib.backend() == 1 and torch.cuda.is_available()
    return C


def _unit(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)


@pytest.mark.parametrize("world,B,D", [(8, 1024, 1024), (4, 96, 256)])
def test_m2_itc_pair_at_global_batch(contrastive, world, B, D):
    """clip_itc_pair_sharded: rank r's B rows against all W B columns, both ITC terms with their own logit scale."""
    from oracle import losses
    from sim_ranks import SimulatedRanks

    Bg = world * B
    # correlated pairs (a trained tower's regime: the diagonal stands out) so that the softmax is neither uniform nor one-hot
    base = _unit(Bg, D, 1)
    sets = [base, torch.nn.functional.normalize(base + 0.8 * _unit(Bg, D, 2), dim=-1), _unit(Bg, D, 3),
            torch.nn.functional.normalize(_unit(Bg, D, 3) + 0.5 * _unit(Bg, D, 4), dim=-1)]
    ls_init = (math.log(1 / 0.07), math.log(10.0))
    ref_in = [s.clone().requires_grad_(True) for s in sets]
    ref_ls = [torch.tensor(v, requires_grad=True) for v in ls_init]
    r1, _ = losses.clip_itc(ref_in[0], ref_in[1], ref_ls[0])
    r2, _ = losses.clip_itc(ref_in[2], ref_in[3], ref_ls[1])
    (r1 + r2).backward()

    sim = SimulatedRanks(world)
    dev_sets = [s.to(DEV) for s in sets]
    ls = [torch.tensor(v, device=DEV, requires_grad=True) for v in ls_init]
    for r in range(world):
        loc = [s[r * B:(r + 1) * B].clone().requires_grad_(True) for s in dev_sets]
        with sim.as_rank(r, dev_sets):
            l1, l2 = contrastive.clip_itc_pair_sharded(loc[0], loc[1], ls[0], loc[2], loc[3], ls[1])
            (l1 + l2).backward()
    parts = torch.stack(sim.loss_parts).sum(0).cpu()     # what the all-reduce would have produced: [2]
    assert abs(float(parts[0]) - float(r1.detach())) <= 1e-4 * abs(float(r1.detach())), (float(parts[0]), float(r1.detach()))
    assert abs(float(parts[1]) - float(r2.detach())) <= 1e-4 * abs(float(r2.detach())), (float(parts[1]), float(r2.detach()))
    # the reduce-scatter sums carry the reference's x W; the data-parallel mean takes it out again
    for got, want in zip(sim.grad_sums, ref_in):
        g = (got / world).cpu()
        torch.testing.assert_close(g, want.grad, rtol=2e-3, atol=2e-3 * float(want.grad.abs().max()))
        cos = torch.nn.functional.cosine_similarity(g.flatten(), want.grad.flatten(), dim=0)
        assert float(cos) > 0.99999, float(cos)
    for got, want in zip(ls, ref_ls):
        assert abs(float(got.grad) / world - float(want.grad)) <= 2e-3 * max(abs(float(want.grad)), 1e-3), (float(got.grad) / world, float(want.grad))


@pytest.mark.parametrize("world,B,n,D", [(8, 1024, 1, 1024), (4, 64, 3, 256)])
def test_mil_nce_at_global_batch(contrastive, world, B, n, D):
    """mil_nce_sharded: text rows against all W B n clips (centre-clip columns against all texts)."""
    from oracle import losses
    from sim_ranks import SimulatedRanks

    Bg = world * B
    T = _unit(Bg, D, 11)
    V = torch.nn.functional.normalize(T.repeat_interleave(n, 0) + 0.9 * _unit(Bg * n, D, 12), dim=-1)
    Tr, Vr = T.clone().requires_grad_(True), V.clone().requires_grad_(True)
    simi = torch.matmul(Vr.view(Bg, n, D), Tr.t()).permute(2, 0, 1)
    mil = simi.unsqueeze(1).expand(Bg, n, Bg, n).reshape(Bg * n, Bg * n)
    ref = losses.mil_nce(mil, Bg, n)
    ref.backward()

    sim = SimulatedRanks(world)
    Td, Vd = T.to(DEV), V.to(DEV)
    for r in range(world):
        t = Td[r * B:(r + 1) * B].clone().requires_grad_(True)
        v = Vd[r * B * n:(r + 1) * B * n].clone().requires_grad_(True)
        with sim.as_rank(r, [Td, Vd.view(Bg, n * D)]):
            contrastive.mil_nce_sharded(t, v, n_clips=n).backward()
    total = float(torch.stack(sim.loss_parts).sum())
    assert abs(total - float(ref.detach())) <= 1e-4 * abs(float(ref.detach())), (total, float(ref.detach()))
    for got, want in ((sim.grad_sums[0], Tr.grad), (sim.grad_sums[1].view(Bg * n, D), Vr.grad)):
        g = (got / world).cpu()
        torch.testing.assert_close(g, want, rtol=2e-3, atol=2e-3 * float(want.abs().max()))
        assert float(torch.nn.functional.cosine_similarity(g.flatten(), want.flatten(), dim=0)) > 0.99999
