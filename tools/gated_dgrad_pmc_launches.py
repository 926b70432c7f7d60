"""Three launches each of the gated dgrad (EPI 8) and of the gated dgrad with column sums (EPI 10) at the video workloads' largest shape: the process to put under
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (HBM-side bytes of the two kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
from antmmf.hip import ops
dev = torch.device("cuda:0")
T, J, R = 352256, 3072, 768
X = torch.randn(T, R, device=dev).bfloat16(); W = (torch.randn(J, R, device=dev) * R ** -0.5).bfloat16(); g = torch.rand(T, J, device=dev).bfloat16()
for _ in range(3):
    ops.gemm(X, W, gate=g, act="gelu", gate_is_grad=True)
    ops.gemm_gated_colsum(X, W, g)
torch.cuda.synchronize()
