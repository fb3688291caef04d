"""Launches the dominant kernels alone at BASELINE C2 shapes (for rocprofv3 --pmc passes, which must not be combined
with tracing domains other than --kernel-trace).  Usage: python3 scripts/kernels_only.py [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R, Q, D = 8192, 128, 512
g = torch.Generator(device='cuda').manual_seed(0)
A = torch.randn(R, 25088, device='cuda', generator=g)
B = torch.randn(4096, 25088, device='cuda', generator=g) * 0.01
bias = torch.zeros(4096, device='cuda')
x = torch.randn(64, 56, 56, 256, device='cuda', generator=g)
w = torch.randn(256, 3, 3, 256, device='cuda', generator=g) * 0.02
cb = torch.zeros(256, device='cuda')
V = torch.tanh(torch.randn(R, D, device='cuda', generator=g)); W = torch.tanh(torch.randn(Q, D, device='cuda', generator=g))
el = torch.tensor([2, 3, 1, 0, 4, 2, 5, 1], dtype=torch.int32, device='cuda')
for _ in range(it):
    ops.gemm_nt(A, B, bias, act=ops.ACT_RELU)
    ops.conv3x3_relu(x, w, cb)
    ops.sim_max_fwd(V, W, el, 8, 8, 128, 16)
torch.cuda.synchronize()
print("done")
