"""Launches the dominant PLAIN-bf16 kernels (BASELINE config C3) alone at C2 shapes (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
M, N, K = 8192, 4096, 25088
A = torch.relu(torch.randn(M, K, device='cuda', generator=g)); B = torch.randn(N, K, device='cuda', generator=g) * 0.01
bias = torch.zeros(N, device='cuda')
Xp, Wp = ops.split_bf16(A, False), ops.split_bf16(B, False)
del A, B
x = torch.relu(torch.randn(64, 56, 56, 256, device='cuda', generator=g)); w = torch.randn(256, 3, 3, 256, device='cuda', generator=g) * 0.02
xp, wp = ops.split_bf16(x, False), ops.split_bf16(w, False); cb = torch.zeros(256, device='cuda')
for _ in range(3):
    ops.gemm_nt_bf16(Xp, Wp, bias, act=1, want_f32=False, want_planes=True)
    ops.conv3x3_bf16(xp, wp, cb)
torch.cuda.synchronize(); print("done")
