"""64->64 @224^2 fp32 conv alone (for PMC passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.relu(torch.randn(64, 224, 224, 64, device='cuda', generator=g)); w = torch.randn(64, 3, 3, 64, device='cuda', generator=g) * 0.02
cb = torch.zeros(64, device='cuda')
x2 = torch.relu(torch.randn(64, 112, 112, 128, device='cuda', generator=g)); w2 = torch.randn(128, 3, 3, 128, device='cuda', generator=g) * 0.02
cb2 = torch.zeros(128, device='cuda')
for _ in range(3):
    ops.conv3x3_relu(x, w, cb)
    ops.conv3x3_relu(x2, w2, cb2)
torch.cuda.synchronize(); print("done")
