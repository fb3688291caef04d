"""Launches the first VGG layer (conv1_1, 3 -> 64 channels) alone at C2's 64 frames, fp32 and plane outputs (rocprofv3 passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(64, 3, 224, 224, device='cuda', generator=g)
w = torch.randn(64, 27, device='cuda', generator=g) * 0.1
b = torch.randn(64, device='cuda', generator=g) * 0.1
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.conv1_3x3_relu_bf16(x, w, b, split=False)
    ops.conv1_3x3_relu_bf16(x, w, b, split=True, il=True)
    ops.conv1_3x3_relu(x, w, b) if hasattr(ops, "conv1_3x3_relu") else None
torch.cuda.synchronize()
print("done")
