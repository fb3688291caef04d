"""Launches the Winograd conv kernels alone at the BASELINE C2 layer shapes (for rocprofv3 --kernel-trace / --pmc passes).
Usage: python3 scripts/wino_only.py [iters] [layer ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
LAYERS = {"conv1_2": (224, 64, 64, True), "conv2_2": (112, 128, 128, True), "conv3_2": (56, 256, 256, False), "conv4_2": (28, 512, 512, False),
          "conv5_1": (14, 512, 512, False)}
names = sys.argv[2:] or list(LAYERS)
g = torch.Generator(device='cuda').manual_seed(0)
F = 64
for name in names:
    H, Cin, Cout, pool = LAYERS[name]
    x = torch.relu(torch.randn(F, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    b = torch.zeros(Cout, device='cuda')
    U = ops.conv3x3_wino_pack(w)
    for _ in range(it):
        ops.conv3x3_wino(x, U, b, Cout, pool=pool)
    torch.cuda.synchronize()
    del x, w, U
print("done")
