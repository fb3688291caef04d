"""Where a Winograd conv result differs from the fp64 reference: error by output channel / pixel (debugging aid)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from nafae_amd import ops
F, H, W, Cin, Cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 16, 16, 64, 64))]
pool = len(sys.argv) > 6 and sys.argv[6] == "pool"
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(F, H, W, Cin, device='cuda', generator=g); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.05
b = torch.randn(Cout, device='cuda', generator=g)
U = ops.conv3x3_wino_pack(w)
y = ops.conv3x3_wino(x, U, b, Cout, relu=False, pool=pool)
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1)
if pool:
    ref = torch.nn.functional.max_pool2d(ref, 2, 2)
ref = ref.permute(0, 2, 3, 1)
e = (y.double() - ref).abs()
print("max err", float(e.max()), "scale", float(ref.abs().max()))
print("per channel max err:", [round(float(v), 3) for v in e.amax(dim=(0, 1, 2))])
print("per (y,x) max err of frame 0:\n", e[0].amax(dim=2).cpu().numpy().round(2))
# which reference value does each output match?
yy = y[0, 0, 0].double(); 
print("out[0,0,0,:8]", yy[:8].cpu().numpy().round(3)); print("ref[0,0,0,:8]", ref[0, 0, 0, :8].cpu().numpy().round(3))
