"""Per-layer times of the exact-fp32 conv stack at BASELINE C2 (64 frames 224^2), HIP-event timed one layer at a time, with the
share of the fp32-MFMA peak (157.3 TF) and the tile count per CU (the 128 x 128 / 128 x 64 tile kernels run one tile per workgroup,
so a layer's time steps at whole tiles per CU)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
F = 64; tot = 0.0
x0 = torch.randn(F, 3, 224, 224, device='cuda', generator=g); w0 = torch.randn(64, 27, device='cuda', generator=g) * 0.1; b0 = torch.zeros(64, device='cuda')
ms = timeit(lambda: ops.conv1_3x3_relu(x0, w0, b0)); tot += ms
print("conv1_1 3->64 @224: %.3f ms (writes 822 MB: %.2f TB/s)" % (ms, 0.822 / ms))
y0 = ops.conv1_3x3_relu(x0, w0, b0)
ms = timeit(lambda: ops.maxpool2x2(y0)); print("pool1 (fp32, 822 MB in, 205 MB out): %.3f ms" % ms)
del x0, y0
for (name, H, Cin, Cout, n) in (("conv1_2", 224, 64, 64, 1), ("conv2_1", 112, 64, 128, 1), ("conv2_2", 112, 128, 128, 1), ("conv3_1", 56, 128, 256, 1),
                                ("conv3_2", 56, 256, 256, 2), ("conv4_1", 28, 256, 512, 1), ("conv4_2", 28, 512, 512, 2), ("conv5_x+rpn", 14, 512, 512, 4)):
    x = torch.relu(torch.randn(F, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    cb = torch.zeros(Cout, device='cuda')
    ms = timeit(lambda: ops.conv3x3_relu(x, w, cb))
    fl = 2.0 * F * H * H * Cout * 9 * Cin
    bw = 64 if Cout <= 64 else 128
    tiles = (F * H * H + 127) // 128 * ((Cout + bw - 1) // bw)
    tot += n * ms
    print("%s %d->%d @%d: %.3f ms x%d  %.1f TF (%.0f%% of 157.3)  tiles %d = %.2f per CU" % (name, Cin, Cout, H, ms, n, fl / ms / 1e9, fl / ms / 1e9 / 1.573, tiles, tiles / 256.0))
    del x, w
print("sum %.3f ms (with conv1_1, without the pools)" % tot)
