"""Per-layer times of the exact-fp32 conv stack at BASELINE C2 (64 frames 224^2): Winograd F(2x2,3x3) kernel (nafae_conv3x3_wino)
next to the direct implicit-GEMM kernel (nafae_conv3x3_relu_ws), HIP-event timed one layer at a time.  TF = direct-conv flops per
second (2 F H W Cout 9 Cin); the Winograd kernel issues 1/2.25 of them, `issued` is its share of the fp32-MFMA peak on those."""
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
F = int(os.environ.get("F", "64")); totw = totd = 0.0
only = os.environ.get("ONLY")
for (name, H, Cin, Cout, n, pool) in (("conv1_2", 224, 64, 64, 1, True), ("conv2_1", 112, 64, 128, 1, False), ("conv2_2", 112, 128, 128, 1, True),
                                      ("conv3_1", 56, 128, 256, 1, False), ("conv3_2", 56, 256, 256, 1, False), ("conv3_3", 56, 256, 256, 1, True),
                                      ("conv4_1", 28, 256, 512, 1, False), ("conv4_2", 28, 512, 512, 1, False), ("conv4_3", 28, 512, 512, 1, True),
                                      ("conv5_x+rpn", 14, 512, 512, 4, False)):
    if only and only != name:
        continue
    x = torch.relu(torch.randn(F, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    cb = torch.zeros(Cout, device='cuda')
    U = ops.conv3x3_wino_pack(w)
    msw = timeit(lambda: ops.conv3x3_wino(x, U, cb, Cout, relu=True, pool=pool, use_workspace=os.environ.get("WS", "1") == "1"))
    msd = timeit(lambda: ops.conv3x3_relu(x, w, cb, pool=pool))
    fl = 2.0 * F * H * H * Cout * 9 * Cin
    units = ((F * (H // 2) * (H // 2) + 63) // 64) * (Cout // 64)
    totw += n * msw; totd += n * msd
    print("%s %d->%d @%d%s: winograd %.3f ms x%d  %.1f TF direct-equivalent (issued %.0f%% of 157.3)  units %d = %.2f per CU | direct %.3f ms (%.0f%%) | x%.2f"
          % (name, Cin, Cout, H, "+pool" if pool else "", msw, n, fl / msw / 1e9, fl / 2.25 / msw / 1e9 / 1.573, units, units / 256.0, msd, fl / msd / 1e9 / 1.573, msd / msw))
    del x, w, U
print("sum winograd %.3f ms, direct %.3f ms (without conv1_1)" % (totw, totd))
