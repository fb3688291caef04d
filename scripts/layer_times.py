"""Per-layer conv times of the detector (bf16x3, or plain bf16 with argv[1] == bf16), HIP-event timed one layer at a time at BASELINE
C2 / C3 (64 frames 224^2)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
F = 64
SPLIT = not (len(sys.argv) > 1 and sys.argv[1] == 'bf16')
tot = 0.0
x0 = torch.relu(torch.randn(F, 3, 224, 224, device='cuda', generator=g))
w0 = torch.randn(64, 27, device='cuda', generator=g) * 0.1; b0 = torch.zeros(64, device='cuda')
ms = timeit(lambda: ops.conv1_3x3_relu_bf16(x0, w0, b0, split=SPLIT, il=SPLIT)); tot += ms
print("conv1_1 3->64 @224: %.3f ms" % ms)
for (name, H, Cin, Cout, pool) in (("conv1_2", 224, 64, 64, True), ("conv2_1", 112, 64, 128, False), ("conv2_2", 112, 128, 128, True),
                                   ("conv3_1", 56, 128, 256, False), ("conv3_2", 56, 256, 256, False), ("conv3_3", 56, 256, 256, True),
                                   ("conv4_1", 28, 256, 512, False), ("conv4_2", 28, 512, 512, False), ("conv4_3", 28, 512, 512, True),
                                   ("conv5_x", 14, 512, 512, False), ("rpn", 14, 512, 512, False)):
    x = torch.relu(torch.randn(F, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    xp, wp = ops.split_bf16(x, SPLIT, SPLIT), ops.split_bf16(w, SPLIT, SPLIT); cb = torch.zeros(Cout, device='cuda')
    ms = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb))
    fl = (3 if SPLIT else 1) * 2.0 * F * H * H * Cout * 9 * Cin
    n = 3 if name == "conv5_x" else 1
    tot += n * ms
    line = "%s %d->%d @%d: %.3f ms x%d  %.0f TF mfma (%.0f%% of 2.5 PF)" % (name, Cin, Cout, H, ms, n, fl / ms / 1e9, fl / ms / 1e9 / 25)
    if pool:
        _, y = ops.conv3x3_bf16(xp, wp, cb)
        mp = timeit(lambda: ops.maxpool2x2_bf16(y))
        fused = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb, pool=True))
        tot += fused - ms
        line += "   + pool %.3f ms (conv+pool in one call: %.3f ms)" % (mp, fused)
    print(line)
    del x, w, xp, wp
print("sum %.3f ms" % tot)
