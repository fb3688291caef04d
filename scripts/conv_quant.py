"""Tile-count quantisation probe: one bf16x3 conv layer timed against the frame count (tiles = pixels / 256)."""
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
for (H, Cin, Cout, bw) in ((56, 256, 256, 256), (28, 512, 512, 256), (14, 512, 512, 128)):
    w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    wp = ops.split_bf16(w, True, True); cb = torch.zeros(Cout, device='cuda')
    for F in (40, 48, 56, 60, 62, 63, 64, 66, 72, 80, 83, 84, 96, 128):
        x = torch.randn(F, H, H, Cin, device='cuda', generator=g)
        xp = ops.split_bf16(x, True, True)
        ms = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb))
        tiles = -(-F * H * H // 256) * (Cout // bw)
        fl = 3 * 2.0 * F * H * H * Cout * 9 * Cin
        print("conv %dx%d %d->%d F=%3d tiles=%4d rounds=%.2f  %.3f ms  %.0f TF mfma  us/round-up=%.1f" % (H, H, Cin, Cout, F, tiles, tiles / 256, ms, fl / ms / 1e9, 1e3 * ms / -(-tiles // 256)))
        del x, xp
