"""What bounds the bf16 conv loop chip-wide?  One long-K layer (256->256 @56^2, K = 2304) run with 16 ... 256 workgroups (one 256x256
tile each, one round, no stream-K): per-CU MFMA rate against the number of busy CUs.  A kernel bound by its own instruction stream
keeps its per-CU rate; one bound by a shared budget (power -> clock, L2 fabric) loses it as CUs are added.
    [NAFAE_LIB=..._exp.so NAFAE_CONV4=0|1] python scripts/conv_occupancy.py [bf16|bf16x3]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
SPLIT = not (len(sys.argv) > 1 and sys.argv[1] == "bf16")
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
H, Cin, Cout = 16, 256, 256                 # 16 x 16 = one 256-pixel tile per frame
w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * 0.02
wp = ops.split_bf16(w, SPLIT, SPLIT); cb = torch.zeros(Cout, device="cuda")
print("%s, 256->256 conv, one 256x256x2304 tile per workgroup (NAFAE_CONV4=%s)" % ("bf16x3" if SPLIT else "bf16", os.environ.get("NAFAE_CONV4", "default")))
for tiles in (16, 64, 96, 104, 112, 120, 128, 192, 256, 512):
    x = torch.relu(torch.randn(tiles, H, H, Cin, device="cuda", generator=g))
    xp = ops.split_bf16(x, SPLIT, SPLIT)
    ms = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb, use_workspace=False))
    fl = (3 if SPLIT else 1) * 2.0 * tiles * 256 * Cout * 9 * Cin
    busy = min(tiles, 256)
    print("  %4d tiles: %.4f ms  %7.1f TF issued in all  %.2f TF per busy CU (%.0f%% of 9.77)" % (tiles, ms, fl / ms / 1e9, fl / ms / 1e9 / busy, fl / ms / 1e9 / busy / 9.77 * 100))
    del x, xp
