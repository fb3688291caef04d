"""Where does bf16_conv4_kernel differ from the fp32 conv?  Error maps for small shapes (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nafae_amd import ops
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
def run(F, H, W, Cin, Cout, split, ws):
    g = torch.Generator(device="cuda").manual_seed(F + 3 * H + Cin)
    x = torch.randn(F, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * (1.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    xp, wp = ops.split_bf16(x, split, split), ops.split_bf16(w, split, split)
    f, _ = ops.conv3x3_bf16(xp, wp, b, relu=False, want_f32=True, use_workspace=ws)
    xr, wr = (x, w) if split else (ops.merge_bf16(xp), ops.merge_bf16(wp))
    ref = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    err = (f - ref).abs()
    scale = float(ref.abs().max())
    print("F%d %dx%d %d->%d split=%d ws=%d: max err %.3g (scale %.3g)" % (F, H, W, Cin, Cout, split, ws, float(err.max()), scale))
    if float(err.max()) > 1e-3 * scale:
        pix = err.amax(dim=3)                      # [F, H, W]
        print(" per-pixel max err, frame 0:\n", pix[0])
        if F > 1:
            print(" frame %d:\n" % (F - 1), pix[F - 1])
        ch = err.amax(dim=(0, 1, 2)).view(-1, 32).amax(1)
        print(" per 32-channel block:", ch)
        bad = (pix.flatten() > 1e-3 * scale).nonzero().flatten()
        print(" bad pixels: %d of %d; first %s last %s" % (len(bad), pix.numel(), bad[:8].tolist(), bad[-8:].tolist()))
for split in (True, False):
    for ws in (False, True):
        run(3, 14, 14, 256, 512, split, ws)
run(4, 14, 14, 256, 512, False, False)
run(3, 16, 16, 256, 512, False, False)     # M = 768: full tiles
run(3, 14, 14, 512, 512, False, False)
run(3, 14, 14, 256, 256, False, False)
run(5, 9, 11, 320, 256, False, False)
run(5, 9, 11, 320, 256, True, False)
run(1, 7, 5, 256, 256, True, False)
run(2, 33, 17, 384, 768, True, True)
run(2, 33, 17, 384, 768, False, True)
