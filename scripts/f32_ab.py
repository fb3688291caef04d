"""A/B of the fp32 tile kernels (EXPERIMENTS build): double-buffered LDS vs the single-buffer / 3-workgroups-per-CU variant.
    python -m nafae_amd.build --experiments
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/f32_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nafae_amd import ops

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best

g = torch.Generator(device="cuda").manual_seed(1)
cases = [("fc6", None)] + [("conv", c) for c in [(64, 224, 64, 64), (64, 112, 128, 128), (64, 56, 256, 256), (64, 28, 512, 512), (64, 14, 512, 512)]]
for kind, c in cases:
    if kind == "fc6":
        A = torch.randn(8192, 25088, device="cuda", generator=g); B = torch.randn(4096, 25088, device="cuda", generator=g) * 0.01
        bias = torch.zeros(4096, device="cuda")
        fn = lambda: ops.gemm_nt(A, B, bias, act=1)
        fl = 2.0 * 8192 * 25088 * 4096
        name = "fc6 8192x25088x4096"
    else:
        F, H, Cin, Cout = c
        x = torch.randn(F, H, H, Cin, device="cuda", generator=g); w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * 0.02
        b = torch.zeros(Cout, device="cuda")
        fn = lambda: ops.conv3x3_relu(x, w, b)
        fl = 2.0 * F * H * H * Cout * 9 * Cin
        name = "conv %d->%d @%d^2" % (Cin, Cout, H)
    res = {}
    outs = {}
    for sb in ("0", "1"):
        os.environ["NAFAE_F32_SB"] = sb
        res[sb] = timeit(fn)
        outs[sb] = fn().clone()
    if kind == "conv" and c[1] == 14:
        os.environ["NAFAE_F32_SB"] = "1"
        for smv in ("0", "1"):
            os.environ["NAFAE_F32_CONV_SMALL"] = smv
            t = timeit(fn)
            o = fn()
            print("    64x64 tiles=%s: %.3f ms (%.1f TF)  max|diff| vs 128x128 %.2e" % (smv, t, fl / t / 1e9, float((o - outs["1"]).abs().max())))
        os.environ.pop("NAFAE_F32_CONV_SMALL")
    same = torch.equal(outs["0"], outs["1"])
    print("%-26s double %.3f ms (%.1f TF)   single-buffer %.3f ms (%.1f TF)   ratio %.3f  identical=%s"
          % (name, res["0"], fl / res["0"] / 1e9, res["1"], fl / res["1"] / 1e9, res["1"] / res["0"], same))
    del outs
