"""HIP-event times of single bf16x3 (PREC=bf16: plain bf16) conv layers at BASELINE C2 shapes (64 frames), one line per layer; used to
compare library variants (NAFAE_LIB=...).  usage: [PREC=bf16] python3 scripts/conv_times.py [layer ...]   layers: c12 c21 c22 c31 c32 c41 c42 c5"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
L = {"c12": (224, 64, 64, True), "c21": (112, 64, 128, False), "c22": (112, 128, 128, True), "c31": (56, 128, 256, False),
     "c32": (56, 256, 256, False), "c41": (28, 256, 512, False), "c42": (28, 512, 512, False), "c5": (14, 512, 512, False)}
g = torch.Generator(device='cuda').manual_seed(0)
SPLIT = os.environ.get('PREC') != 'bf16'
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for name in (sys.argv[1:] or list(L)):
    H, Cin, Cout, pool = L[name]
    x = torch.relu(torch.randn(64, H, H, Cin, device='cuda', generator=g)); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    xp, wp = ops.split_bf16(x, SPLIT, SPLIT), ops.split_bf16(w, SPLIT, SPLIT); cb = torch.zeros(Cout, device='cuda')
    ms = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb, pool=pool))
    fl = (3 if SPLIT else 1) * 2.0 * 64 * H * H * Cout * 9 * Cin
    out.append("%s %.3f ms (%.0f%%)" % (name, ms, fl / ms / 1e9 / 25))
    del x, w, xp, wp
print(os.path.basename(os.environ.get("NAFAE_LIB", "default")), "bf16x3" if SPLIT else "bf16", " | ".join(out))
