"""Randomised shape sweep of the bf16x3 / plain-bf16 conv entry (all schedules) against torch's fp32 conv."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
g = torch.Generator(device='cuda').manual_seed(1)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    Cin = random.choice([32, 64, 96, 128, 256, 512]); Cout = random.choice([64, 128, 192, 256, 512])
    H = random.choice([14, 16, 20, 28, 32, 48, 56, 64, 112]); W = random.choice([14, 16, 24, 28, 32, 48, 56, 80, 112])
    F = random.choice([1, 2, 3, 5, 8, 17, 33, 64])
    if F * H * W * max(Cin, Cout) > 2.5e8:
        F = max(1, int(2.5e8 / (H * W * max(Cin, Cout))))
    split = random.random() < 0.7
    pool = random.random() < 0.4 and H % 2 == 0 and W % 2 == 0
    relu = random.random() < 0.8
    x = torch.randn(F, H, W, Cin, device='cuda', generator=g); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * (1.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda', generator=g)
    il = split and random.random() < 0.85
    xp, wp = ops.split_bf16(x, split, il), ops.split_bf16(w, split, il)
    xr, wr = ops.merge_bf16(xp), ops.merge_bf16(wp)
    ref = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), b, padding=1)
    if relu: ref = torch.relu(ref)
    if pool: ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    ref = ref.permute(0, 2, 3, 1)
    try:
        if pool:
            _, p = ops.conv3x3_bf16(xp, wp, b, relu=relu, pool=True)
        else:
            _, p = ops.conv3x3_bf16(xp, wp, b, relu=relu)
        out = ops.merge_bf16(p)
        err = float((out - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        tol = 5e-5 if split else 8e-3
        ok = err <= tol
    except Exception as e:
        ok, err = False, repr(e)
    if not ok:
        bad += 1
    print("%s F=%d H=%d W=%d Cin=%d Cout=%d split=%s il=%s relu=%s pool=%s err=%s" % ("ok " if ok else "BAD", F, H, W, Cin, Cout, split, il, relu, pool, err))
print("bad:", bad)
