"""Randomised shape sweep of the exact-fp32 conv entry (tile kernel with buffer loads, stream-K / split schedule, fused pool) and
of the fp32 GEMM (full-tile fast path and the general kernel) against fp64 references.  python scripts/stress_conv_f32.py SEED N"""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
g = torch.Generator(device='cuda').manual_seed(2)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(N):
    Cin = random.choice([32, 64, 96, 128, 256, 512]); Cout = random.choice([4, 36, 64, 128, 132, 192, 256, 512])
    H = random.choice([2, 6, 14, 16, 20, 28, 32, 48, 56, 64, 112]); W = random.choice([2, 5, 14, 16, 24, 28, 32, 48, 56, 80, 112])
    F = random.choice([1, 2, 3, 5, 8, 17, 33, 64])
    if F * H * W * max(Cin, Cout) > 1.2e8:
        F = max(1, int(1.2e8 / (H * W * max(Cin, Cout))))
    pool = random.random() < 0.4 and H % 2 == 0 and W % 2 == 0
    relu = random.random() < 0.8
    ws = random.random() < 0.7
    x = torch.randn(F, H, W, Cin, device='cuda', generator=g); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * (1.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda', generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1)
    if relu: ref = torch.relu(ref)
    if pool: ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    ref = ref.permute(0, 2, 3, 1)
    try:
        out = ops.conv3x3_relu(x, w, b, relu=relu, use_workspace=ws, pool=pool)
        out2 = ops.conv3x3_relu(x, w, b, relu=relu, use_workspace=ws, pool=pool)
        err = float((out.double() - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        ok = err <= 2e-5 and torch.equal(out, out2)
    except Exception as e:
        ok, err = False, repr(e)
    bad += (not ok)
    print("%s conv F=%d H=%d W=%d Cin=%d Cout=%d relu=%s pool=%s ws=%s err=%s" % ("ok " if ok else "BAD", F, H, W, Cin, Cout, relu, pool, ws, err))
for it in range(N // 2):
    M = random.choice([1, 7, 128, 200, 256, 384, 1000, 2048]); Nn = random.choice([4, 54, 64, 128, 192, 256, 512, 1000]); K = random.choice([32, 64, 100, 512, 1024, 4096, 12288])
    A = torch.randn(M, K, device='cuda', generator=g); B = torch.randn(Nn, K, device='cuda', generator=g) * K ** -0.5; b = torch.randn(Nn, device='cuda', generator=g)
    act = random.choice([ops.ACT_NONE, ops.ACT_RELU])
    ref = A.double() @ B.double().T + b.double()
    if act == ops.ACT_RELU: ref = torch.relu(ref)
    try:
        out = ops.gemm_nt(A, B, b, act=act)
        err = float((out.double() - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        ok = err <= 2e-5
    except Exception as e:
        ok, err = False, repr(e)
    bad += (not ok)
    print("%s gemm M=%d N=%d K=%d act=%d err=%s" % ("ok " if ok else "BAD", M, Nn, K, act, err))
print("bad:", bad)
