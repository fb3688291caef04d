"""Randomised shape sweep of nafae_gemm_nt_bf16 (split interleaved / split separate / plain) against torch fp32."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
g = torch.Generator(device='cuda').manual_seed(2)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    M = random.choice([1, 7, 64, 100, 256, 300, 777, 1024, 2500, 8192]); N = random.choice([4, 64, 72, 128, 200, 256, 512, 1000, 4096])
    K = random.choice([32, 64, 96, 200, 512, 1000, 4096, 25088])
    if M * K > 6e7: M = max(1, int(6e7 / K))
    split = random.random() < 0.7
    il = split and K % 32 == 0 and N % 32 == 0 and random.random() < 0.8
    if K % 8: continue
    A = torch.randn(M, K, device='cuda', generator=g); B = torch.randn(N, K, device='cuda', generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device='cuda', generator=g)
    Ap, Bp = ops.split_bf16(A, split, il), ops.split_bf16(B, split, il)
    ref = torch.relu((ops.merge_bf16(Ap).double() @ ops.merge_bf16(Bp).double().t()).float() * 0.5 + bias)
    try:
        f, p = ops.gemm_nt_bf16(Ap, Bp, bias, alpha=0.5, act=ops.ACT_RELU, want_f32=True, want_planes=(N % 4 == 0))
        err = float((f - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
        ok = err <= (5e-5 if split else 2e-5)      # plain: exact bf16 products, fp32 accumulation
        if p is not None:
            e2 = float((ops.merge_bf16(p) - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
            ok = ok and e2 <= (5e-5 if split else 8e-3)
    except Exception as e:
        ok, err = False, repr(e)
    if not ok: bad += 1
    print("%s M=%d N=%d K=%d split=%s il=%s err=%s" % ("ok " if ok else "BAD", M, N, K, split, il, err))
print("bad:", bad)
