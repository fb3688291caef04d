"""Micro-benchmark of the bf16 / bf16x3 GEMM at the fc6 shape (and a conv layer), HIP-event timed.
usage: python scripts/bench_gemm_bf16.py [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=it):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M, N, K = 8192, 4096, 25088
A = torch.randn(M, K, device='cuda', generator=g); B = torch.randn(N, K, device='cuda', generator=g) * 0.01
bias = torch.zeros(N, device='cuda')
for split, il in ((True, True), (True, False), (False, False)):
    Xp, Wp = ops.split_bf16(A, split, il), ops.split_bf16(B, split, il)
    nprod = 3 if split else 1
    for act, tag in ((1, 'real'), (-1, 'zero-page loads'), (-2, 'L2-resident loads')):
        ms = timeit(lambda: ops.gemm_nt_bf16(Xp, Wp, bias, act=act, want_f32=False, want_planes=True))
        print("gemm fc6 split=%s il=%s %-16s %.3f ms  alg %.0f TF  mfma %.0f TF (%.1f%% of 2.5PF)" % (split, il, tag, ms, 2*M*N*K/ms/1e9, nprod*2*M*N*K/ms/1e9, nprod*2*M*N*K/ms/1e9/25))
    del Xp, Wp
del A, B
for (F, H, Cin, Cout) in ((64, 224, 64, 64), (64, 112, 128, 128), (64, 56, 256, 256), (64, 28, 512, 512), (64, 14, 512, 512)):
    x = torch.randn(F, H, H, Cin, device='cuda', generator=g); w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * 0.02
    cb = torch.zeros(Cout, device='cuda')
    fl = 2.0 * F * H * H * Cout * 9 * Cin
    for split in (True, False):
        xp, wp = ops.split_bf16(x, split, split), ops.split_bf16(w, split, split)
        ms = timeit(lambda: ops.conv3x3_bf16(xp, wp, cb))
        nprod = 3 if split else 1
        print("conv %dx%d %d->%d split=%s %.3f ms alg %.0f TF mfma %.0f TF (%.1f%%)" % (H, H, Cin, Cout, split, ms, fl/ms/1e9, nprod*fl/ms/1e9, nprod*fl/ms/1e9/25))
    ms = timeit(lambda: ops.conv3x3_relu(x, w, cb))
    print("conv %dx%d %d->%d fp32-MFMA %.3f ms alg %.0f TF (%.1f%% of 157)" % (H, H, Cin, Cout, ms, fl/ms/1e9, fl/ms/1e9/1.573))
