"""fc6 / fc7 at BASELINE C5 (19 200 rows: 1 200 tiles of 256 x 256 = 4.69 rounds on 256 CUs) through nafae_gemm_nt_ws with and without
the stream-K tail, next to the C2 / C4 shapes (exact rounds: the plain kernel either way).  HIP-event timed."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (M, N, K) in ((19200, 4096, 25088), (19200, 4096, 4096), (8192, 4096, 25088), (16384, 4096, 25088)):
    A = torch.randn(M, K, device='cuda', generator=g); B = torch.randn(N, K, device='cuda', generator=g) * 0.01
    b = torch.zeros(N, device='cuda'); out = torch.empty(M, N, device='cuda')
    t_sk = timeit(lambda: ops.gemm_nt(A, B, b, act=ops.ACT_RELU, out=out))
    t_pl = timeit(lambda: ops.gemm_nt(A, B, b, act=ops.ACT_RELU, out=out, use_workspace=False))
    fl = 2.0 * M * N * K
    print("%dx%dx%d: with workspace %.3f ms (%.1f TF, %.3f of 157.3) | plain %.3f ms (%.3f)" % (M, N, K, t_sk, fl / t_sk / 1e9, fl / t_sk / 1e9 / 157.3, t_pl, fl / t_pl / 1e9 / 157.3))
    del A, B, out
