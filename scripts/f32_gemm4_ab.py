"""A/B of the 4-wave fp32 GEMM (one wave per SIMD, 128x128 register tiles, LDS-DMA) against the 128x128-tile kernel (EXPERIMENTS
build):  NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/f32_gemm4_ab.py
Both arms run as subprocesses (NAFAE_F32_GEMM4=0/1); the integer checksum of the output bits must agree between them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "arm":
    sys.path.insert(0, ROOT)
    import torch
    from nafae_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    def run(M, N, K, iters, act):
        A = torch.relu(torch.randn(M, K, device='cuda', generator=g)); B = torch.randn(N, K, device='cuda', generator=g) * 0.01
        bias = torch.randn(N, device='cuda', generator=g)
        out = ops.gemm_nt(A, B, bias, act=act)
        bits = int(out.view(torch.int32).to(torch.int64).sum())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2): ops.gemm_nt(A, B, bias, act=act)
        e0.record()
        for _ in range(iters): ops.gemm_nt(A, B, bias, act=act)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print("  M=%d N=%d K=%d act=%d: %.4f ms  %.1f TF (%.3f of 157.3)  bit checksum %d" % (M, N, K, act, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3, bits))
    run(512, 512, 96, 3, 1)          # (too few tiles: both arms take the 128x128 kernel)
    run(4096, 4096, 1056, 5, 1)      # 256 tiles of 256x256: one per CU
    run(8192, 4096, 25088, 5, 1)     # fc6
    run(8192, 4096, 4096, 10, 1)     # fc7
    run(16384, 4096, 4096, 5, 0)     # C4's fc7, no activation
    sys.exit(0)
for arm in ("0", "1", "0", "1"):
    print("NAFAE_F32_GEMM4=%s" % arm, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=dict(os.environ, NAFAE_F32_GEMM4=arm), check=False)
