"""A/B of the 4-wave (one wave per SIMD, 128x128 register tiles) bf16 GEMM against the 8-wave kernel (EXPERIMENTS build):
    NAFAE_LIB=nafae_amd/csrc/libnafae_hip_exp.so python scripts/gemm4_ab.py
Runs both arms as subprocesses (NAFAE_GEMM4=0/1): checks the small shape bit for bit across arms via a checksum, times fc6 / fc7."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "arm":
    sys.path.insert(0, ROOT)
    import torch
    from nafae_amd import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    def run(M, N, K, split, iters):
        A = torch.relu(torch.randn(M, K, device='cuda', generator=g)); B = torch.randn(N, K, device='cuda', generator=g) * 0.01
        bias = torch.randn(N, device='cuda', generator=g)
        Xp, Wp = (ops.split_bf16(A, True, True), ops.split_bf16(B, True, True)) if split else (ops.split_bf16(A, False), ops.split_bf16(B, False))
        out = ops.gemm_nt_bf16(Xp, Wp, bias, act=1, want_f32=True, want_planes=False)
        ref = torch.relu(A.double() @ B.double().T + bias.double())
        err = float((out[0].double() - ref).abs().max() / ref.abs().max())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2): ops.gemm_nt_bf16(Xp, Wp, bias, act=1, want_f32=False, want_planes=True)
        e0.record()
        for _ in range(iters): ops.gemm_nt_bf16(Xp, Wp, bias, act=1, want_f32=False, want_planes=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * M * N * K
        print("  M=%d N=%d K=%d %s: %.4f ms  %.0f TF  rel.err vs fp64 %.2e  checksum %.9e" % (M, N, K, "bf16x3" if split else "bf16", ms, fl / ms / 1e9, err, float(out[0].double().sum())))
    for split in (True, False):
        run(512, 512, 1024, split, 5)
        run(8192, 4096, 25088, split, 5)
        run(8192, 4096, 4096, split, 10)
    sys.exit(0)
for arm in ("0", "1", "0", "1"):                  # 0: 8-wave kernels, 1: 4-wave kernel for both forms
    print("NAFAE_GEMM4=%s" % arm, flush=True)
    env = dict(os.environ, NAFAE_GEMM4=arm)
    subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=env, check=False)
