"""Child process of tests/test_gpu_bf16.py's A/B tests: runs the listed conv cases against the EXPERIMENTS build of the library
(NAFAE_LIB=.../libnafae_hip_exp.so, the only build that honours the NAFAE_* dispatch switches) with whatever switches the
parent put into the environment, and saves the merged fp32 outputs.  One process per arm: the library caches some switches
in function-local statics, so an arm cannot be changed inside a process.

    python tests/exp_arm_worker.py <kind: patch|pair|gemm4|conv4> <out.pt>
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PATCH_CASES = [(8, 64, 128, 64, 64), (3, 112, 112, 64, 64), (5, 48, 160, 64, 64), (4, 64, 64, 128, 128), (2, 112, 112, 64, 128),
               (9, 32, 32, 256, 64)]
PAIR_CASES = [(8, 64, 128, 64, 64), (3, 112, 112, 64, 128), (4, 64, 64, 128, 128), (40, 56, 56, 128, 256), (64, 28, 28, 256, 512),
              (6, 14, 14, 512, 512), (2, 20, 36, 64, 64)]


# (M, N, K): full 256x256 tiles, at least one per CU -- the shapes the one-wave-per-SIMD GEMMs take
GEMM4_CASES = [(4096, 4096, 512), (8192, 4096, 1056), (5120, 4096, 96), (4096, 4096, 32), (4096, 4096, 64)]   # (1, 2, 3 k-tiles too)


# (F, H, W, Cin, Cout): the long-K layers of the one-wave-per-SIMD conv -- VGG shapes (tile counts 784 / 392 / 98: cut tiles with a
# workspace), frames whose pixel count is no multiple of 256, odd widths (every tile crosses rows and frame seams), one tile only
CONV4_CASES = [(64, 56, 56, 256, 256), (64, 28, 28, 512, 512), (64, 14, 14, 512, 512), (3, 14, 14, 256, 512), (5, 9, 11, 320, 256),
               (1, 7, 5, 256, 256), (2, 33, 17, 384, 768)]


def inputs(case, seed_of):
    F, H, W, Cin, Cout = case
    g = torch.Generator(device="cuda").manual_seed(seed_of(case))
    x = torch.randn(F, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * (1.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    return x, w, b


def main():
    kind, out_path = sys.argv[1], sys.argv[2]
    from nafae_amd import _lib, ops
    assert _lib.LIB_PATH.endswith("_exp.so"), _lib.LIB_PATH
    out = {}
    if kind == "patch":
        for c in PATCH_CASES:
            x, w, b = inputs(c, lambda c: c[0] * c[1] + c[2])
            xp, wp = ops.split_bf16(x, True, True), ops.split_bf16(w, True, True)
            _, p = ops.conv3x3_bf16(xp, wp, b, relu=True)
            out[c] = ops.merge_bf16(p).cpu()
    elif kind == "gemm4":
        for (M, N, K) in GEMM4_CASES:
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            A = torch.relu(torch.randn(M, K, device="cuda", generator=g))
            B = torch.randn(N, K, device="cuda", generator=g) * 0.05
            bias = torch.randn(N, device="cuda", generator=g)
            f32 = ops.gemm_nt(A, B, bias, act=1, use_workspace=False)      # whole tiles (the stream-K tail rounds cut tiles differently)
            x3 = ops.gemm_nt_bf16(ops.split_bf16(A, True, True), ops.split_bf16(B, True, True), bias, act=1, want_f32=True, want_planes=False)[0]
            pl = ops.gemm_nt_bf16(ops.split_bf16(A, False), ops.split_bf16(B, False), bias, act=0, want_f32=True, want_planes=False)[0] if K % 64 == 0 else None
            ref = torch.relu(A.double() @ B.double().T + bias.double())
            out[(M, N, K)] = (f32.cpu(), x3.cpu(), None if pl is None else pl.cpu(), float((f32.double() - ref).abs().max() / ref.abs().max()),
                              float((x3.double() - ref).abs().max() / ref.abs().max()))
    elif kind == "conv4":
        for c in CONV4_CASES:
            x, w, b = inputs(c, lambda c: c[0] + 3 * c[1] + c[3])
            res = []
            for split in (True, False):
                xp, wp = ops.split_bf16(x, split, split), ops.split_bf16(w, split, split)
                f0, p0 = ops.conv3x3_bf16(xp, wp, b, relu=True, want_f32=True, use_workspace=False)   # whole tiles only
                f1, p1 = ops.conv3x3_bf16(xp, wp, b, relu=True, want_f32=True)                          # stream-K where it applies
                _, p2 = ops.conv3x3_bf16(xp, wp, b, relu=True)
                res.append((f0.cpu(), ops.merge_bf16(p0).cpu(), f1.cpu(), ops.merge_bf16(p1).cpu(), ops.merge_bf16(p2).cpu()))
            out[c] = res
    else:
        for c in PAIR_CASES:
            x, w, b = inputs(c, lambda c: c[0] + c[1] + c[3])
            xp, wp = ops.split_bf16(x, False), ops.split_bf16(w, False)
            f, p1 = ops.conv3x3_bf16(xp, wp, b, relu=True, want_f32=True)
            _, p2 = ops.conv3x3_bf16(xp, wp, b, relu=True)
            out[c] = (f.cpu(), ops.merge_bf16(p1).cpu(), ops.merge_bf16(p2).cpu())
    torch.save(out, out_path)


if __name__ == "__main__":
    main()
