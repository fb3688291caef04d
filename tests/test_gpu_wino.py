"""Winograd F(2x2,3x3) fp32 conv (nafae_conv3x3_wino, csrc/wino.hip) against the fp64 convolution of the same operands and against the
direct implicit-GEMM kernel (nafae_conv3x3_relu_ws).  Replaces the reference's cuDNN conv layers (vgg16_rpn.py:38, rpn/rpn.py:63).

Tolerance: every operation is fp32; Winograd's transforms change the summation order and add a few roundings, so the bar is
2e-5 of the layer's largest |output| (measured 1e-6 .. 4e-6), next to the direct kernel's own error on the same case.  Every case
is launched three times and the three results must be bit-identical (no atomics: a difference means a staging race).

Shapes cover: both strip widths (8 and 7 tiles), partial strips (tile columns beyond the frame), row groups that change inside a
workgroup (frame / strip changes: the 14^2 and 28^2 VGG layers), tile counts that are not a multiple of 64 (masked tail), one to
many cout blocks, Cin 64 .. 512, the fused 2x2 max-pool, ReLU on / off, and every VGG16 layer shape at a reduced frame count."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
REPEAT = 3


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from nafae_amd import ops as _ops
    return _ops


def _ref64(x, w, b, relu, pool):
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1)
    if relu:
        y = torch.relu(y)
    if pool:
        y = torch.nn.functional.max_pool2d(y, 2, 2)
    return y.permute(0, 2, 3, 1).contiguous()


def _run(ops, F, H, W, Cin, Cout, relu, pool, seed, use_workspace=True):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(F, H, W, Cin, device='cuda', generator=g)
    if seed & 1:
        x = torch.relu(x)
    w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda', generator=g) * 0.1
    assert ops.wino_supported(F, H, W, Cin, Cout)
    U = ops.conv3x3_wino_pack(w)
    ys = [ops.conv3x3_wino(x, U, b, Cout, relu=relu, pool=pool, use_workspace=use_workspace) for _ in range(REPEAT)]
    torch.cuda.synchronize()
    for y in ys[1:]:
        assert torch.equal(ys[0], y), "results differ between launches"
    ref = _ref64(x, w, b, relu, pool)
    scale = float(ref.abs().max())
    err = float((ys[0].double() - ref).abs().max()) / scale
    yd = ops.conv3x3_relu(x, w, b, relu=relu, pool=pool)
    err_d = float((yd.double() - ref).abs().max()) / scale
    return err, err_d


# (F, H, W, Cin, Cout): VGG16 layers 1_2 .. 5_3 + RPN at a reduced frame count (strip 8: 224, 112; strip 7: 56, 28, 14)
VGG = [(2, 224, 224, 64, 64), (3, 112, 112, 64, 128), (3, 112, 112, 128, 128), (5, 56, 56, 128, 256), (5, 56, 56, 256, 256),
       (9, 28, 28, 256, 512), (9, 28, 28, 512, 512), (33, 14, 14, 512, 512)]


@pytest.mark.parametrize("shape", VGG, ids=lambda s: "F%d_%dx%d_%d-%d" % s)
@pytest.mark.parametrize("pool", [False, True], ids=["", "pool"])
def test_wino_vgg_layers(ops, shape, pool):
    F, H, W, Cin, Cout = shape
    err, err_d = _run(ops, F, H, W, Cin, Cout, True, pool, 1234 + H + Cin)
    print("\n[wino %dx%d %d->%d F=%d%s] max err / max|y|: winograd %.2e, direct %.2e" % (H, W, Cin, Cout, F, " +pool" if pool else "", err, err_d))
    assert err < 2e-5


def _cases(n, seed):
    rs = random.Random(seed)
    out = []
    for _ in range(n):
        Cin = rs.choice([64, 96, 128, 256]); Cout = rs.choice([64, 128, 192, 256])
        H = rs.choice([8, 10, 12, 14, 16, 20, 28, 30, 32, 48, 56]); W = rs.choice([8, 10, 14, 16, 18, 22, 28, 32, 36, 48, 56, 60])
        F = rs.choice([1, 2, 3, 5, 8, 17])
        if F * H * W * max(Cin, Cout) > 3.0e7:
            F = max(1, int(3.0e7 / (H * W * max(Cin, Cout))))
        out.append((F, H, W, Cin, Cout, rs.random() < 0.7, rs.random() < 0.4))
    return out


@pytest.mark.parametrize("case", _cases(40, 20261004),
                         ids=lambda c: "F%d_%dx%d_%d-%d%s%s" % (c[0], c[1], c[2], c[3], c[4], "_relu" if c[5] else "", "_pool" if c[6] else ""))
@pytest.mark.parametrize("ws", [True, False], ids=["streamk", "plain"])
def test_wino_random_shapes(ops, case, ws):
    F, H, W, Cin, Cout, relu, pool = case
    err, err_d = _run(ops, F, H, W, Cin, Cout, relu, pool, F * 131 + H * 17 + W * 3 + Cin, use_workspace=ws)
    assert err < 2e-5, "winograd %.2e (direct %.2e)" % (err, err_d)


# tile counts that leave a partial last round on 256 CUs (units = tile groups x cout blocks): the stream-K tail with one, a few and many
# pieces per unit, pieces that straddle two units, and workgroups without a piece
@pytest.mark.parametrize("shape", [(17, 56, 56, 64, 128), (9, 28, 28, 512, 512), (64, 14, 14, 512, 512), (21, 28, 28, 128, 256),
                                   (3, 112, 112, 64, 64), (40, 14, 14, 256, 512)], ids=lambda s: "F%d_%dx%d_%d-%d" % s)
@pytest.mark.parametrize("pool", [False, True], ids=["", "pool"])
def test_wino_streamk_tail(ops, shape, pool):
    F, H, W, Cin, Cout = shape
    from nafae_amd import _lib
    nws = int(_lib.lib().nafae_conv3x3_wino_workspace_bytes(F, H, W, Cin, Cout))
    err, err_d = _run(ops, F, H, W, Cin, Cout, True, pool, 77 + F + Cin, use_workspace=True)
    err_p, _ = _run(ops, F, H, W, Cin, Cout, True, pool, 77 + F + Cin, use_workspace=False)
    print("\n[wino stream-K %dx%d %d->%d F=%d%s] workspace %d B | err stream-K %.2e, plain %.2e, direct %.2e" %
          (H, W, Cin, Cout, F, " +pool" if pool else "", nws, err, err_p, err_d))
    assert err < 2e-5 and err_p < 2e-5


def test_wino_rejects_unsupported(ops):
    assert not ops.wino_supported(4, 15, 16, 64, 64)      # odd height
    assert not ops.wino_supported(4, 16, 16, 32, 64)      # Cin < 64
    assert not ops.wino_supported(4, 16, 16, 64, 96)      # Cout % 64
    assert not ops.wino_supported(4, 6, 16, 64, 64)       # fewer than 4 tile rows
    x = torch.zeros(1, 15, 16, 64, device='cuda')
    U = torch.zeros(16 * 64 * 64, device='cuda')
    with pytest.raises(Exception):
        ops.conv3x3_wino(x, U, torch.zeros(64, device='cuda'), 64)


# ---- size-independent properties at BASELINE C2's full sizes (64 frames): no reference needed, every pixel and channel checked
FULL = [(64, 224, 224, 64, 64), (64, 56, 56, 256, 256), (64, 28, 28, 512, 512), (64, 14, 14, 512, 512)]


@pytest.mark.parametrize("shape", FULL, ids=lambda s: "F%d_%dx%d_%d-%d" % s)
def test_wino_full_size_zero_input_gives_the_bias_exactly(ops, shape):
    """conv(0) + b = b: every product and every transform sum is an exact zero, so ReLU(bias) must come out BIT FOR BIT at every pixel
    of every frame -- borders, strip edges, partial tile groups and stream-K pieces included (a stale LDS row or a wrong padding lane
    would show up as a non-bias value)."""
    F, H, W, Cin, Cout = shape
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.zeros(F, H, W, Cin, device='cuda')
    w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g)
    b = torch.randn(Cout, device='cuda', generator=g)
    U = ops.conv3x3_wino_pack(w)
    for pool in (False, True):
        y = ops.conv3x3_wino(x, U, b, Cout, relu=True, pool=pool)
        assert y.shape == ((F, H // 2, W // 2, Cout) if pool else (F, H, W, Cout))
        assert torch.equal(y, torch.relu(b).expand_as(y))


@pytest.mark.parametrize("shape", FULL[1:], ids=lambda s: "F%d_%dx%d_%d-%d" % s)
def test_wino_full_size_linearity(ops, shape):
    """Without ReLU and bias the layer is linear: conv(2 x - 0.5 z) = 2 conv(x) - 0.5 conv(z) up to fp32 rounding, at 64 frames (scaling
    by powers of two is exact, so the only difference is the rounding of the sums: bar 2e-5 of the largest output)."""
    F, H, W, Cin, Cout = shape
    g = torch.Generator(device='cuda').manual_seed(6)
    x = torch.randn(F, H, W, Cin, device='cuda', generator=g)
    z = torch.randn(F, H, W, Cin, device='cuda', generator=g)
    w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.zeros(Cout, device='cuda')
    U = ops.conv3x3_wino_pack(w)
    yx = ops.conv3x3_wino(x, U, b, Cout, relu=False)
    yz = ops.conv3x3_wino(z, U, b, Cout, relu=False)
    yc = ops.conv3x3_wino(2.0 * x - 0.5 * z, U, b, Cout, relu=False)
    err = float((yc - (2.0 * yx - 0.5 * yz)).abs().max()) / float(yc.abs().max())
    print("\n[wino linearity %dx%d %d->%d F=%d] max deviation / max|y| %.2e" % (H, W, Cin, Cout, F, err))
    assert err < 2e-5


def test_wino_impulse_response_is_the_kernel(ops):
    """One 1.0 in an otherwise zero input: the output around it is the (flipped) 3x3 kernel of every output channel, everything else
    zero -- at frame corners, edges, strip boundaries (x = 13 / 14 / 15 / 16 with 7- and 8-tile strips) and in the interior."""
    F, H, W, Cin, Cout = 3, 32, 32, 64, 64
    g = torch.Generator(device='cuda').manual_seed(7)
    w = torch.randn(Cout, 3, 3, Cin, device='cuda', generator=g)
    b = torch.zeros(Cout, device='cuda')
    U = ops.conv3x3_wino_pack(w)
    for (f, y0, x0, c) in ((0, 0, 0, 0), (1, 31, 31, 63), (2, 0, 31, 5), (0, 31, 0, 17), (1, 13, 13, 1), (2, 14, 15, 33), (0, 15, 16, 40), (1, 7, 20, 9)):
        x = torch.zeros(F, H, W, Cin, device='cuda')
        x[f, y0, x0, c] = 1.0
        y = ops.conv3x3_wino(x, U, b, Cout, relu=False)
        want = torch.zeros_like(y)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                yy, xx = y0 + dy, x0 + dx                  # output pixel that sees the impulse through tap (1 - dy, 1 - dx)
                if 0 <= yy < H and 0 <= xx < W:
                    want[f, yy, xx, :] = w[:, 1 - dy, 1 - dx, c]
        assert float((y - want).abs().max()) <= 2e-6 * float(w.abs().max()), (f, y0, x0, c)
