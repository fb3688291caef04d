"""Round 4: the similarity operand planes (simplanes.hip).  Producers (stand-alone and fused into the dropout + tanh epilogue of
VisEbd / WordEbd, model.py:627-628, 641-642) against torch; the many-live-column kernel fed by them against the fp64 reference and
against round 3's kernel on the fp32 operands; the adversarial cases of the one-product fp16 filter (values beyond the fp16 range,
subnormals, mass ties that overflow the candidate lists)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes_ref(X, kind):
    if kind == "f16":
        return X.half()
    hi = X.bfloat16()
    lo = (X - hi.float()).bfloat16()
    R, D = X.shape
    return torch.stack([hi.view(R, D // 32, 32), lo.view(R, D // 32, 32)], 2).reshape(R, 2 * D)


@pytest.mark.parametrize("kind", ["bf16x3", "f16"])
@pytest.mark.parametrize("shape", [(1, 64), (7, 128), (300, 512), (1000, 192), (5, 1024)])
def test_sim_planes_standalone_bits_and_stats(kind, shape):
    from nafae_amd import ops
    R, D = shape
    g = torch.Generator().manual_seed(R * 1000 + D)
    X = (torch.randn(R, D, generator=g) * torch.tensor(10.0) ** torch.randint(-6, 3, (R, 1), generator=g).float()).cuda()
    P = ops.sim_planes(X, kind)
    assert torch.equal(P.planes.view(torch.int16), _planes_ref(X, kind).view(torch.int16))
    assert torch.equal(P.stats[:, 0], X.abs().max(1)[0])
    n64 = X.double().norm(dim=1)
    assert float(((P.stats[:, 1].double() - n64).abs() / n64).max()) < 1e-6


@pytest.mark.parametrize("kind", ["bf16x3", "f16"])
def test_dropout_tanh_planes_equal_unfused(kind):
    """The plane-emitting epilogue writes the same y as the plain kernels (masked, unmasked, seeded) and the planes of that y."""
    from nafae_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(333, 512, device="cuda", generator=g) * 1.7
    mask = (torch.rand(333, 512, device="cuda", generator=g) >= 0.3).to(torch.uint8)
    for a, b in ((ops.dropout_tanh(x, None, 1.0), ops.dropout_tanh(x, None, 1.0, planes=kind)),
                 (ops.dropout_tanh(x, mask, 1.0 / 0.7), ops.dropout_tanh(x, mask, 1.0 / 0.7, planes=kind)),
                 (ops.dropout_tanh_seeded(x, 123456789, 0.2), ops.dropout_tanh_seeded(x, 123456789, 0.2, planes=kind))):
        assert torch.equal(a, b)
        P = ops.attached_sim_planes(b)
        assert P is not None and P.kind == kind and ops.attached_sim_planes(a) is None
        Pr = ops.sim_planes(b, kind)
        assert torch.equal(P.planes.view(torch.int16), Pr.planes.view(torch.int16)) and torch.equal(P.stats, Pr.stats)
    b.add_(1.0)                                   # an in-place change invalidates the attached planes
    assert ops.attached_sim_planes(b) is None


def _ref64(V, W, lens, Na, Nb, Ne):
    Q = Na * Ne
    S = V.double() @ W.double().t()
    masked = (torch.arange(Ne)[None, :] >= torch.tensor(lens)[:, None]).view(1, Q)
    S = S.masked_fill(masked, 0).view(-1, Nb, Q)
    m, i = S.max(1)
    t2 = S.topk(2, dim=1)[0]
    return m, i, t2[:, 0] - t2[:, 1], masked.expand(m.shape[0], Q)


@pytest.mark.parametrize("kind", ["bf16x3", "f16"])
@pytest.mark.parametrize("cfg", [(8, 8, 300, 64), (8, 8, 256, 32), (8, 8, 128, 16), (2, 3, 700, 40)], ids=["C5", "C4", "C2x", "Nb700"])
def test_planes_kernel_all_live_equals_round3_kernel(kind, cfg):
    """Every slot live at the BASELINE shapes (and a frame of 700 proposals: three super-tiles): planes attached by the caller ==
    planes split by the entry point's own pre-pass (fp32 operands only) bit for bit, whatever the plane kind (every route ends in the
    same exact-fp32 evaluation), and both match fp64 where fp64 is decided."""
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne = cfg
    D = 512
    lens = [Ne] * Na
    if Na * Ne <= 64:
        pytest.skip("not a many-live-column shape")
    V, W = syn.embeddings(Na * Ns * Nb, Na * Ne, D, seed=6)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    Vg, Wg = V.cuda(), W.cuda()
    S0, D0 = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne, lens=lens, planes=False)
    S1, D1 = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne, lens=lens, planes=kind)
    assert torch.equal(D0, D1) and torch.equal(S0, S1)
    m, i, gap, masked = _ref64(V, W, lens, Na, Nb, Ne)
    scale = float(m.abs().max())
    assert not ((D1.cpu() != i) & (gap > 1e-5 * scale)).any()
    assert float((S1.cpu().double() - m).abs().max()) < 2e-6 * scale


@pytest.mark.parametrize("cfg", [(8, 8, 300, 64, None), (8, 8, 256, 32, None), (2, 3, 224, 40, [1, 0]), (2, 2, 700, 64, [64, 0]),
                                 (3, 2, 333, 30, [30, 30, 4]), (1, 5, 640, 8, [8])],
                         ids=["C5hist", "C4hist", "one-live", "Nb700-64live", "two-groups", "Nb640"])
def test_narrow_planes_kernel_few_live_columns(cfg):
    """Few live columns, long frames, fp16 planes: the narrow form of the planes kernel (four row quarters x one 32-column block, ring
    of 2-3 stages) == the fp32 live-column kernel bit for bit (both end in the same exact fp32 evaluation), one and two column
    groups, one to three super-tiles, frames that end inside a row block; masked slots (0, 0)."""
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, lens = cfg
    D = 512
    lens = lens or syn.entity_lengths(Na, Ne, seed=1234)
    assert sum(lens) <= 64 and Nb >= 224
    V, W = syn.embeddings(Na * Ns * Nb, Na * Ne, D, seed=9)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    Vg, Wg = V.cuda(), W.cuda()
    S0, D0 = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne, lens=lens, planes=False)
    S1, D1 = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne, lens=lens, planes="f16")
    m, i, gap, masked = _ref64(V, W, lens, Na, Nb, Ne)
    scale = float(m.abs().max())
    assert not ((D1.cpu() != i) & (gap > 1e-5 * scale)).any()
    assert float((S1.cpu().double() - m).abs().max()) < 2e-6 * scale
    assert (S1.cpu()[masked] == 0).all() and (D1.cpu()[masked] == 0).all()
    assert torch.equal(D0, D1) and torch.allclose(S0, S1, rtol=0, atol=3e-6 * scale)
    S2, D2 = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne, lens=lens, planes="f16")
    assert torch.equal(S1, S2) and torch.equal(D1, D2)


@pytest.mark.parametrize("kind", ["bf16x3", "f16"])
def test_planes_kernel_extreme_values_and_mass_ties(kind):
    """What the one-product fp16 filter cannot represent must still come out exactly: rows scaled beyond the fp16 range (Inf in the
    planes -> the column is evaluated exactly over all rows), rows in the fp16 subnormal range (absolute rounding 2^-25), and 40
    identical winning rows (the candidate list overflows -> exact over all rows, first index wins)."""
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D = 2, 3, 200, 40, 512
    lens = [40, 31]
    V = torch.tanh(syn.randn(31, "Vx", (Na * Ns * Nb, D)))
    W = torch.tanh(syn.randn(31, "Wx", (Na * Ne, D)))
    V3 = V.view(Na * Ns, Nb, D)
    V3[1] *= 3.0e5                         # frame 1: every element beyond 65504 -> Inf in fp16
    V3[2] *= 1.0e-6                        # frame 2: fp16 subnormals / zeros
    V3[3, 50:90] = torch.tanh(W[3] * 2.0)  # frame 3: 40 identical rows that win query 3 (and others)
    V3[4, 7] *= 1.0e5                      # frame 4: one huge row among ordinary ones
    W[5] *= 1.0e-7                         # a tiny query
    m, i, gap, masked = _ref64(V, W, lens, Na, Nb, Ne)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    S, Di = ops.sim_max_fwd(V.cuda(), W.cuda(), lt, Na, Ns, Nb, Ne, lens=lens, planes=kind)
    S, Di = S.cpu().double(), Di.cpu()
    assert (S[masked] == 0).all() and (Di[masked] == 0).all()
    live = ~masked
    rel = ((S - m).abs() / m.abs().clamp_min(1e-30))[live]
    assert float(rel.max()) < 5e-6, float(rel.max())                     # fp32 dot products of the winners, frame by frame scale
    decided = gap > 1e-5 * m.abs()
    assert not ((Di != i) & live & decided).any()
    assert (Di[3, 3] == 50) and int(i[3, 3]) == 50                       # the first of the identical rows


@pytest.mark.parametrize("kind", ["bf16x3", "f16"])
def test_model_modules_hand_planes_to_dvsa(kind):
    """VisEbd / WordEbd emit the planes in their tanh epilogue, DVSA picks them up from the tensors (B1 signature unchanged) and the
    result equals the run without planes bit for bit (many-live-column shape); an in-place edit of V drops them."""
    from nafae_amd import ops
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    from nafae_amd.model import default_args, GroundModel
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    Na, Ns, Nb, Ne = 2, 2, 96, 40
    cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
    torch.manual_seed(3)
    model = GroundModel(args, cfg).cuda()
    model.train(); model.DVSA.init_train()
    lens = [40, 33]
    g = torch.Generator(device="cuda").manual_seed(8)
    fc7 = torch.randn(Na * Ns * Nb, 4096, device="cuda", generator=g).abs() * 30
    glove = torch.randn(Na * Ne, args.glove_dim, device="cuda", generator=g) * 0.4
    old = ops.SIM_PLANES_DEFAULT
    try:
        ops.SIM_PLANES_DEFAULT = kind
        out = {}
        for sp in ("auto", None):
            model.vis_ebd.sim_planes = model.word_ebd.sim_planes = sp
            V = model.vis_ebd(fc7)
            W = model.word_ebd(glove)
            P = ops.attached_sim_planes(V)
            assert (P is not None and P.kind == kind and ops.attached_sim_planes(W) is not None) if sp else P is None
            D_ind, D_sim, L = model.DVSA(V, W, lens)
            L.backward()
            out[sp] = (D_ind, D_sim, L.detach(), model.vis_ebd.fc1.weight.grad.clone())
            model.zero_grad()
        for a, b in zip(out["auto"], out[None]):
            assert torch.equal(a, b)
    finally:
        ops.SIM_PLANES_DEFAULT = old
        model.vis_ebd.sim_planes = model.word_ebd.sim_planes = "auto"
