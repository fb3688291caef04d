"""The similarity entry points nafae_sim_max_fwd_ws / _planes (live columns only; fp32-MFMA live-column kernel, planes kernel with a
bf16x3 / fp16 filter and exact fp32 finish, exact-fp32 fallback) against an fp64 evaluation of model.py:548-551,580-583,610-612 and against the
first-generation exact-fp32 kernel, over ragged / degenerate / adversarial shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(V, W, lens, Na, Nb, Ne):
    """fp64: masked S_ -> per-frame max / first arg-max, plus the top-2 gap."""
    Q = Na * Ne
    S = (V.double() @ W.double().t())
    masked = (torch.arange(Ne)[None, :] >= torch.tensor(lens)[:, None]).view(1, Q)
    S = S.masked_fill(masked, 0).view(-1, Nb, Q)
    m, i = S.max(1)
    if Nb > 1:
        t2 = S.topk(2, dim=1)[0]
        gap = t2[:, 0] - t2[:, 1]
    else:
        gap = torch.full_like(m, float("inf"))
    return m, i, gap, masked.expand(m.shape[0], Q)


PLANES = {"kind": False}


@pytest.fixture(autouse=True, params=[False, "bf16x3", "f16"], ids=["fp32-operands", "planes-bf16x3", "planes-f16"])
def plane_kind(request):
    """Every test of this file runs three times: on the fp32 operands alone (round 3's routes), and with the operand planes of
    round 4 (simplanes.hip: the many-live-column shapes then take sim_planes_kernel with a bf16x3 / a one-product fp16 filter; the
    other shapes ignore the planes) -- the contract (fp32 dot products, torch.max's tie / NaN rules) is the same."""
    PLANES["kind"] = request.param
    yield request.param
    PLANES["kind"] = False


def _run(V, W, lens_dev, Na, Ns, Nb, Ne, **kw):
    from nafae_amd import ops
    lt = torch.tensor(lens_dev, dtype=torch.int32, device="cuda")
    if "exact_fp32" not in kw:
        kw.setdefault("planes", PLANES["kind"])
    return ops.sim_max_fwd(V.cuda(), W.cuda(), lt, Na, Ns, Nb, Ne, **kw)


CASES = [
    # Na, Ns, Nb, Ne, D, lens
    (2, 2, 32, 8, 512, [3, 5]),                 # C1
    (3, 5, 20, 13, 64, [2, 0, 4]),              # reference defaults, a zero-length segment, Nb < 32
    (1, 4, 32, 8, 64, [3]),
    (4, 3, 7, 5, 32, [5, 1, 0, 2]),             # D = 32: a single 32-k chunk
    (4, 6, 40, 6, 128, [1, 6, 3, 2]),           # Nb = 40: second row block has 8 valid rows
    (8, 8, 128, 16, 512, None),                 # C2, histogram lengths (k-split over 4 waves)
    (2, 3, 300, 64, 512, [64, 10]),             # 74 live columns: two column groups of 64
    (2, 2, 33, 4, 512, [4, 4]),                 # one row beyond a block
    (1, 1, 1, 1, 32, [1]),                      # a single pair
    (2, 2, 64, 40, 1024, [40, 17]),             # D = 1024: 32-column groups only
    (2, 2, 50, 4, 96, [2, 4]),                  # 3 chunks (no k-split possible)
    (3, 2, 16, 4, 40, [1, 4, 2]),               # D % 32 != 0: falls back to the exact-fp32 kernel
    (2, 2, 32, 8, 512, [0, 0]),                 # nothing live
    (5, 1, 10, 3, 64, [3, 3, 3, 3, 3]),         # every slot live
    (70, 1, 40, 2, 128, [(i % 3 == 0) + (i % 35 == 0) for i in range(70)]),   # more than 64 segments, 26 live columns (fp32-MFMA kernel)
    (3, 2, 45, 12, 384, [12, 7, 12]),           # 31 live columns, D = 384: three 128-B lines per K quarter
    (2, 3, 31, 20, 256, [20, 12]),              # 32 live columns exactly
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "Na%d_Ns%d_Nb%d_Ne%d_D%d" % c[:5])
@pytest.mark.parametrize("with_hint", [True, False])
def test_sim_max_v2_matches_fp64(case, with_hint):
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D, lens = case
    lens = lens if lens is not None else syn.entity_lengths(Na, Ne, seed=1234)
    V = torch.tanh(syn.randn(21, "V%d" % D, (Na * Ns * Nb, D)))
    W = torch.tanh(syn.randn(21, "W%d" % D, (Na * Ne, D)))
    m, i, gap, masked = _ref(V, W, lens, Na, Nb, Ne)
    S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=lens if with_hint else None)
    S, Di = S.cpu().double(), Di.cpu()
    scale = max(float(m.abs().max()), 1e-6)
    assert (S[masked] == 0).all() and (Di[masked] == 0).all()
    assert float((S - m).abs().max()) < 2e-6 * scale             # S_max is an fp32 dot product of the winning pair
    bad = (Di != i) & ~masked & (gap > 1e-5 * scale)
    assert not bad.any(), "D_ind differs from the fp64 arg-max at %d decided entries" % int(bad.sum())
    # and the first-generation exact-fp32 kernel agrees with it wherever fp32 itself is decided
    S1, D1 = _run(V, W, lens, Na, Ns, Nb, Ne, exact_fp32=True)
    assert not ((D1.cpu() != Di) & ~masked & (gap > 1e-5 * scale)).any()
    assert float((S1.cpu().double() - S).abs().max()) < 2e-5 * scale


def _random_cases(n, seed):
    import random
    rs = random.Random(seed)
    out = []
    for _ in range(n):
        Na, Ns = rs.randint(1, 8), rs.randint(1, 8)
        Nb = rs.choice([1, 5, 31, 32, 33, 64, 100, 128, 129, 256, 300, 320])
        Ne = rs.choice([1, 3, 8, 16, 20, 32, 64])
        D = rs.choice([32, 64, 96, 128, 256, 512, 1024])
        kind = rs.random()
        if kind < 0.25:
            lens = [Ne] * Na                                     # every slot live (the dense path when Na * Ne > 64)
        elif kind < 0.35:
            lens = [0] * Na
        else:
            lens = [rs.randint(0, Ne) for _ in range(Na)]
        if Na * Ns * Nb * D > 6e6:
            Ns = max(1, int(6e6 / (Na * Nb * D)))
        out.append((Na, Ns, Nb, Ne, D, lens))
    return out


@pytest.mark.parametrize("case", _random_cases(48, 2024), ids=lambda c: "Na%d_Ns%d_Nb%d_Ne%d_D%d_L%d" % (c[:5] + (sum(c[5]),)))
def test_sim_max_v2_random_shapes(case):
    """Seeded random sweep over the shape / length space (both the live-column kernel and the dense tile path, ragged row
    blocks, empty and full segments): same checks as above."""
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D, lens = case
    V = torch.tanh(syn.randn(33, "Vr%d_%d" % (D, Nb), (Na * Ns * Nb, D)))
    W = torch.tanh(syn.randn(33, "Wr%d_%d" % (D, Ne), (Na * Ne, D)))
    m, i, gap, masked = _ref(V, W, lens, Na, Nb, Ne)
    for hint in (lens, None):
        S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=hint)
        S, Di = S.cpu().double(), Di.cpu()
        scale = max(float(m.abs().max()), 1e-6)
        assert (S[masked] == 0).all() and (Di[masked] == 0).all()
        assert float((S - m).abs().max()) < 2e-6 * scale
        assert not ((Di != i) & ~masked & (gap > 1e-5 * scale)).any()


def test_sim_max_v2_ties_pick_first_index():
    """Duplicate proposals (zero-padded rois give identical rows) -> the FIRST maximal index, like torch.max(dim)."""
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D = 2, 2, 70, 4, 512
    lens = [4, 2]
    V = torch.tanh(syn.randn(5, "Vt", (Na * Ns * Nb, D)))
    W = torch.tanh(syn.randn(5, "Wt", (Na * Ne, D)))
    V3 = V.view(Na * Ns, Nb, D)
    V3[:, 37] = V3[:, 5]            # same block / different blocks, higher index never wins
    V3[:, 69] = V3[:, 5]
    V3[0, :] = V3[0, 0]             # a frame of identical rows: every column's arg-max is 0
    m, i, gap, masked = _ref(V, W, lens, Na, Nb, Ne)
    S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=lens)
    Di = Di.cpu()
    assert (Di[0][~masked[0]] == 0).all()
    assert not (Di == 37).any() and not (Di == 69).any()
    decided_or_dup = (gap > 1e-5 * float(m.abs().max())) | (gap == 0)
    assert not ((Di != i) & ~masked & decided_or_dup).any()


def test_sim_max_v2_near_ties_are_decided_in_fp32():
    """Two proposals whose exact scores differ by ~1e-5 -- far inside the bf16x3 error but well above fp32 rounding: the
    refinement path (exact fp32 dot products of the listed candidates) must pick the fp64 winner in both index orders."""
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D = 2, 2, 96, 4, 512
    lens = [4, 3]
    V = torch.tanh(syn.randn(9, "Vn", (Na * Ns * Nb, D))) * 0.5
    W = torch.tanh(syn.randn(9, "Wn", (Na * Ne, D)))
    V3 = V.view(Na * Ns, Nb, D)
    Wd = W.double()
    # make row 11 of every frame the clear winner for query 0 (segment 0 slot 0), then plant near-copies of it
    V3[:, 11] = torch.tanh(W[0] * 1.5) * 0.1       # score ~28 (ulp 2e-6) against <10 for the random rows
    for f, (j, sign) in enumerate([(70, +1.0), (3, +1.0), (70, -1.0), (3, -1.0)]):
        r = V3[f, 11].clone()
        k = int(torch.argmax(W[0].abs()))
        r[k] = r[k] + sign * 2.0 ** -14 * torch.sign(W[0, k])      # exact score changes by +-2^-14 * |w_k| ~ 5e-5
        V3[f, j] = r
    m, i, gap, masked = _ref(V, W, lens, Na, Nb, Ne)
    assert float(gap[:, 0].max()) < 2e-4 and float(gap[:, 0].min()) > 1e-5       # planted: tiny but fp32-resolvable
    S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=lens)
    assert Di.cpu()[:, 0].tolist() == i[:, 0].tolist() == [70, 3, 11, 11]
    # refined entries carry the fp32 value of the winner
    assert float((S.cpu().double()[:, 0] - m[:, 0]).abs().max()) < 3e-6 * float(m.abs().max())


def test_sim_max_v2_frames_equals_whole_batch():
    """The frame-sharded call (a rank's F frames against all Q queries) returns the rows of the whole-batch call."""
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D = 4, 4, 48, 6, 512
    lens = [2, 6, 0, 3]
    V = torch.tanh(syn.randn(3, "Vf", (Na * Ns * Nb, D))).cuda()
    W = torch.tanh(syn.randn(3, "Wf", (Na * Ne, D))).cuda()
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    S, Di = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne, lens=lens, planes=PLANES["kind"])
    for lo, hi in ((0, 4), (4, 16), (15, 16)):
        Sl, Dl = ops.sim_max_fwd_frames(V[lo * Nb:hi * Nb].contiguous(), W, lt, Nb, Na, Ne, lens=lens, planes=PLANES["kind"])
        # (the launch plan -- how K is split over waves -- depends on the number of frames, so the fp32 sums may differ in
        # the last bit between the two calls; the indices may not)
        assert torch.allclose(Sl, S[lo:hi], rtol=0, atol=2e-6 * float(S.abs().max())) and torch.equal(Dl, Di[lo:hi])


def test_sim_max_v2_is_deterministic_and_graph_capturable():
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, Ne, D = 8, 8, 300, 64, 512
    lens = syn.entity_lengths(Na, Ne, seed=1234)
    V, W = syn.embeddings(Na * Ns * Nb, Na * Ne, D, seed=2)
    V, W = V.cuda(), W.cuda()
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    pk = PLANES["kind"]
    if pk:        # the planes as the embedding modules hand them over: attached to the tensors, produced outside the graph
        ops.attach_sim_planes(V, ops.sim_planes(V, pk))
        ops.attach_sim_planes(W, ops.sim_planes(W, pk))
    a = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne, lens=lens)
    b = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne)               # no hint: sized for all Q columns, another K split / route
    assert torch.allclose(a[0], b[0], rtol=0, atol=2e-6 * float(a[0].abs().max())) and torch.equal(a[1], b[1])
    a2 = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne, lens=lens)   # same plan: bit-identical run to run
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])
    b2 = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne)              # (with planes this is sim_planes_kernel: LDS list order varies,
    assert torch.equal(b[0], b2[0]) and torch.equal(b[1], b2[1])    # the decision must not)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne, lens=lens)    # allocate the stream's workspace outside the capture
        ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            c = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne, lens=lens)
            d = ops.sim_max_fwd(V, W, lt, Na, Ns, Nb, Ne)
        g.replay()
        st.synchronize()
    assert torch.equal(b[0], d[0]) and torch.equal(b[1], d[1])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])


@pytest.mark.parametrize("Ne,lens", [(8, [8, 5]), (40, [40, 33])], ids=["few", "dense"])
def test_sim_max_unsquashed_embeddings_with_planted_near_ties(Ne, lens):
    """The reference's DVSA.forward accepts ANY embeddings (model.py:548), not only tanh outputs: |v| up to 10 here.  The few-column
    kernel is exact fp32 throughout; the planes kernel builds its refinement margin from the operands' row statistics (max |x|, l2 norm)
    accordingly (round 2 assumed |V|, |W| <= 1).  Planted near-ties (exact gap ~1e-5 of the score scale) must come out like the
    fp64 arg-max in both index orders."""
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, D = 2, 3, 96, 512
    V = syn.randn(11, "Vu%d" % Ne, (Na * Ns * Nb, D)) * 3.3            # |v| up to ~10
    W = syn.randn(11, "Wu%d" % Ne, (Na * Ne, D)) * 2.0
    V3 = V.view(Na * Ns, Nb, D)
    V3[:, 11] = W[0] * 0.8                                              # a clear winner for query 0 ...
    for f in range(Na * Ns):
        j, sign = [(70, +1.0), (3, +1.0), (70, -1.0), (3, -1.0), (40, +1.0), (12, -1.0)][f]
        r = V3[f, 11].clone()
        k = int(torch.argmax(W[0].abs()))
        r[k] = r[k] + sign * 2.0 ** -9 * torch.sign(W[0, k])           # ... and a near-copy: the exact score moves by ~2^-9 * |w_k|
        V3[f, j] = r
    m, i, gap, masked = _ref(V, W, lens, Na, Nb, Ne)
    scale = float(m.abs().max())
    assert float(gap[:, 0].max()) < 1e-4 * scale and float(gap[:, 0].min()) > 2e-6 * scale
    for hint in (lens, None):
        S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=hint)
        S, Di = S.cpu().double(), Di.cpu()
        assert Di[:, 0].tolist() == i[:, 0].tolist() == [70, 3, 11, 11, 40, 11]
        assert float((S - m).abs().max()) < 2e-6 * scale
        assert not ((Di != i) & ~masked & (gap > 1e-5 * scale)).any()
        assert (S[masked] == 0).all() and (Di[masked] == 0).all()


@pytest.mark.parametrize("Ne,lens,hint", [(8, [8, 5], 9), (8, [8, 8], 4), (40, [40, 33], 40), (64, [64, 64], 64), (100, [100, 100], 70)],
                         ids=["few9of13", "few4of16", "live40of73", "live64of128", "frames70of200"])
def test_sim_max_too_small_live_hint_is_loud(Ne, lens, hint):
    """max_live_cols is an UPPER BOUND the host promises.  If it is too small the live columns beyond it cannot be computed:
    they must come back as NaN (like the loss kernel's NaN for the same mistake), never as silently wrong numbers; the columns
    inside the bound stay correct and the masked slots stay (0, 0)."""
    from nafae_amd import _lib, ops
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, D = 2, 2, 96, 512
    V = torch.tanh(syn.randn(13, "Vh", (Na * Ns * Nb, D))).cuda()
    W = torch.tanh(syn.randn(13, "Wh%d" % Ne, (Na * Ne, D))).cuda()
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    F, Q = Na * Ns, Na * Ne
    S = torch.empty(F, Q, device="cuda")
    Di = torch.empty(F, Q, device="cuda", dtype=torch.int64)
    L = _lib.lib()
    nws = int(L.nafae_sim_max_workspace_bytes(F, Nb, Na, Ne, D))
    ws = torch.zeros(max(nws, 16), device="cuda", dtype=torch.uint8)      # (zeroed once: the arrival counters of the one-launch merge)
    pk = PLANES["kind"]
    if pk:
        vp, wp = ops.sim_planes(V, pk), ops.sim_planes(W, pk)
        rc = L.nafae_sim_max_fwd_planes(ops._p(V), ops._p(W), ops._p(lt), F, Nb, Na, Ne, D, hint, ops.SIM_PLANES_KINDS[pk],
                                        ops._p(vp.planes), ops._p(vp.stats), ops._p(wp.planes), ops._p(wp.stats), ops._p(S), ops._p(Di),
                                        ops._p(ws), ws.numel(), ops._stream())
    else:
        rc = L.nafae_sim_max_fwd_ws(ops._p(V), ops._p(W), ops._p(lt), F, Nb, Na, Ne, D, hint, ops._p(S), ops._p(Di), ops._p(ws),
                                    ws.numel(), ops._stream())
    assert rc == 0
    m, i, gap, masked = _ref(V.cpu(), W.cpu(), lens, Na, Nb, Ne)
    S, Di = S.cpu().double(), Di.cpu()
    live_index = torch.full((Q,), -1, dtype=torch.long)
    n = 0
    for a, l in enumerate(lens):
        for e in range(l):
            live_index[a * Ne + e] = n
            n += 1
    assert n > hint
    # (the fp32 live-column route, hint <= 64, computes exactly `hint` columns -- frames of 96 rows stay on it whatever the planes --
    # the planes kernel whole groups of 64)
    computed = (live_index >= 0) & (live_index < (hint if hint <= 64 else ((hint + 63) // 64) * 64))
    lost = (live_index >= 0) & ~computed
    assert lost.any()
    assert torch.isnan(S[:, lost]).all() and (Di[:, lost] == 0).all()
    scale = float(m.abs().max())
    assert float((S[:, computed] - m[:, computed]).abs().max()) < 2e-6 * scale
    assert (S[masked] == 0).all() and (Di[masked] == 0).all()


@pytest.mark.parametrize("Ne,lens", [(8, [8, 5]), (40, [40, 33])], ids=["few", "dense"])
def test_sim_max_nan_and_inf_inputs_follow_torch_max(Ne, lens):
    """Diverged training puts NaN / Inf into the embeddings.  torch.max propagates a NaN with the index of the first NaN row
    (model.py:610-612 -> D_ind feeds the box gathers of postprocess / record_det): the kernels must never emit an out-of-range
    index, and a single NaN row of V must reach S_max instead of being skipped."""
    from nafae_amd import synthetic as syn
    Na, Ns, Nb, D = 2, 2, 96, 512
    V = torch.tanh(syn.randn(17, "Vq", (Na * Ns * Nb, D)))
    W = torch.tanh(syn.randn(17, "Wq%d" % Ne, (Na * Ne, D)))
    V3 = V.view(Na * Ns, Nb, D)
    V3[0, 50, 7] = float("nan")          # one NaN element in one proposal of frame 0
    V3[1, 20, 3] = float("inf")          # an Inf: the score of that row is +-Inf (or NaN where 0 * Inf meets)
    V3[1, 60, 3] = float("inf")
    W[1, 100] = float("nan")             # a NaN query row: every frame's column 1 is NaN, first index 0
    Q = Na * Ne
    S_ = V @ W.t()
    masked = (torch.arange(Ne)[None, :] >= torch.tensor(lens)[:, None]).view(1, Q)
    S_ = S_.masked_fill(masked, 0).view(-1, Nb, Q)
    m, i = S_.max(1)                     # torch's own semantics are the oracle here
    S, Di = _run(V, W, lens, Na, Ns, Nb, Ne, lens=lens)
    S, Di = S.cpu(), Di.cpu()
    assert int(Di.min()) >= 0 and int(Di.max()) < Nb
    live = ~masked.expand_as(m)
    assert torch.equal(torch.isnan(S) & live, torch.isnan(m) & live)
    nanpos = torch.isnan(m) & live
    assert nanpos[0][live[0]].all() and nanpos[:, 1].all()          # frame 0 (NaN row) and column 1 (NaN query) entirely
    assert torch.equal(Di[nanpos], i[nanpos])              # the index of the FIRST NaN row
    fin = live & torch.isfinite(m)
    assert torch.allclose(S[fin], m[fin], rtol=0, atol=3e-6 * float(m[fin].abs().max()))
    inf = live & torch.isinf(m)
    assert inf.any() and torch.equal(S[inf], m[inf]) and torch.equal(Di[inf], i[inf])
