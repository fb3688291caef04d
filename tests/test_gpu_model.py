"""GPU parity tests, model level: the host-side mirror of the reference's model.py classes (nafae_amd/model.py,
detector.py) against golden vectors generated from the imported reference and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    return cfg


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def test_embed_and_dvsa_modules_match_reference(gpu):
    """VisEbd / WordEbd / DVSA modules chained exactly like model.py:711,749,768-772, against the reference's
    outputs and parameter gradients (tests/golden/embed.npz)."""
    from nafae_amd.model import DVSA, VisEbd, WordEbd, default_args
    g = np.load(os.path.join(G, "embed.npz"))
    Na, Ns, Nb, Ne, D, FC, Gd = [int(x) for x in g["shape"]]
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0,
                        vis_lam=4.13, word_ebd_dim=D, vis_fc_dim=FC, glove_dim=Gd)
    ve, we, dv = VisEbd(args).cuda(), WordEbd(args).cuda(), DVSA(args, gpu).cuda()
    with torch.no_grad():
        for p, k in ((ve.fc1.weight, "ve_w"), (ve.fc1.bias, "ve_b"), (we.fc1.weight, "we_w"), (we.fc1.bias, "we_b"),
                     (we.bn.weight, "bn_w"), (we.bn.bias, "bn_b")):
            p.copy_(torch.from_numpy(g[k]))
    fc7, glove = torch.from_numpy(g["fc7"]).cuda(), torch.from_numpy(g["glove"]).cuda()
    lens = g["lens"].tolist()
    ve.train(); we.train(); dv.init_train()
    V, W = ve(fc7), we(glove)
    assert relerr(V.detach().cpu(), g["V_train"]) < TOL and relerr(W.detach().cpu(), g["W_train"]) < TOL
    Di, Ds, L = dv(V, W, lens)
    loss = torch.nn.L1Loss()(L, torch.zeros_like(L))
    loss.backward()
    assert np.array_equal(Di.cpu().numpy(), g["D_ind_train"])
    assert abs(float(L) - float(g["loss_train"])) < TOL * abs(float(g["loss_train"]))
    for p, k in ((ve.fc1.weight, "g_ve_w"), (ve.fc1.bias, "g_ve_b"), (we.fc1.weight, "g_we_w"), (we.fc1.bias, "g_we_b"),
                 (we.bn.weight, "g_bn_w"), (we.bn.bias, "g_bn_b")):
        # (the bias in front of a train-mode BatchNorm has an exactly-zero true gradient: absolute floor)
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - g[k]).max()
        assert err < TOL * max(np.abs(g[k]).max(), 1e-2), (k, err)
    assert relerr(we.bn.running_mean.cpu(), g["run_mean"]) < TOL and relerr(we.bn.running_var.cpu(), g["run_var"]) < TOL
    assert all(p.grad is None for p in dv.parameters())          # DVSA's own parameters never get a gradient
    ve.eval(); we.eval(); dv.init_eval()
    with torch.no_grad():
        V, W = ve(fc7), we(glove)
        Di, Ds, L = dv(V, W, lens)
    assert relerr(W.cpu(), g["W_eval"]) < TOL
    assert np.array_equal(Di.cpu().numpy(), g["D_ind_eval"])
    assert relerr(Ds.cpu(), g["D_sim_eval"]) < TOL
    assert abs(float(L) - float(g["loss_eval"])) < TOL * abs(float(g["loss_eval"]))


PRECISIONS = ["bf16x3", "f32"]      # both must meet the 1e-4 fp32 bar (plain bf16: tests/test_gpu_bf16.py)


def _detector(seed, precision, device="cuda"):
    from nafae_amd import synthetic as syn
    from nafae_amd.detector import vgg16
    fr = vgg16(np.array([''] * 2501), pretrained=False, class_agnostic=False)
    fr.create_architecture()
    fr.load_state_dict(syn.detector_state(seed=seed, heads=False), strict=False)
    fr.precision = precision
    return fr.eval().to(device)


def _base_nhwc_f32(fr, im):
    from nafae_amd import ops
    b = fr.base_features(im)
    return ops.merge_bf16(b) if isinstance(b, ops.Planes) else b


@pytest.mark.parametrize("precision", PRECISIONS)
def test_detector_matches_reference_golden(gpu, precision):
    """fasterRCNN.forward against the reference's own forward on 2 small frames (tests/golden/detector.npz)."""
    from nafae_amd import synthetic as syn
    from oracle import detector as OD
    g = np.load(os.path.join(G, "detector.npz"))
    gpu.TEST.RPN_POST_NMS_TOP_N = int(g["post_nms_topN"])
    fr = _detector(int(g["seed"]), precision)
    h, w = [int(x) for x in g["frames_hw"]]
    im, im_info = syn.frames(2, h, w, seed=int(g["seed"]))
    base = _base_nhwc_f32(fr, im.cuda())
    assert relerr(base.permute(0, 3, 1, 2).cpu(), g["base_feat"]) < TOL
    rois, roi_scores, pooled, fc7 = fr(im.cuda(), im_info.cuda(), None, None)
    assert tuple(pooled.shape) == (16, 512, 7, 7) and tuple(fc7.shape) == (16, 4096)
    # box coordinates are fp32 values downstream of 14 conv layers: "same proposal" = within 0.02 px (1e-4 of 224)
    same = (np.abs(rois.cpu().numpy() - g["rois"]) < 0.02).all(-1).reshape(-1)
    assert same.sum() >= same.size - 1, "rois differ from the reference: %s" % same     # measured: all identical; one near-tie flip tolerated
    assert np.allclose(roi_scores.cpu().numpy().reshape(-1)[same], g["roi_scores"].reshape(-1)[same], rtol=1e-5)
    assert relerr(pooled.cpu().numpy()[same][:, ::37], g["pooled_sub"][same]) < TOL
    assert relerr(fc7.cpu().numpy()[same], g["fc7"][same]) < TOL
    # stage-wise with teacher forcing: oracle RPN/proposals fed with the HIP base_feat must give the HIP rois
    sd = syn.detector_state(seed=int(g["seed"]), heads=False)
    rp = {k[len('RCNN_rpn.'):]: v for k, v in sd.items() if k.startswith('RCNN_rpn.')}
    prob, deltas = OD.rpn_head(base.permute(0, 3, 1, 2).cpu().contiguous(), rp)
    s, props = OD.decode_proposals(prob, deltas, im_info, 16, [4, 8, 16, 32], [0.5, 1, 2])
    r_o, rs_o, _ = OD.select_proposals(s, props, OD.sort_desc(s), 6000, int(g["post_nms_topN"]), 0.7)
    assert ((rois.cpu() - r_o).abs().max(-1)[0] < 1e-3).all()      # identical base_feat in, identical proposals out
    # ROI-Align + head fed with the HIP rois
    pooled_o = OD.roi_align_avg(base.permute(0, 3, 1, 2).cpu().contiguous(), rois.cpu().view(-1, 5))
    assert relerr(pooled.cpu(), pooled_o) < 2e-5
    assert relerr(fc7.cpu(), OD.head_to_tail(pooled_o, sd)) < TOL
    # _head_to_tail on the logical NCHW view gives the same fc7 (vgg16_rpn.py:56-61 signature)
    assert relerr(fr._head_to_tail(pooled).cpu(), fc7.cpu()) < 1e-5


@pytest.mark.parametrize("precision", PRECISIONS)
def test_detector_config_c1_against_oracle(gpu, precision):
    """BASELINE config C1: 4 frames 224x224, 32 proposals/frame; whole detector vs the CPU oracle."""
    from nafae_amd import synthetic as syn
    from oracle import detector as OD
    gpu.TEST.RPN_POST_NMS_TOP_N = 32
    fr = _detector(1234, precision)
    im, im_info = syn.frames(4, 224, 224, seed=1234)
    rois, roi_scores, pooled, fc7 = fr(im.cuda(), im_info.cuda(), None, None)
    sd = syn.detector_state(seed=1234, heads=False)
    ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=[4, 8, 16, 32], ANCHOR_RATIOS=[0.5, 1, 2], RPN_PRE_NMS_TOP_N=6000,
                RPN_POST_NMS_TOP_N=32, RPN_NMS_THRESH=0.7, POOLING_SIZE=7)
    r_o, s_o, pooled_o, fc7_o = OD.detector_forward(im, im_info, sd, ocfg)
    same = ((rois.cpu() - r_o).abs() < 0.02).all(-1).view(-1).numpy()
    assert same.sum() >= same.size - 1, same.mean()        # measured: all identical; one near-tie flip tolerated
    assert relerr(fc7.cpu().numpy()[same], fc7_o.numpy()[same]) < TOL
    assert (fr.n_keep.cpu() <= 32).all()


@pytest.mark.parametrize("F,H,W", [(1, 64, 96), (3, 96, 80), (2, 112, 160), (5, 144, 48), (2, 48, 208), (7, 80, 80)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_detector_other_frame_sizes_against_oracle(gpu, precision, F, H, W):
    """The whole detector at frame sizes other than 224 x 224 (feature maps that are not multiples of the conv kernels' tile
    shapes, frame counts that leave ragged tiles, layers that change schedule: patch / raster-run / stream-K, fused or separate
    pool) against the CPU oracle."""
    from nafae_amd import synthetic as syn
    from oracle import detector as OD
    torch.set_num_threads(16)
    gpu.TEST.RPN_POST_NMS_TOP_N = 16
    fr = _detector(77, precision)
    im, im_info = syn.frames(F, H, W, seed=100 + H + W)
    rois, roi_scores, pooled, fc7 = fr(im.cuda(), im_info.cuda(), None, None)
    sd = syn.detector_state(seed=77, heads=False)
    ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=[4, 8, 16, 32], ANCHOR_RATIOS=[0.5, 1, 2], RPN_PRE_NMS_TOP_N=6000,
                RPN_POST_NMS_TOP_N=16, RPN_NMS_THRESH=0.7, POOLING_SIZE=7)
    r_o, s_o, pooled_o, fc7_o = OD.detector_forward(im, im_info, sd, ocfg)
    same = ((rois.cpu() - r_o).abs() < 0.02).all(-1).view(-1).numpy()
    assert same.sum() >= same.size - 1, same.mean()        # measured: all identical; one near-tie flip tolerated
    assert relerr(fc7.cpu().numpy()[same], fc7_o.numpy()[same]) < TOL


@pytest.mark.parametrize("F,H,W", [(1, 64, 96), (3, 96, 80), (2, 112, 160), (9, 224, 224)])
def test_conv_stack_plain_bf16_other_sizes(gpu, F, H, W):
    """BASELINE C3's arithmetic (plain bf16 operands, fp32 accumulation) through the conv stack at small frame counts and sizes:
    few tiles send the 64-channel layers to the raster-run kernel's 256 x 64 tiles instead of the patch kernel (the configuration
    that had the staging-order bug).  Against the oracle's fp32 conv stack, at the bf16 tolerance."""
    from nafae_amd import synthetic as syn
    from oracle import detector as OD
    torch.set_num_threads(16)
    fr = _detector(78, "bf16")
    im, _ = syn.frames(F, H, W, seed=300 + H + W)
    got = _base_nhwc_f32(fr, im.cuda()).cpu().permute(0, 3, 1, 2)
    want = OD.vgg16_features(im, syn.detector_state(seed=78, heads=False))
    assert relerr(got, want) < 3e-2


def test_full_train_step_and_eval_step(gpu):
    """One iteration of the reference's train loop body (model.py:706-774) and of validate (model.py:875-947)."""
    from nafae_amd.model import default_args, postprocess, stepRCNN
    from nafae_amd.train import eval_step, make_batch, setup_training, train_step
    Na, Ns, Ne, Nb = 2, 3, 8, 16
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne, Delta=10.0, vis_lam=4.13)
    model, opt, crit, _ = setup_training(args, seed=3)
    batch = make_batch(Na, Ns, Ne, H=96, W=80, seed=3, lens=[3, 5])
    w0 = model.vis_ebd.fc1.weight.detach().clone()
    base0 = model.fasterRCNN.RCNN_top[0].weight.detach().clone()
    losses = [float(train_step(model, opt, crit, batch, args)[0]) for _ in range(3)]
    assert all(np.isfinite(losses))
    assert not torch.equal(w0, model.vis_ebd.fc1.weight)                     # trainable params moved
    assert torch.equal(base0, model.fasterRCNN.RCNN_top[0].weight)           # detector stayed frozen
    assert int(model.word_ebd.bn.num_batches_tracked) == 3
    model.eval(); model.DVSA.init_eval()
    L, D, D_sim, rois = eval_step(model, batch)
    assert D.dtype == torch.int64 and tuple(D.shape) == (Na * Ns, Na * Ne) and int(D.max()) < Nb
    Dp, Sp = postprocess(D.cpu().numpy(), D_sim.cpu().numpy(), Na, Ns, Nb, Ne)
    assert Dp.shape == (Na, Ns, Ne) and Dp.max() < Na * Ns * Nb
    # long-segment chunking (model.py:429-454)
    r1, f1, c1 = stepRCNN(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes, model, step_size=4)
    r2, _, f2, c2 = model.fasterRCNN(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
    # The fp32 conv engine's stream-K split points depend on the number of frames in the launch, so a 4-frame chunk and the 6-frame batch
    # sum the same products in a different order: boxes agree to fp32 rounding of the conv stack (1e-4 relative, the parity bar), the
    # proposal COUNT per frame is exact.
    assert torch.allclose(r1[:, :, 1:], r2[:, :, 1:], rtol=1e-4, atol=2e-3)
    assert torch.allclose(f1, f2, rtol=1e-3, atol=1e-3 * float(f2.abs().max()))
    assert torch.allclose(c1, c2, rtol=1e-3, atol=1e-3 * float(c2.abs().max()))


def test_dropout_is_active_in_train_mode_only(gpu):
    from nafae_amd.model import VisEbd, default_args
    args = default_args(vis_fc_dim=64, word_ebd_dim=32, dropout_rate=0.5)
    ve = VisEbd(args).cuda()
    x = torch.randn(256, 64, device="cuda") * 100
    ve.train()
    y = ve(x)
    frac0 = float((y == 0).float().mean())
    assert 0.4 < frac0 < 0.6
    ve.eval()
    assert float((ve(x) == 0).float().mean()) < 0.01


def test_fused_clip_adam_matches_torch(gpu):
    """nafae_adam_step == clip_grad_norm_ + torch.optim.Adam (model.py:773-774, :1077-1082), incl. the clipping branch."""
    from nafae_amd.parallel import FusedClipAdam, GradAllReducer
    torch.manual_seed(0)
    mods = [torch.nn.Linear(40, 24), torch.nn.Linear(12, 24), torch.nn.BatchNorm1d(24)]
    ref = [torch.nn.Linear(40, 24), torch.nn.Linear(12, 24), torch.nn.BatchNorm1d(24)]
    for a, b in zip(mods, ref):
        b.load_state_dict(a.state_dict())
    mods = [m.cuda() for m in mods]
    ref = [m.cuda() for m in ref]
    ps = [p for m in mods for p in m.parameters()]
    rps = [p for m in ref for p in m.parameters()]
    red = GradAllReducer(ps)
    opt = FusedClipAdam(red, lr=1e-3, weight_decay=1e-5, max_norm=5.0)
    topt = torch.optim.Adam(rps, lr=1e-3, weight_decay=1e-5)
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(4):
        scale = 10.0 if it % 2 == 0 else 0.01          # alternate clipped / unclipped steps
        red.zero_grad()
        for p, rp in zip(ps, rps):
            gr = torch.randn(p.shape, device="cuda", generator=g) * scale
            p.grad.copy_(gr)
            rp.grad = gr.clone()
        tn = torch.nn.utils.clip_grad_norm_(rps, 5.0)
        topt.step()
        opt.step()
        assert abs(float(opt.total_norm) - float(tn)) < 1e-4 * float(tn)
        for p, rp in zip(ps, rps):
            assert relerr(p.detach().cpu(), rp.detach().cpu()) < 1e-5
            assert relerr(p.grad.cpu(), rp.grad.cpu()) < 1e-5


def test_validate_path_end_to_end(gpu):
    """validate() body (model.py:869-947) on a synthetic video: stepRCNN -> embeddings -> DVSA(eval) -> postprocess ->
    record_det -> evaluate_box, with gt boxes planted on the grounded boxes so the expected accuracy is known."""
    from nafae_amd import evaluate as E
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training, validate_segment
    Na, Ns, Ne, Nb = 2, 3, 4, 8
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne)
    model, _, _, _ = setup_training(args, seed=5)
    model.eval(); model.DVSA.init_eval()
    batch = make_batch(Na, Ns, Ne, H=96, W=80, seed=5, lens=[2, 3])
    vid_entities = [['bowl', 'egg'], ['pan', 'oil', 'salt']]
    img_ids = list(range(Na * Ns))
    dets = [[], [], [], []]
    loss = validate_segment(model, batch, vid_entities, img_ids, args, dets)
    assert np.isfinite(loss) and len(dets[0]) == Ns * (2 + 3)
    classes = ['bowl', 'egg', 'pan', 'oil', 'salt']
    # gt = exactly the grounded boxes -> every query and every box is matched
    recs = [{'label': [], 'bbox': [], 'thr': [], 'img_ids': []} for _ in img_ids]
    for i, l, b in zip(dets[0], dets[1], dets[2]):
        recs[i]['label'].append(l); recs[i]['bbox'].append(b); recs[i]['thr'].append(0.5); recs[i]['img_ids'].append(i)
    assert abs(E.evaluate_box(recs, dets, classes) - 1.0) < 1e-5
    # gt far away from everything -> zero
    for r in recs:
        r['bbox'] = [np.array([1000., 1000., 1010., 1010.]) for _ in r['bbox']]
    assert E.evaluate_box(recs, dets, classes) == 0.0


def test_pipelined_trainer_equals_sequential(gpu):
    """detector(k+1) overlapped with tail(k) on a second stream gives the same training trajectory as the sequential
    loop (the detector is frozen, so nothing it computes depends on the previous optimiser step)."""
    from nafae_amd.model import default_args
    from nafae_amd.train import PipelinedTrainer, make_batch, setup_training, train_step
    Na, Ns, Ne, Nb = 2, 4, 8, 16
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, Delta=10.0, vis_lam=4.13, dropout_rate=0.0)
    batches = [make_batch(Na, Ns, Ne, H=96, W=80, seed=10 + i, lens=[3, 5]) for i in range(4)]
    m1, o1, c1, r1 = setup_training(args, seed=3)
    seq = [float(train_step(m1, o1, c1, b, args, r1)[0]) for b in batches]
    m2, o2, c2, r2 = setup_training(args, seed=3)
    pipe = PipelinedTrainer(m2, o2, c2, args, r2)
    pipe.submit(batches[0])
    par = [float(pipe.step(batches[i + 1] if i + 1 < len(batches) else None)[0]) for i in range(len(batches))]
    torch.cuda.synchronize()
    assert np.allclose(seq, par, rtol=1e-5), (seq, par)
    assert len(set(seq)) > 1                                        # the trajectory actually moves
    assert seq == par and torch.equal(o1.flat_params, o2.flat_params)          # bit for bit: no kernel of a step depends on timing


def test_reference_default_shapes(gpu):
    """The reference's own default configuration (model.py:109-111,181-183,214-216; cfgs/vgg16.yml:14): Na=8 segments x
    Ns=5 frames of 224x224, Nb=20 proposals per frame, Ne=13 query slots -- R=800, Q=104, none of them tile multiples."""
    from nafae_amd.config import cfg_from_file, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import eval_step, make_batch, setup_training, train_step
    from oracle import dvsa as O
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    args = default_args(Delta=10.0, vis_lam=4.13, dropout_rate=0.0)
    Na, Ns, Ne, Nb = args.batch_size, args.sample_num, args.max_ent_len, gpu.TEST.RPN_POST_NMS_TOP_N
    assert (Na, Ns, Ne, Nb) == (8, 5, 13, 20)
    model, opt, crit, red = setup_training(args, seed=11)
    batch = make_batch(Na, Ns, Ne, seed=11)
    with torch.no_grad():
        rois, _, pooled, fc7 = model.fasterRCNN(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
    assert tuple(rois.shape) == (40, 20, 5) and tuple(fc7.shape) == (800, 4096) and tuple(pooled.shape) == (800, 512, 7, 7)
    V, W = model.vis_ebd(fc7), model.word_ebd(batch.glove_feats)
    D, Ds, L = model.DVSA(V, W, batch.entities_length)
    Do, Dso, Lo = O.dvsa_forward(V.detach().cpu(), W.detach().cpu(), batch.entities_length, Na, Nb, Ne, 10.0, 4.13, "train")
    assert torch.equal(D.cpu(), Do) and relerr(Ds.detach().cpu(), Dso) < TOL and abs(float(L) - float(Lo)) < TOL * abs(float(Lo))
    l0 = float(train_step(model, opt, crit, batch, args, red)[0])
    l1 = float(train_step(model, opt, crit, batch, args, red)[0])
    assert np.isfinite([l0, l1]).all() and l1 != l0
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))


def test_stepRCNN_streamed_host_input(gpu):
    """stepRCNN (model.py:429-454) fed from HOST memory -- float32 NCHW and raw uint8 HWC frames -- through the
    double-buffered copy stream equals the device-resident call chunk for chunk (150 frames = 64 + 64 + 22)."""
    from nafae_amd.model import default_args, stepRCNN
    from nafae_amd.train import build_model
    gpu.TEST.RPN_POST_NMS_TOP_N = 16
    model = build_model(default_args(), seed=5).eval()
    Ns, H, W = 150, 64, 96
    g = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 255, (Ns, H, W, 3), dtype=torch.uint8, generator=g)          # decoded BGR frames
    f32 = (u8.float() - 127.5).permute(0, 3, 1, 2).contiguous()                        # youcook2.py:212-214 on the host
    info = torch.tensor([[H, W, 1.0]]).repeat(Ns, 1)
    gt, nb = torch.zeros(1, 1, 5, device="cuda"), torch.zeros(1, device="cuda")
    ref = stepRCNN(f32.cuda(), info.cuda(), gt, nb, model)
    assert tuple(ref[0].shape) == (Ns, 16, 5) and tuple(ref[1].shape) == (Ns * 16, 512, 7, 7) and tuple(ref[2].shape) == (Ns * 16, 4096)
    for host in (f32, u8):
        out = stepRCNN(host, info, gt, nb, model)
        assert all(torch.equal(a, b) for a, b in zip(out, ref))
    out = stepRCNN(u8, info, gt, nb, model, need_roi_feats=False)
    assert out[1] is None and torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])
    assert model.fasterRCNN.materialize_pooled is True
    with pytest.raises(TypeError):
        stepRCNN(f32.double(), info, gt, nb, model)


def test_prepare_batch_raw_frames_equals_host_preprocessing(gpu):
    """prepare_batch (model.py:684-747): decoded uint8 frames handed to the GPU as they are (the -127.5 and the HWC reading happen
    inside the first conv layer) == the reference's host-side float path: same values, and the two Batches train identically."""
    import argparse
    from nafae_amd.model import default_args
    from nafae_amd.train import combine_batches_synthetic, prepare_batch, setup_training, train_step
    args = default_args(batch_size=2, sample_num=2, max_ent_len=8, dropout_rate=0.0)
    gpu.TEST.RPN_POST_NMS_TOP_N = 32
    lb = list(combine_batches_synthetic(2, 2, 8, seed=5))
    vocab = sorted(set(lb[1]))
    glove = argparse.Namespace(stoi={w: i for i, w in enumerate(vocab)}, vectors=torch.randn(len(vocab), 200) * 0.4)
    bf = prepare_batch(tuple(lb), glove, args)
    u8 = (lb[0] + 127.5).astype(np.uint8)
    br = prepare_batch((u8,) + tuple(lb[1:]), glove, args, raw_frames=True)
    assert br.im_data.dtype == torch.uint8 and tuple(br.im_data.shape) == (4, 224, 224, 3)
    assert torch.equal(bf.im_data, (br.im_data.float() - 127.5).permute(0, 3, 1, 2)) and torch.equal(bf.glove_feats, br.glove_feats)
    losses = []
    for b in (br, bf):
        model, opt, crit, red = setup_training(args, seed=3)
        losses.append(float(train_step(model, opt, crit, b, args, red)[0]))
    assert np.isfinite(losses[0]) and losses[0] == losses[1]


def test_train_epoch_pipelined_equals_sequential(gpu):
    """train_epoch (the body of model.py:676-795) over a list of loader tuples, one of them without entities (skipped like
    the reference): pipelined and sequential runs take the same steps and end with identical parameters."""
    import argparse
    from nafae_amd.model import default_args
    from nafae_amd.train import combine_batches_synthetic, setup_training, train_epoch
    args = default_args(batch_size=2, sample_num=2, max_ent_len=8, dropout_rate=0.0)
    gpu.TEST.RPN_POST_NMS_TOP_N = 32
    loader = [list(combine_batches_synthetic(2, 2, 8, seed=40 + i)) for i in range(4)]
    loader[2][1], loader[2][2] = [], [0, 0]                                        # nothing to ground in this batch
    vocab = sorted({w for lb in loader for w in lb[1]})
    glove = argparse.Namespace(stoi={w: i for i, w in enumerate(vocab)},
                               vectors=torch.randn(len(vocab), 200, generator=torch.Generator().manual_seed(1)) * 0.4)
    res = []
    for pipelined in (False, True):
        model, opt, crit, red = setup_training(args, seed=9)
        seen = []
        mean_loss, n = train_epoch([tuple(lb) for lb in loader], model, glove, crit, opt, red, args, pipelined=pipelined,
                                   on_step=lambda i, loss, D, D_sim, rois, b: seen.append((i, float(loss))))
        assert n == 3 and [i for i, _ in seen] == [0, 1, 2]
        assert abs(mean_loss - np.mean([l for _, l in seen])) < 1e-4 * abs(mean_loss)
        res.append((mean_loss, opt.flat_params.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])       # every kernel of a step has a fixed summation order


def test_validate_epoch_over_loader_tuples(gpu, tmp_path):
    """validate_epoch = validate() (model.py:800-991) over loader tuples: skip rule, 800-frame cap, detection pickle in the
    reference's format, accuracy 1.0 against ground truth planted on the grounded boxes."""
    import argparse, pickle
    from nafae_amd.model import default_args
    from nafae_amd.train import combine_batches_synthetic, setup_training, validate_epoch
    Na, Ns, Ne, Nb = 1, 5, 4, 8
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=2, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne)
    model, _, _, _ = setup_training(args, seed=5)
    vids = []
    for v in range(3):
        lb = list(combine_batches_synthetic(Na, Ns, Ne, H=64, W=64, seed=60 + v))
        lb[2] = [2]; lb[1] = ['bowl', 'egg']; lb[7] = list(range(Ns * v, Ns * v + Ns))
        vids.append(lb)
    vids[1][1], vids[1][2] = [], [0]                                               # a video without entities: skipped
    glove = argparse.Namespace(stoi={'bowl': 0, 'egg': 1}, vectors=torch.randn(2, 200, generator=torch.Generator().manual_seed(2)) * 0.4)
    path = str(tmp_path / "ground_res.pkl")
    acc, loss, dets = validate_epoch([tuple(v) for v in vids], model, glove, args, result_path=path, max_frames=3)
    assert acc is None and np.isfinite(loss)
    assert len(dets[0]) == 2 * 3 * 2 and sorted(set(dets[0])) == [0, 1, 2, 10, 11, 12]      # 2 videos x 3 capped frames x 2 entities
    assert pickle.load(open(path, 'rb'))[1] == dets[1]
    classes = ['bowl', 'egg']
    recs = [{'label': [], 'bbox': [], 'thr': [], 'img_ids': []} for _ in range(3 * Ns)]      # indexed by image id
    for i, l, b in zip(dets[0], dets[1], dets[2]):
        recs[i]['label'].append(l); recs[i]['bbox'].append(b); recs[i]['thr'].append(0.5); recs[i]['img_ids'].append(i)
    acc2, _, dets2 = validate_epoch([tuple(v) for v in vids], model, glove, args, recs=recs, class_list=classes, max_frames=3)
    assert dets2[1] == dets[1]
    assert abs(acc2 - 1.0) < 1e-5                                                  # gt planted on the grounded boxes


def test_visebd_uses_planes_only_while_fc7_is_untouched(gpu):
    """VisEbd continues on the detector's split-bf16 planes of fc7 (same values to ~1e-5); an in-place edit of fc7 or a
    derived tensor falls back to the exact fp32 GEMM on the tensor's actual contents."""
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training
    gpu.TEST.RPN_POST_NMS_TOP_N = 32
    args = default_args(batch_size=2, sample_num=2, max_ent_len=8, dropout_rate=0.0)
    model, _, _, _ = setup_training(args, seed=4)
    model.fasterRCNN.precision = 'bf16x3'                          # (whatever NAFAE_PRECISION says)
    batch = make_batch(2, 2, 8, seed=4)
    with torch.no_grad():
        _, _, _, fc7 = model.fasterRCNN(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
        assert getattr(fc7, "_nafae_planes", None) is not None
        v_planes = model.vis_ebd(fc7)
        v_exact = model.vis_ebd(fc7.clone())                       # a copy carries no planes: fp32 GEMM
        assert relerr(v_planes.cpu(), v_exact.cpu()) < 5e-5 and not torch.equal(v_planes, v_exact)
        fc7.mul_(0.5)                                              # in-place edit: the planes are stale and must be ignored
        assert torch.equal(model.vis_ebd(fc7), model.vis_ebd(fc7.clone()))
