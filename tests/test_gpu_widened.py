"""The rows of SURVEY.md section 8(f) on the GPU, each against the reference-generated fixtures / the CPU oracle:
checkpoint I/O (8f.3: model.py:1039-1064, :1115-1126) and the streamed detector (8f.4: model.py:429-454, :851-854)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


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


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_detector_checkpoint_file_to_reference_outputs(gpu, tmp_path, precision):
    """A faster_rcnn_gnome.pth-shaped file ({'model': state_dict incl. the unused 2501-class heads, 'pooling_mode'}) ->
    checkpoint.load_detector_checkpoint -> lazy re-pack into kernel layout -> detector forward == the REFERENCE's own forward
    on the same weights (tests/golden/detector.npz was produced by the imported reference with these seed-77 weights)."""
    from nafae_amd import synthetic as syn
    from nafae_amd.checkpoint import load_detector_checkpoint
    from nafae_amd.model import GroundModel, default_args
    g = np.load(os.path.join(G, "detector.npz"))
    gpu.TEST.RPN_POST_NMS_TOP_N = int(g["post_nms_topN"])
    path = str(tmp_path / "faster_rcnn_gnome.pth")
    torch.save({'model': syn.detector_state(seed=int(g["seed"]), heads=True), 'pooling_mode': 'align', 'epoch': 7}, path)
    model = GroundModel(default_args(), gpu).cuda()
    model.fasterRCNN.precision = precision
    h, w = [int(x) for x in g["frames_hw"]]
    im, im_info = syn.frames(2, h, w, seed=int(g["seed"]))
    before = model.fasterRCNN(im.cuda(), im_info.cuda(), None, None)[3].clone()      # random-init weights, packs once
    load_detector_checkpoint(model, path)
    assert gpu.POOLING_MODE == 'align'
    rois, roi_scores, pooled, fc7 = model.fasterRCNN(im.cuda(), im_info.cuda(), None, None)
    assert not torch.equal(before, fc7)                                              # the packed copies were rebuilt
    same = (np.abs(rois.cpu().numpy() - g["rois"]) < 0.02).all(-1).reshape(-1)
    assert same.all(), same
    assert relerr(fc7.cpu().numpy(), g["fc7"]) < 1e-4
    assert relerr(pooled.cpu().numpy()[:, ::37], g["pooled_sub"]) < 1e-4
    assert sorted(model.state_dict().keys()) == [str(k) for k in g["state_keys"]]


def test_ground_checkpoint_roundtrip_with_optimizer_state(gpu, tmp_path):
    """vis_ground_{session}_{epoch}_{batch}.pth (model.py:1115-1126): model AND optimiser state written after step 1; a fresh
    process state (new model, new optimiser) that loads it and takes step 2 ends bit-identical to the run that never stopped."""
    from nafae_amd.checkpoint import checkpoint_name, load_ground_checkpoint, save_ground_checkpoint
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training, train_step
    Na, Ns, Ne, Nb = 2, 3, 4, 16
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
    b1 = make_batch(Na, Ns, Ne, H=64, W=96, seed=11, lens=[2, 4])
    b2 = make_batch(Na, Ns, Ne, H=64, W=96, seed=12, lens=[3, 1])
    model, opt, crit, red = setup_training(args, seed=5)
    train_step(model, opt, crit, b1, args, red)
    path = checkpoint_name(str(tmp_path), 'vgg16', 'YouCookII', 1, 3, 10021)
    save_ground_checkpoint(model, opt, 1, 3, path)
    ck = torch.load(path, map_location='cpu')
    assert set(ck.keys()) == {'session', 'epoch', 'model', 'optimizer', 'pooling_mode'}
    assert len(ck['optimizer']['state']) == len(red.params) and ck['optimizer']['param_groups'][0]['lr'] == args.lr
    # the reference's index space (model.py:1077-1082): three groups DVSA (10 parameters, never stepped: no state), word_ebd, vis_ebd
    assert [len(g['params']) for g in ck['optimizer']['param_groups']] == [10, 4, 2]
    assert sorted(ck['optimizer']['state'].keys()) == [10, 11, 12, 13, 14, 15]
    assert float(ck['optimizer']['state'][10]['exp_avg'].abs().max()) > 0
    train_step(model, opt, crit, b2, args, red)
    want = {k: v.clone() for k, v in model.state_dict().items()}
    # "new process": different seed, then load
    model2, opt2, crit2, red2 = setup_training(args, seed=99)
    assert load_ground_checkpoint(model2, path, resume=True) == 4
    opt2.load_state_dict(ck['optimizer'])
    assert opt2.step_count == 1
    train_step(model2, opt2, crit2, b2, args, red2)
    got = model2.state_dict()
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # without the optimiser state the second step differs (Adam's moments restart) -- the state is what made it equal
    model3, opt3, crit3, red3 = setup_training(args, seed=99)
    load_ground_checkpoint(model3, path, resume=True)
    train_step(model3, opt3, crit3, b2, args, red3)
    assert not torch.equal(model3.state_dict()['vis_ebd.fc1.weight'], want['vis_ebd.fc1.weight'])


def test_reference_optimizer_state_lands_on_the_right_parameters(gpu):
    """The 'optimizer' entry of a REFERENCE checkpoint is the state_dict of torch.optim.Adam over the three param groups
    DVSA, word_ebd, vis_ebd (model.py:1077-1082, :1118-1124).  Loading it must put every moment tensor on the parameter it
    belongs to although FusedClipAdam's flat buffer keeps vis_ebd first -- four of the six trainable tensors are [512]
    vectors, so a swap would not even raise -- and a state of the wrong shape must raise."""
    import copy
    from nafae_amd.model import default_args
    from nafae_amd.train import setup_training
    args = default_args(batch_size=2, sample_num=3, max_ent_len=4, dropout_rate=0.0)
    model, opt, crit, red = setup_training(args, seed=5)
    ref = copy.deepcopy(model)
    ropt = torch.optim.Adam([{'params': ref.DVSA.parameters()}, {'params': ref.word_ebd.parameters()},
                             {'params': ref.vis_ebd.parameters()}], lr=args.lr, weight_decay=args.weight_decay)
    g = torch.Generator(device="cuda").manual_seed(3)
    for m in (ref.word_ebd, ref.vis_ebd):
        for p in m.parameters():
            p.grad = torch.randn(p.shape, device="cuda", generator=g)
    ropt.step()
    sd = ropt.state_dict()
    assert sorted(sd['state'].keys()) == [10, 11, 12, 13, 14, 15]
    opt.load_state_dict(sd)
    assert opt.step_count == 1
    names = {id(p): n for n, p in model.named_parameters()}
    ref_state = {n: ropt.state[p] for n, p in ref.named_parameters() if p in ropt.state}
    o = 0
    for p in red.params:
        k, n = p.numel(), names[id(p)]
        assert torch.equal(opt.exp_avg[o:o + k].view_as(p), ref_state[n]['exp_avg']), n
        assert torch.equal(opt.exp_avg_sq[o:o + k].view_as(p), ref_state[n]['exp_avg_sq']), n
        o += k
    back = opt.state_dict()
    for i in sd['state']:
        assert torch.equal(back['state'][i]['exp_avg'], sd['state'][i]['exp_avg'])
    bad = copy.deepcopy(sd)
    bad['state'][14], bad['state'][10] = bad['state'][10], bad['state'][14]        # vis_ebd.fc1.weight <-> word_ebd.fc1.weight
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)


def test_stepRCNN_stream_vs_oracle_and_unbounded_chunk(gpu):
    """150 HOST frames (raw uint8, as decoded) through stepRCNN: (a) the reference's 64-frame chunks, (b) step_size=None,
    the chunk sized from free HBM with no 64-frame / 800-frame limit.  Both against the CPU ORACLE detector on a sampled
    subset of the frames (first / chunk borders / last), not against each other only."""
    from nafae_amd import synthetic as syn
    from nafae_amd.model import auto_step_size, default_args, stepRCNN
    from nafae_amd.train import build_model
    from oracle import detector as OD
    Nb = 16
    gpu.TEST.RPN_POST_NMS_TOP_N = Nb
    model = build_model(default_args(), seed=5).eval()
    Ns, H, W = 150, 64, 96
    gen = torch.Generator().manual_seed(9)
    u8 = torch.randint(0, 255, (Ns, H, W, 3), dtype=torch.uint8, generator=gen)
    f32 = (u8.float() - 127.5).permute(0, 3, 1, 2).contiguous()
    info = torch.tensor([[H, W, 1.0]]).repeat(Ns, 1)
    gt, nb = torch.zeros(1, 1, 5, device="cuda"), torch.zeros(1, device="cuda")
    a = stepRCNN(u8, info, gt, nb, model, need_roi_feats=False)
    n_auto = auto_step_size(model.fasterRCNN, tuple(u8.shape), "cuda", need_roi_feats=False)
    assert n_auto > 64
    b = stepRCNN(u8, info, gt, nb, model, step_size=None, need_roi_feats=False)
    assert tuple(b[0].shape) == (Ns, Nb, 5) and b[1] is None and tuple(b[2].shape) == (Ns * Nb, 4096)
    sd = syn.detector_state(seed=5, heads=False)
    ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=[4, 8, 16, 32], ANCHOR_RATIOS=[0.5, 1, 2], RPN_PRE_NMS_TOP_N=6000,
                RPN_POST_NMS_TOP_N=Nb, RPN_NMS_THRESH=0.7, POOLING_SIZE=7)
    pick = [0, 63, 64, 127, 128, 149]
    r_o, s_o, _, fc7_o = OD.detector_forward(f32[pick], info[pick], sd, ocfg)
    for out in (a, b):
        rois = out[0].cpu()[pick]
        fc7 = out[2].cpu().view(Ns, Nb, 4096)[pick].reshape(-1, 4096)
        r_cmp = r_o.clone()
        r_cmp[:, :, 0] = torch.tensor(pick, dtype=torch.float32)[:, None]          # col 0 = frame index WITHIN THE CHUNK in the reference
        chunk = 64 if out is a else n_auto
        r_cmp[:, :, 0] = torch.tensor([p % chunk for p in pick], dtype=torch.float32)[:, None]
        same = ((rois - r_cmp).abs() < 0.02).all(-1).view(-1).numpy()
        assert same.mean() >= 0.99, same.mean()
        assert relerr(fc7.numpy()[same], fc7_o.numpy()[same]) < 1e-4
    # the two chunkings agree with each other up to the stream-K summation order of the conv tiles
    assert relerr(a[2].cpu(), b[2].cpu()) < 1e-5
