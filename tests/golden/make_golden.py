"""Generates tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE (/root/reference) on seeded inputs.

Run once in the build container:   python tests/golden/make_golden.py
Only the resulting arrays (inputs + expected outputs) are committed; the reference sources never
travel.  Where the reference is CUDA-only (NMS, ROI-Align) the oracle's C restatement is plugged in
(see ref_harness.plug_native_ops), so those two fixtures pin the reference's *glue* around the
native ops, not the native ops themselves (which the reference cannot run on CPU at all).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import ref_harness as H  # noqa: E402
from nafae_amd import synthetic as syn  # noqa: E402
from oracle import native as onative  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def gold_anchors(m):
    from model.rpn.generate_anchors import generate_anchors
    save("anchors",
         default=generate_anchors(),
         vgg16_yml=generate_anchors(scales=np.array([4, 8, 16, 32]), ratios=np.array([0.5, 1, 2])))


DVSA_CASES = [
    # name, Na, Ns, Nb, Ne, lens, D, Delta, vis_lam
    ("c1", 2, 2, 32, 8, [3, 5], 512, 10.0, 4.13),          # BASELINE config C1 (Ns=2: degenerate clustering)
    ("c1b", 2, 4, 32, 8, [3, 5], 64, 10.0, 4.13),
    ("ragged", 3, 5, 20, 13, [2, 0, 4], 64, 1.0, 1.0),     # reference defaults Nb=20, Ne=13, a zero-length segment
    ("na1", 1, 4, 32, 8, [3], 64, 10.0, 4.13),             # Na=1: ranking term == 2*Delta
    ("full", 4, 3, 7, 5, [5, 1, 0, 2], 32, 5.0, 1.0),      # len == Ne, Nb not a multiple of anything
    ("big", 4, 6, 40, 6, [1, 6, 3, 2], 128, 10.0, 4.13),
]


def gold_dvsa(m):
    for (name, Na, Ns, Nb, Ne, lens, D, Delta, lam) in DVSA_CASES:
        V = torch.tanh(syn.randn(11, "V" + name, (Na * Ns * Nb, D)))
        W = torch.tanh(syn.randn(11, "W" + name, (Na * Ne, D)))
        out = dict(V=V, W=W, lens=np.array(lens), shape=np.array([Na, Ns, Nb, Ne, D]),
                   Delta=np.float32(Delta), vis_lam=np.float32(lam))
        args = H.make_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne,
                           dropout_rate=0.0, Delta=Delta, vis_lam=lam, word_ebd_dim=D)
        m.cfg.TEST.RPN_POST_NMS_TOP_N = Nb
        d = m.DVSA(args, m.cfg)
        for phase in ("train", "eval"):
            d.init_train() if phase == "train" else d.init_eval()
            v = V.clone().requires_grad_()
            w = W.clone().requires_grad_()
            Di, Ds, L = d(v, w, list(lens))
            L.backward()
            out.update({"D_ind_" + phase: Di, "D_sim_" + phase: Ds, "loss_" + phase: L,
                        "dV_" + phase: v.grad, "dW_" + phase: w.grad})
        Dp, Sp = m.postprocess(out["D_ind_eval"].numpy(), out["D_sim_eval"].detach().numpy(), Na, Ns, Nb, Ne)
        out.update(post_D=Dp, post_sim=Sp)
        save("dvsa_" + name, **out)


def gold_embed(m):
    """VisEbd / WordEbd (model.py:616-642) alone and chained into DVSA with gradients to the fc/bn params."""
    Na, Ns, Nb, Ne, D, FC, G = 2, 3, 8, 4, 32, 48, 16
    lens = [2, 3]
    args = H.make_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0,
                       vis_lam=4.13, word_ebd_dim=D, vis_fc_dim=FC, glove_dim=G)
    m.cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    ve, we, dv = m.VisEbd(args), m.WordEbd(args), m.DVSA(args, m.cfg)
    with torch.no_grad():
        ve.fc1.weight.copy_(syn.randn(5, "ve.w", (D, FC), 0.05)); ve.fc1.bias.copy_(syn.randn(5, "ve.b", (D,), 0.1))
        we.fc1.weight.copy_(syn.randn(5, "we.w", (D, G), 0.3)); we.fc1.bias.copy_(syn.randn(5, "we.b", (D,), 0.1))
        we.bn.weight.copy_(1 + syn.randn(5, "bn.w", (D,), 0.2)); we.bn.bias.copy_(syn.randn(5, "bn.b", (D,), 0.2))
    fc7 = torch.relu(syn.randn(5, "fc7", (Na * Ns * Nb, FC), 60.0))
    glove = syn.glove(Na, Ne, lens, dim=G, seed=5)
    out = dict(fc7=fc7, glove=glove, lens=np.array(lens), shape=np.array([Na, Ns, Nb, Ne, D, FC, G]),
               ve_w=ve.fc1.weight, ve_b=ve.fc1.bias, we_w=we.fc1.weight, we_b=we.fc1.bias,
               bn_w=we.bn.weight, bn_b=we.bn.bias)
    # train mode (BN batch statistics), dropout p=0
    ve.train(); we.train(); dv.init_train()
    V = ve(fc7); W = we(glove)
    Di, Ds, L = dv(V, W, lens)
    L.backward()
    out.update(V_train=V, W_train=W, loss_train=L, D_ind_train=Di,
               g_ve_w=ve.fc1.weight.grad, g_ve_b=ve.fc1.bias.grad, g_we_w=we.fc1.weight.grad,
               g_we_b=we.fc1.bias.grad, g_bn_w=we.bn.weight.grad, g_bn_b=we.bn.bias.grad,
               run_mean=we.bn.running_mean, run_var=we.bn.running_var)
    # eval mode (BN running statistics as updated by the one train step above)
    ve.eval(); we.eval(); dv.init_eval()
    with torch.no_grad():
        V = ve(fc7); W = we(glove)
        Di, Ds, L = dv(V, W, lens)
    out.update(V_eval=V, W_eval=W, loss_eval=L, D_ind_eval=Di, D_sim_eval=Ds)
    save("embed", **out)


def _nms_fn(dets, thresh):
    keep = onative.nms(dets.detach().numpy(), float(thresh))
    return torch.from_numpy(keep.astype(np.int32)).view(-1, 1)


def _ra_fn(features, rois, ah, aw, scale):
    return torch.from_numpy(onative.roi_align_forward(features.detach().numpy(), rois.detach().numpy(), ah, aw, scale))


def gold_proposal(m):
    """_ProposalLayer.forward (proposal_layer.py:49-171) + _RPN softmax pairing (rpn/rpn.py:63-72)."""
    from model.rpn.proposal_layer import _ProposalLayer
    from model.rpn.rpn import _RPN
    import torch.nn.functional as F
    H.plug_native_ops(_nms_fn, _ra_fn)
    cfg = m.cfg
    cfg.TEST.RPN_POST_NMS_TOP_N = 32
    Fr, A, Hh, Ww = 3, 12, 6, 5                       # non-square map, image 96 x 80
    cls = syn.randn(7, "cls", (Fr, 2 * A, Hh, Ww), 2.0)
    r = _RPN.reshape(cls, 2)
    prob = _RPN.reshape(F.softmax(r, dim=1), 2 * A)   # same call chain as rpn.py:67-69 (implicit dim == 1)
    deltas = syn.randn(7, "deltas", (Fr, 4 * A, Hh, Ww), 0.5)
    im_info = torch.tensor([[96, 80, 1.0]] * Fr)
    pl = _ProposalLayer(cfg.FEAT_STRIDE[0], cfg.ANCHOR_SCALES, cfg.ANCHOR_RATIOS)
    rois = pl((prob, deltas, im_info, 'TEST'))
    save("proposal", cls=cls, prob=prob, deltas=deltas, im_info=im_info, rois=rois,
         roi_scores=pl.get_roi_score(), post_nms_topN=np.int32(32),
         scales=np.array(cfg.ANCHOR_SCALES), ratios=np.array(cfg.ANCHOR_RATIOS))


def gold_detector(m):
    """Full frozen-detector forward of the reference (faster_rcnn/rpn.py:39-87) on 2 small frames with
    seeded synthetic weights (nafae_amd.synthetic.detector_state(seed=77))."""
    H.plug_native_ops(_nms_fn, _ra_fn)
    cfg = m.cfg
    cfg.TEST.RPN_POST_NMS_TOP_N = 8
    args = H.make_args()
    gm = m.GroundModel(args, cfg)
    sd = syn.detector_state(seed=77)
    gm.fasterRCNN.load_state_dict(sd, strict=True)
    gm.fasterRCNN.eval()
    im, im_info = syn.frames(2, 64, 48, seed=77)
    with torch.no_grad():
        rois, roi_scores, pooled, fc7 = gm.fasterRCNN(im, im_info, torch.zeros(1, 1, 5), torch.zeros(1))
        base = gm.fasterRCNN.RCNN_base(im)
    save("detector", seed=np.int32(77), frames_hw=np.array([64, 48]), post_nms_topN=np.int32(8),
         rois=rois, roi_scores=roi_scores, fc7=fc7, base_feat=base,
         pooled_sub=pooled[:, ::37], pooled_sum=pooled.double().sum(), state_keys=np.array(sorted(gm.state_dict().keys())))


def gold_eval(m):
    """record_det (model.py:477-487) + phrase/box accuracy (youcook_eval.py:135-336) on synthetic detections / gt."""
    import contextlib, io
    rs = np.random.RandomState(5)
    classes = ['bowl', 'egg', 'pan', 'oil', 'salt']
    num_imgs, Nb = 12, 6
    recs = []
    for i in range(num_imgs):
        k = rs.randint(0, 4)
        xy = rs.rand(k, 2) * 120
        recs.append({'label': [classes[j] for j in rs.randint(0, 5, k)],
                     'bbox': np.concatenate([xy, xy + 20 + rs.rand(k, 2) * 80], 1),
                     'thr': [0.5] * k, 'img_ids': [i] * k})
    # detections through the reference's record_det: Na=3 segments x Ns=4 frames, entity lists per segment
    Na, Ns, Ne = 3, 4, 3
    vid_entities = [['bowl', 'egg'], ['pan'], ['oil', 'salt', 'egg']]
    D = rs.randint(0, Nb, (Na, Ns, Ne)) + (np.arange(Na)[:, None, None] * Ns * Nb + np.arange(Ns)[None, :, None] * Nb)
    D_sim = rs.rand(Na, Ns, Ne)
    img_ids = list(rs.permutation(num_imgs))          # frame ids of the Na*Ns sampled frames (not sorted)
    bxy = rs.rand(Na * Ns * Nb, 2) * 120
    infer_boxes = np.concatenate([bxy, bxy + 20 + rs.rand(Na * Ns * Nb, 2) * 80], 1)
    # make some detections hit a gt box exactly
    for f in range(0, Na * Ns, 2):
        rec = recs[img_ids[f]]
        if len(rec['label']):
            infer_boxes[f * Nb:(f + 1) * Nb] = rec['bbox'][0] + rs.randn(Nb, 4) * 3
    dets = [[], [], [], []]
    m.record_det(dets[0], dets[1], dets[2], dets[3], Nb, vid_entities, D, D_sim, img_ids, infer_boxes)
    from datasets.youcook_eval import box_accuracy, phrase_accuracy
    with contextlib.redirect_stdout(io.StringIO()):
        pa = phrase_accuracy(recs, dets, classes)
        ba = box_accuracy(recs, dets, classes)
    rec_lab = np.array(['|'.join(r['label']) for r in recs])
    rec_box = np.zeros((num_imgs, 3, 4)); rec_n = np.zeros(num_imgs, dtype=int)
    for i, r in enumerate(recs):
        rec_n[i] = len(r['label']); rec_box[i, :rec_n[i]] = r['bbox']
    save("eval", classes=np.array(classes), rec_lab=rec_lab, rec_box=rec_box, rec_n=rec_n, D=D, D_sim=D_sim,
         img_ids=np.array(img_ids), infer_boxes=infer_boxes, Nb=np.int32(Nb),
         ent=np.array(['|'.join(e) for e in vid_entities]),
         det_img=np.array(dets[0]), det_lab=np.array(dets[1]), det_box=np.array(dets[2]), det_conf=np.array(dets[3]),
         phrase_acc=np.float64(pa), box_acc=np.float64(ba))


if __name__ == "__main__":
    torch.manual_seed(0)
    np.random.seed(3)
    m = H.load_reference()
    m.cfg_from_file(os.path.join(H.REF, "cfgs", "vgg16.yml"))
    gold_anchors(m)
    gold_dvsa(m)
    gold_embed(m)
    gold_proposal(m)
    gold_detector(m)
    gold_eval(m)
