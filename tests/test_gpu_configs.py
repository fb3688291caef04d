"""Every BASELINE.json configuration AT ITS SIZE on the GPU, against the CPU oracle.

  C2  64 frames 224x224 (Na=8, Ns=8), 128 proposals/frame, 16 query slots: detector + embeddings + DVSA + backward, in the
      exact-fp32 and the split-bf16 arithmetic, against tests/golden/config_c2.npz (oracle outputs precomputed in the build
      container by tests/golden/make_config_golden.py: ~0.5-3 s/frame of CPU is too slow to repeat here).
  C3  the same at plain bf16, with the bf16 tolerance written below.
  C4  per-GPU shape (8,8,256,32) and
  C5  (8,8,300,64): similarity forward, loss tail, clustering term and backward against oracle.dvsa run here on the host.

The measured agreement rates are printed (pytest -s / the captured log) and the asserted thresholds are the measured values
with a small margin, not round numbers.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def box_iou(a, b):
    """IoU with the +1 width convention of the reference (nms_cuda_kernel.cu:31-39); a [n,4], b [m,4] -> [n,m]."""
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.maximum(x2 - x1 + 1, 0) * np.maximum(y2 - y1 + 1, 0)
    sa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    sb = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (sa[:, None] + sb[None, :] - inter)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


_CONFIG_CACHE = {}


def load_config(name):
    """Model + seeded batch + oracle fixture of one BASELINE config at size (cached: the 4096 x 25088 fc6 weight alone takes
    seconds to draw).  Only one config's model is kept alive."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if name in _CONFIG_CACHE:
        return _CONFIG_CACHE[name]
    _CONFIG_CACHE.clear()
    torch.cuda.empty_cache()
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training
    g = np.load(os.path.join(G, "config_%s.npz" % name))
    Na, Ns, Nb, Ne = [int(x) for x in g["shape"]]
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=float(g["Delta"]),
                        vis_lam=float(g["vis_lam"]))
    model, opt, crit, reducer = setup_training(args, device="cuda", seed=int(g["seed"]))
    batch = make_batch(Na, Ns, Ne, seed=int(g["seed"]), device="cuda")
    assert batch.entities_length == g["lens"].tolist()
    _CONFIG_CACHE[name] = dict(g=g, cfg=cfg, args=args, model=model, opt=opt, crit=crit, reducer=reducer, batch=batch,
                               dims=(Na, Ns, Nb, Ne))
    return _CONFIG_CACHE[name]


# precision: (base_feat / fc7 / V tolerance relative to tensor scale, min identical-ROI fraction, loss tolerance,
#             min D_ind agreement on comparable entries, gradient tolerance)
C2_BARS = {
    # measured (round 2): f32 8192/8192 identical rois with one conv tile per workgroup; on the stream-K conv schedules that are
    # the default a cut tile sums its K range as two fp32 chains, and proposals at an NMS / top-N near-tie can flip against the
    # oracle's own summation order: 8190/8192 with stream-K over whole layers, 8192/8192 with the split schedule that shipped;
    # bf16x3 8182/8192 (59 of 64 frames with all 128 identical)
    "f32": dict(feat=1e-4, rois=0.9995, loss=1e-4, dind=1.0, grad=5e-4, ground=0.999),
    "bf16x3": dict(feat=1e-4, rois=0.998, loss=1e-4, dind=1.0, grad=5e-4, ground=0.995),
    # C4 / C5 (round 3, measured): 256 / 300 proposals per frame reach far down the score list, where neighbouring candidates
    # are closer than the arithmetic noise more often, so more top-N / NMS decisions flip against the oracle's summation order:
    #   f32     16380/16384 and 19194/19200 identical rois (62 / 61 of 64 frames entirely), D_ind 1051/1051 and 1034/1034 decided
    #           entries, every grounded box identical, loss 2.3e-6 / 3.2e-6;
    #   bf16x3  16270/16384 and 18960/19200 (47 / 39 frames entirely), D_ind 796/796 and 661/661, grounded box identical 0.9825 /
    #           0.9871 (IoU >= 0.5: 1.0 / 0.9991).  A "same" proposal may still differ by up to 0.02 px, which moves its ROI-Align
    #           samples: V / D_sim of such rows agree to 1.4e-4 / 1.7e-4 only, hence feat_soft (fc7 itself: 2.8e-5).
    ("c4", "f32"): dict(feat=1e-4, rois=0.9995, loss=1e-4, dind=1.0, grad=5e-4, ground=0.999),
    ("c5", "f32"): dict(feat=1e-4, rois=0.9995, loss=1e-4, dind=1.0, grad=5e-4, ground=0.999),
    ("c4", "bf16x3"): dict(feat=1e-4, feat_soft=4e-4, rois=0.990, loss=1e-4, dind=1.0, grad=5e-4, ground=0.975),
    ("c5", "bf16x3"): dict(feat=1e-4, feat_soft=4e-4, rois=0.984, loss=1e-4, dind=1.0, grad=5e-4, ground=0.975),
    # BASELINE config C3: bf16 operands (8-bit mantissa), fp32 accumulation.  Stated tolerance: 3e-2 of the tensor scale on
    # features and 2e-2 on the loss; proposals and grounding are compared geometrically (a 1e-2 feature error moves box
    # coordinates by more than 0.02 px and flips NMS decisions, so index-wise comparison is meaningless): the share of oracle
    # proposals that have a HIP proposal with IoU >= 0.9 in the same frame, and the share of live (frame, query) pairs whose
    # grounded box overlaps the oracle's grounded box with IoU >= 0.5.
    # Measured (rounds 2 and 3): 0.9825 of the oracle's proposals matched, 0.9507 of the grounded boxes at IoU >= 0.5; the bars
    # sit a few points below that, so a regression that halves the agreement fails.
    "bf16": dict(feat=3e-2, rois=None, loss=2e-2, dind=None, grad=None, ground=None, roi_iou=0.95, ground_iou=0.90),
}


FULL_CASES = [("c2", "f32"), ("c2", "bf16x3"), ("c2", "bf16"),       # C2 and (bf16) C3: 128 proposals, 16 query slots
              ("c4", "f32"), ("c4", "bf16x3"),                       # C4 per-GPU share: 256 proposals, 32 query slots
              ("c5", "f32"), ("c5", "bf16x3")]                       # C5: 300 proposals, 64 query slots


@pytest.mark.parametrize("name,precision", FULL_CASES, ids=["%s-%s" % c for c in FULL_CASES])
def test_full_size_detector_and_grounding(name, precision, capsys):
    """Every BASELINE config at 64 frames -- C2 (f32, bf16x3) / C3 (bf16) with 128 proposals x 16 query slots, C4's per-GPU
    share (256 x 32: ROI-Align of 16 384 ROIs, fc6 at M = 16 384, top-N 256 after NMS) and C5 (300 x 64: 19 200 ROIs) -- end
    to end (detector, embeddings, DVSA, backward) against the oracle fixture of that config."""
    from nafae_amd import ops
    c2 = load_config(name)
    g, model, batch = c2["g"], c2["model"], c2["batch"]
    Na, Ns, Nb, Ne = c2["dims"]
    bars = C2_BARS.get((name, precision), C2_BARS[precision])
    c2["cfg"].TEST.RPN_POST_NMS_TOP_N = Nb
    fr = model.fasterRCNN
    fr.precision = precision
    F, Q = Na * Ns, Na * Ne
    lens = batch.entities_length

    base = fr.base_features(batch.im_data)
    base = ops.merge_bf16(base) if isinstance(base, ops.Planes) else base
    e_base = max(relerr(base[0].permute(2, 0, 1).cpu(), g["base_feat_f0"]) * np.abs(g["base_feat_f0"]).max(),
                 relerr(base[F - 1].permute(2, 0, 1).cpu(), g["base_feat_f63"]) * np.abs(g["base_feat_f63"]).max()) / float(g["base_absmax"])
    rois, roi_scores, pooled, fc7 = fr(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
    assert tuple(rois.shape) == (F, Nb, 5) and tuple(fc7.shape) == (F * Nb, 4096)
    rois_np = rois.cpu().numpy()
    same = (np.abs(rois_np - g["rois"]) < 0.02).all(-1)                             # [F, Nb]: same proposal within 0.02 px
    frame_ok = same.all(1)
    roi_iou = np.mean([(box_iou(g["rois"][f_, :, 1:], rois_np[f_, :, 1:]).max(1) >= 0.9).mean() for f_ in range(F)])
    fr_rows = g["fc7_rows"]
    row_same = same.reshape(-1)[fr_rows]
    e_fc7 = float(np.abs(fc7[torch.from_numpy(fr_rows).cuda()].cpu().numpy()[row_same] - g["fc7_sample"][row_same]).max()
                  / float(g["fc7_absmax"])) if row_same.any() else float("nan")

    model.train(); model.DVSA.init_train(); fr.eval()
    c2["reducer"].zero_grad()
    V = model.vis_ebd(fc7)
    W = model.word_ebd(batch.glove_feats)
    e_W = relerr(W.detach().cpu(), g["W"])
    v_same = same.reshape(-1)[g["v_rows"]]
    e_V = relerr(V.detach()[torch.from_numpy(g["v_rows"]).cuda()].cpu().numpy()[v_same], g["V_sample"][v_same]) if v_same.any() else float("nan")
    D_ind, D_sim, L = model.DVSA(V, W, lens)
    loss = c2["crit"](L, torch.zeros_like(L))
    loss.backward()
    # grounding indices: comparable where the frame's whole proposal set is the oracle's and the query slot is live;
    # a mismatch is legitimate only where the oracle's own top-2 gap is below the noise of the features feeding it
    live = np.zeros((Na, Ne), dtype=bool)
    for a_, l in enumerate(lens):
        live[a_, :l] = True
    comparable = frame_ok[:, None] & live.reshape(1, Q)
    scale = float(np.abs(g["D_sim"]).max())
    decided = g["top2_gap"] >= 1e-4 * scale
    Dg = D_ind.cpu().numpy()
    agree = (Dg == g["D_ind"])
    n_cmp = int((comparable & decided).sum())
    dind_rate = float(agree[comparable & decided].mean()) if n_cmp else float("nan")
    n_close = int((comparable & ~decided).sum())
    e_sim = float(np.abs(D_sim.cpu().numpy() - g["D_sim"])[comparable].max() / scale) if comparable.any() else float("nan")
    e_loss = abs(float(L.detach()) - float(g["loss"])) / abs(float(g["loss"]))
    # grounded boxes (what the evaluation consumes): box of the arg-max proposal, HIP vs oracle, for every live (frame, query)
    fi, qi = np.nonzero(np.broadcast_to(live.reshape(1, Q), (F, Q)))
    gb_hip = rois_np[fi, Dg[fi, qi], 1:]
    gb_ora = g["rois"][fi, g["D_ind"][fi, qi], 1:]
    giou = np.array([box_iou(gb_hip[k:k + 1], gb_ora[k:k + 1])[0, 0] for k in range(len(fi))])
    ground_same, ground_half = float((giou >= 0.999).mean()), float((giou >= 0.5).mean())
    with capsys.disabled():
        print("\n[" + name.upper() + " %-6s] base_feat %.2e | identical rois %.4f (%d/%d), frames with all %d rois identical %d/%d | fc7 %.2e | V %.2e W %.2e"
              " | D_ind agree %.5f on %d decided entries (%d near-ties excluded) | D_sim %.2e | loss %.6f vs %.6f (rel %.1e)"
              " | oracle rois matched at IoU>=0.9: %.4f | grounded box identical %.4f, IoU>=0.5 %.4f (%d live pairs)"
              % (precision, e_base, same.mean(), same.sum(), same.size, Nb, frame_ok.sum(), F, e_fc7, e_V, e_W, dind_rate, n_cmp,
                 n_close, e_sim, float(L.detach()), float(g["loss"]), e_loss, roi_iou, ground_same, ground_half, len(fi)))
    assert e_base < bars["feat"], e_base
    assert e_W < 1e-4
    if precision == "bf16":
        assert roi_iou >= bars["roi_iou"] and ground_half >= bars["ground_iou"], (roi_iou, ground_half)
        assert e_loss < bars["loss"], e_loss
        return
    assert same.mean() >= bars["rois"], same.mean()
    assert e_fc7 < bars["feat"], e_fc7
    assert np.isnan(e_V) or e_V < bars.get("feat_soft", bars["feat"])
    assert n_cmp > 0 and dind_rate >= bars["dind"], (dind_rate, n_cmp)
    assert ground_same >= bars["ground"], ground_same
    assert e_sim < bars.get("feat_soft", 1e-4), e_sim
    if frame_ok.all():
        assert e_loss < bars["loss"], e_loss
    else:       # a frame whose proposal set differs feeds different rows into the loss: bounded, not equal
        assert e_loss < max(bars["loss"], 2e-2), e_loss
    if bars["grad"] is not None and frame_ok.all():
        we, ve = model.word_ebd, model.vis_ebd
        for p, k in ((ve.fc1.bias, "g_ve_b"), (we.fc1.weight, "g_we_w"), (we.bn.weight, "g_bn_w"), (we.bn.bias, "g_bn_b")):
            err = np.abs(p.grad.cpu().numpy().astype(np.float64) - g[k]).max()
            assert err < bars["grad"] * max(np.abs(g[k]).max(), 1e-2), (k, err)
        gw = ve.fc1.weight.grad
        assert relerr(gw[:8].cpu(), g["g_ve_w_rows"]) < bars["grad"] * max(1.0, np.abs(gw.cpu().numpy()).max() / max(np.abs(g["g_ve_w_rows"]).max(), 1e-30))
        assert abs(float(gw.double().norm()) - float(g["g_ve_w_norm"])) < bars["grad"] * float(g["g_ve_w_norm"])


def synthetic_ground_truth(Na, Ns, lens, seed, H=224, W=224, vocab=("bowl", "egg", "pan", "oil", "salt", "water", "knife", "rice")):
    """Seeded ground truth that depends on NEITHER detection path: per segment the entity labels (a small vocabulary, repeats
    allowed -- youcook_eval's per-frame label bookkeeping sees them), per (frame, entity) one gt box drawn uniformly with sides of
    60..200 px and an individual IoU threshold from {0.1, 0.3, 0.5} (youcook_eval.py's `thr` field), so that the accuracy is a
    mid-range number on which a changed grounded box can show.  -> (vid_entities, recs, class_list) in the formats of
    model.py:906-909 and youcook_eval.parse_gt."""
    rs = np.random.RandomState(seed)
    vid_entities = [[vocab[i] for i in rs.randint(0, len(vocab), l)] for l in lens]
    recs = []
    for a in range(Na):
        for s_ in range(Ns):
            labels, boxes, thrs = [], [], []
            for ent in vid_entities[a]:
                w, h = rs.randint(60, 201), rs.randint(60, 201)
                x1, y1 = rs.randint(0, W - w + 1), rs.randint(0, H - h + 1)
                labels.append(ent); boxes.append([float(x1), float(y1), float(x1 + w - 1), float(y1 + h - 1)])
                thrs.append(float(rs.choice([0.1, 0.3, 0.5])))
            recs.append({"label": labels, "bbox": boxes, "thr": thrs, "img_ids": [a * Ns + s_] * len(labels)})
    return vid_entities, recs, list(vocab)


def accuracies(rois, D_ind, D_sim, dims, vid_entities, recs, class_list):
    """Detections of one path through the reference's own chain: postprocess (model.py:457-474) -> record_det (:477-487) ->
    box_accuracy / phrase_accuracy (youcook_eval.py:241-336, :135-237).  -> (macro box, micro box, macro phrase, micro phrase)."""
    from nafae_amd import evaluate as E
    from nafae_amd.model import postprocess
    Na, Ns, Nb, Ne = dims
    Dp, Sp = postprocess(np.asarray(D_ind), np.asarray(D_sim), Na, Ns, Nb, Ne)
    dets = [[], [], [], []]
    E.record_det(dets[0], dets[1], dets[2], dets[3], Nb, vid_entities, Dp, Sp, list(range(Na * Ns)),
                 np.asarray(rois)[:, :, 1:5].reshape(-1, 4))
    out = []
    for fn in (E.box_accuracy, E.phrase_accuracy):
        out += list(fn(recs, dets, class_list, both=True))
    return out, dets


ACC_CASES = [("c2", "f32"), ("c2", "bf16x3"), ("c4", "f32"), ("c4", "bf16x3"), ("c5", "f32"), ("c5", "bf16x3")]


@pytest.mark.parametrize("name,precision", ACC_CASES, ids=["%s-%s" % c for c in ACC_CASES])
def test_grounding_accuracy_delta_vs_oracle(name, precision, capsys):
    """north_star: "grounding accuracy within +-0.1 % of the reference on identical inputs".  The HIP detections of the config's
    seeded 64-frame batch and the ORACLE's detections of the same batch (rebuilt from the fixture's rois / D_ind / D_sim) go
    through postprocess -> record_det -> box_accuracy and phrase_accuracy (model.py:457-487, youcook_eval.py:135-336) against one
    synthetic ground truth that is independent of both; |delta| <= 0.001 on all four figures."""
    c = load_config(name)
    g, model, batch = c["g"], c["model"], c["batch"]
    Na, Ns, Nb, Ne = c["dims"]
    c["cfg"].TEST.RPN_POST_NMS_TOP_N = Nb
    fr = model.fasterRCNN
    fr.precision = precision
    # (the fixture's W comes from WordEbd in TRAIN mode -- BatchNorm on batch statistics, dropout_rate 0 -- like every other test
    # on these fixtures; D_ind / D_sim do not depend on DVSA's phase, model.py:610-612)
    model.train(); model.DVSA.init_train(); fr.eval()
    with torch.no_grad():
        rois, roi_scores, pooled, fc7 = fr(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
        V = model.vis_ebd(fc7)
        W = model.word_ebd(batch.glove_feats)
        D_ind, D_sim, L = model.DVSA(V, W, batch.entities_length)
    # validate() records the own-segment pairs only (Ns x sum(lens) = 152 detections here): one changed match would be 0.66 % of
    # one ground truth, so the comparison is pooled over N_GT independent ground truths (mean of each figure), which resolves 0.08 %
    N_GT = 8
    hip_all, ora_all, moved, n_det = [], [], 0, 0
    r_h, d_h, s_h = rois.cpu().numpy(), D_ind.cpu().numpy(), D_sim.cpu().numpy()
    for k in range(N_GT):
        vid_entities, recs, class_list = synthetic_ground_truth(Na, Ns, batch.entities_length, seed=4242 + 17 * k + Nb)
        hip, dets_h = accuracies(r_h, d_h, s_h, c["dims"], vid_entities, recs, class_list)
        ora, dets_o = accuracies(g["rois"], g["D_ind"], g["D_sim"], c["dims"], vid_entities, recs, class_list)
        hip_all.append(hip); ora_all.append(ora)
        n_det = len(dets_h[0])
        assert n_det == len(dets_o[0]) == Ns * sum(batch.entities_length)
        moved = int(sum(1 for bh, bo in zip(dets_h[2], dets_o[2]) if np.abs(np.asarray(bh) - np.asarray(bo)).max() >= 0.02))
    hip_all, ora_all = np.array(hip_all), np.array(ora_all)
    hip, ora = hip_all.mean(0), ora_all.mean(0)
    worst_single = float(np.abs(hip_all - ora_all).max())
    with capsys.disabled():
        print("\n[%s %-6s accuracy, mean over %d ground truths] box macro %.4f vs oracle %.4f | box micro %.4f vs %.4f | phrase macro "
              "%.4f vs %.4f | phrase micro %.4f vs %.4f | %d detections, %d grounded boxes differ by >= 0.02 px | max |delta| of the "
              "means %.5f (worst single ground truth %.5f)"
              % (name.upper(), precision, N_GT, hip[0], ora[0], hip[1], ora[1], hip[2], ora[2], hip[3], ora[3], n_det, moved,
                 float(np.abs(hip - ora).max()), worst_single))
    assert 0.02 < ora[1] < 0.98, "the synthetic ground truth must give a mid-range accuracy"
    assert float(np.abs(hip - ora).max()) <= 0.001, (hip, ora)


@pytest.mark.parametrize("name", ["c4", "c5"])
def test_full_step_at_size_properties(name, capsys):
    """C4's per-GPU share and C5 as FULL TRAINING STEPS on one GPU (what `bench.py --gpus 8` runs per rank): two pipelined
    steps (detector of step k+1 overlapping the tail of step k), then the size-independent properties: every frame keeps
    exactly top-N proposals whose rows are inside the frame and sorted by score, padding rows are zero, the loss is finite
    and positive, the step is bit-reproducible (same inputs, same state -> identical rois / fc7 / D_ind / loss / gradients),
    and the optimiser moved exactly the parameters that own gradients."""
    from nafae_amd.train import PipelinedTrainer, detector_forward
    c = load_config(name)
    model, batch, reducer, opt, crit, args = c["model"], c["batch"], c["reducer"], c["opt"], c["crit"], c["args"]
    Na, Ns, Nb, Ne = c["dims"]
    F, Q = Na * Ns, Na * Ne
    c["cfg"].TEST.RPN_POST_NMS_TOP_N = Nb
    fr = model.fasterRCNN
    fr.precision = "f32"
    model.train(); model.DVSA.init_train(); fr.eval()

    def fwd_bwd():
        rois, roi_scores, _, fc7 = detector_forward(model, batch)
        reducer.zero_grad()
        V = model.vis_ebd(fc7)
        W = model.word_ebd(batch.glove_feats)
        D_ind, D_sim, L = model.DVSA(V, W, batch.entities_length)
        L.backward()
        return rois, roi_scores, fc7, D_ind, D_sim, L.detach().clone(), reducer.flat.clone(), fr.n_keep.clone()

    a = fwd_bwd()
    b = fwd_bwd()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    rois, roi_scores, fc7, D_ind, D_sim, L, grad, n_keep = a
    assert tuple(rois.shape) == (F, Nb, 5) and tuple(fc7.shape) == (F * Nb, 4096) and tuple(D_ind.shape) == (F, Q)
    nk = n_keep.cpu().numpy()
    assert ((nk >= 1) & (nk <= Nb)).all()
    r = rois.cpu().numpy(); sc = roi_scores.cpu().numpy()
    for f in range(F):
        k = nk[f]
        assert (r[f, :, 0] == f).all()                                  # column 0 = frame index on every row (proposal_layer.py:153-163)
        assert (r[f, :k, 1] >= 0).all() and (r[f, :k, 3] <= 223).all() and (r[f, :k, 2] >= 0).all() and (r[f, :k, 4] <= 223).all()
        assert (np.diff(sc[f, :k]) <= 0).all()                          # kept in descending score order
        assert (r[f, k:, 1:] == 0).all() and (sc[f, k:] == 0).all()     # rows beyond n_keep stay zero
    assert torch.isfinite(fc7).all() and torch.isfinite(D_sim).all() and torch.isfinite(grad).all()
    assert float(L) > 0 and np.isfinite(float(L))
    Di = D_ind.cpu().numpy()
    assert (Di >= 0).all() and (Di < Nb).all()
    # two real optimiser steps through the pipelined trainer
    p0 = torch.cat([p.detach().reshape(-1) for p in reducer.params])
    pipe = PipelinedTrainer(model, opt, crit, args, reducer)
    pipe.submit(batch)
    l1 = pipe.step(batch)[0]
    l2 = pipe.step(None)[0]
    torch.cuda.synchronize()
    p1 = torch.cat([p.detach().reshape(-1) for p in reducer.params])
    assert np.isfinite(float(l1)) and np.isfinite(float(l2))
    assert abs(float(l1) - float(L)) <= 1e-6 * abs(float(L))           # first pipelined step = the forward above
    assert not torch.equal(p0.reshape(-1), p1)
    with capsys.disabled():
        print("\n[%s full step] kept %d..%d of %d proposals/frame | loss %.6f -> %.6f after one Adam step | |grad| %.4e"
              % (name.upper(), nk.min(), nk.max(), Nb, float(l1), float(l2), float(grad.norm())))
    _CONFIG_CACHE.clear()        # the optimiser moved the parameters: the cached model no longer matches the fixture


SIM_CONFIGS = {
    # name: (Na, Ns, Nb, Ne)
    "C4": (8, 8, 256, 32),
    "C5": (8, 8, 300, 64),
}


@pytest.mark.parametrize("lens_kind", ["histogram", "all_live", "ragged"])
@pytest.mark.parametrize("name", ["C4", "C5"])
def test_sim_loss_full_size_vs_oracle(name, lens_kind, capsys):
    """C4's per-GPU shape and C5: sim+max forward, loss tail, clustering term and backward at R x Q = 16384 x 256 /
    19200 x 512 against oracle.dvsa (the restatement of model.py:517-614) evaluated on the host in this test."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    from oracle import dvsa as O
    Na, Ns, Nb, Ne = SIM_CONFIGS[name]
    F, Q, R, D = Na * Ns, Na * Ne, Na * Ns * Nb, 512
    lens = {"histogram": syn.entity_lengths(Na, Ne, seed=1234), "all_live": [Ne] * Na,
            "ragged": [Ne, 0, 1, Ne // 2, 3, 0, Ne - 1, 2][:Na]}[lens_kind]
    V, W = syn.embeddings(R, Q, D, seed=3)
    torch.set_num_threads(os.cpu_count() or 1)
    Vo, Wo = V.clone().requires_grad_(), W.clone().requires_grad_()
    Di_o, Ds_o, L_o, parts = O.dvsa_forward(Vo, Wo, lens, Na, Nb, Ne, 10.0, 4.13, 'train', return_parts=True)
    L_o.backward()
    with torch.no_grad():
        top2 = (V @ W.t()).view(F, Nb, Q).topk(2, dim=1)[0]
        gap = (top2[:, 0] - top2[:, 1]).numpy()
    Vg, Wg = V.cuda(), W.cuda()
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    S_max, D_ind = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne)
    loss_out, dS, ws = ops.loss_fwd_bwd(S_max, D_ind, Vg, lt, Na, Ns, Nb, Ne, 10.0, 4.13, True)
    dV, dW = ops.sim_bwd(dS, D_ind, Vg, Wg, lt, Na, Ns, Nb, Ne, True, ws)
    live = np.zeros((Na, Ne), dtype=bool)
    for a_, l in enumerate(lens):
        live[a_, :l] = True
    live = np.broadcast_to(live.reshape(1, Q), (F, Q))
    scale = float(Ds_o.abs().max())
    Dg, Do = D_ind.cpu().numpy(), Di_o.numpy()
    decided = gap >= 1e-4 * scale
    mism = (Dg != Do) & live
    e_S = float((S_max.cpu() - Ds_o.detach()).abs().max() / scale)
    e_L = abs(float(loss_out[0]) - float(L_o)) / abs(float(L_o))
    e_vis = abs(float(loss_out[2]) - float(parts['vis_loss'])) / max(abs(float(parts['vis_loss'])), 1e-30)
    e_dV, e_dW = relerr(dV.cpu(), Vo.grad), relerr(dW.cpu(), Wo.grad)
    with capsys.disabled():
        print("\n[%s %-9s R=%d Q=%d live=%d] D_ind mismatches %d (of which oracle near-ties %d) | S_max %.2e | loss %.6f vs %.6f "
              "(rel %.1e) | vis_loss rel %.1e dem %d vs %d | dV %.2e dW %.2e"
              % (name, lens_kind, R, Q, int(live[0].sum()), int(mism.sum()), int((mism & ~decided).sum()), e_S, float(loss_out[0]),
                 float(L_o), e_L, e_vis, int(loss_out[3]), parts['dem'], e_dV, e_dW))
    assert not (mism & decided).any(), "D_ind differs from the oracle where the oracle's top-2 gap is >= 1e-4 * scale"
    assert (Dg[~live] == 0).all() and (S_max.cpu().numpy()[~live] == 0).all()        # masked slots: (0, 0) like model.py:551
    assert e_S < 1e-4, e_S
    assert int(loss_out[3]) == parts['dem']
    assert e_L < 1e-4 and e_vis < 1e-4, (e_L, e_vis)
    if not mism.any():
        assert e_dV < 5e-4 and e_dW < 5e-4, (e_dV, e_dW)


def _random_simloss_cases(n, seed):
    import random
    rs = random.Random(seed)
    out = []
    for _ in range(n):
        Na, Ns = rs.randint(1, 6), rs.randint(1, 6)
        Nb = rs.choice([1, 7, 20, 32, 33, 64, 100, 128])
        Ne = rs.choice([1, 4, 8, 13, 16, 32])
        kind = rs.random()
        lens = [Ne] * Na if kind < 0.25 else ([rs.randint(0, Ne) for _ in range(Na)])
        if sum(lens) == 0:
            lens[0] = max(1, Ne // 2)          # (the reference skips batches without any entity, model.py:685-686)
        out.append((Na, Ns, Nb, Ne, lens, rs.choice([64, 128, 512])))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", _random_simloss_cases(36, 77), ids=lambda c: "Na%d_Ns%d_Nb%d_Ne%d_D%d_L%d" % (c[:4] + (c[5], sum(c[4]))))
def test_sim_loss_random_shapes_vs_oracle(case):
    """Seeded random sweep of the whole similarity + ranking / clustering loss + backward chain (simmax.hip, simloss.hip:
    live-column and dense paths, LDS and global-memory loss tails, ragged segment lengths) against oracle.dvsa."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from nafae_amd import ops
    from nafae_amd import synthetic as syn
    from oracle import dvsa as O
    Na, Ns, Nb, Ne, lens, D = case
    F, Q, R = Na * Ns, Na * Ne, Na * Ns * Nb
    torch.set_num_threads(8)      # (the full-size tests above raise it to every host thread, which slows these small shapes ~20x)
    V, W = syn.embeddings(R, Q, D, seed=5 + Nb)
    Vo, Wo = V.clone().requires_grad_(), W.clone().requires_grad_()
    Di_o, Ds_o, L_o, parts = O.dvsa_forward(Vo, Wo, lens, Na, Nb, Ne, 10.0, 4.13, 'train', return_parts=True)
    L_o.backward()
    with torch.no_grad():
        S = (V @ W.t()).view(F, Nb, Q)
        gap = (S.topk(2, dim=1)[0][:, 0] - S.topk(2, dim=1)[0][:, 1]).numpy() if Nb > 1 else np.full((F, Q), np.inf)
    Vg, Wg = V.cuda(), W.cuda()
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    S_max, D_ind = ops.sim_max_fwd(Vg, Wg, lt, Na, Ns, Nb, Ne)
    loss_out, dS, ws = ops.loss_fwd_bwd(S_max, D_ind, Vg, lt, Na, Ns, Nb, Ne, 10.0, 4.13, True)
    dV, dW = ops.sim_bwd(dS, D_ind, Vg, Wg, lt, Na, Ns, Nb, Ne, True, ws)
    live = np.zeros((Na, Ne), dtype=bool)
    for a_, l in enumerate(lens):
        live[a_, :l] = True
    live = np.broadcast_to(live.reshape(1, Q), (F, Q))
    scale = max(float(Ds_o.detach().abs().max()), 1e-6)
    decided = gap >= 1e-4 * scale
    Dg, Do = D_ind.cpu().numpy(), Di_o.numpy()
    mism = (Dg != Do) & live
    assert (Dg[~live] == 0).all() and (S_max.cpu().numpy()[~live] == 0).all()
    assert float((S_max.cpu() - Ds_o.detach()).abs().max() / scale) < 1e-4
    if (mism & decided).any():
        pytest.fail("D_ind differs from the oracle at decided entries")
    if np.isnan(float(L_o.detach())):   # degenerate batches (one frame per segment: the reference's own clustering term is 0/0): NaN both
        assert np.isnan(float(loss_out[0]))
        return
    if not mism.any():          # (an undecided arg-max feeds another row into the clustering term: compare the losses only when equal)
        assert abs(float(loss_out[0]) - float(L_o.detach())) < 1e-4 * max(abs(float(L_o.detach())), 1e-3)
        assert int(loss_out[3]) == parts['dem']
        assert relerr(dV.cpu(), Vo.grad) < 5e-4 and relerr(dW.cpu(), Wo.grad) < 5e-4

