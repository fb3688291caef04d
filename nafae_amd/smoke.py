"""One small invocation of the whole hot path on cuda:0, checked against the CPU oracle
(BASELINE config C1: 4 frames 224x224, 32 proposals/frame, 8 query slots, Na=2, Ns=2, lens [3,5])."""
import os

import torch


def run(verbose=True):
    from . import _lib
    from . import synthetic as syn
    from .config import cfg, cfg_from_file, reset_cfg
    from .model import default_args
    from .train import make_batch, setup_training, train_step
    # the oracle is the checker here, never the thing being run
    from oracle import detector as OD
    from oracle import dvsa as O

    assert torch.cuda.is_available(), "smoke needs a GPU"
    reset_cfg()
    cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgs", "vgg16.yml"))
    Na, Ns, Ne, Nb = 2, 2, 8, 32
    cfg.TEST.RPN_POST_NMS_TOP_N = Nb
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
    model, opt, crit, _ = setup_training(args, device='cuda:0', seed=1234)
    batch = make_batch(Na, Ns, Ne, seed=1234, device='cuda:0', lens=[3, 5])

    # forward pieces, each checked against the oracle fed with the SAME inputs
    fr = model.fasterRCNN
    base = fr.base_features(batch.im_data)
    sd = syn.detector_state(seed=1234, heads=False)
    base_o = OD.vgg16_features(batch.im_data.cpu(), sd)
    e_base = float((base.permute(0, 3, 1, 2).cpu() - base_o).abs().max() / base_o.abs().max())
    rois, roi_scores, pooled, fc7 = fr(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
    ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=cfg.ANCHOR_SCALES, ANCHOR_RATIOS=cfg.ANCHOR_RATIOS,
                RPN_PRE_NMS_TOP_N=cfg.TEST.RPN_PRE_NMS_TOP_N, RPN_POST_NMS_TOP_N=Nb,
                RPN_NMS_THRESH=cfg.TEST.RPN_NMS_THRESH, POOLING_SIZE=7)
    r_o, s_o, pooled_o, fc7_o = OD.detector_forward(batch.im_data.cpu(), batch.im_info.cpu(), sd, ocfg)
    same = (rois.cpu() == r_o).all(-1).view(-1)
    e_fc7 = float((fc7.cpu()[same] - fc7_o[same]).abs().max() / fc7_o.abs().max())
    V = model.vis_ebd(fc7)
    W = model.word_ebd(batch.glove_feats)
    D, D_sim, L = model.DVSA(V, W, batch.entities_length)
    D_o, Ds_o, L_o = O.dvsa_forward(V.detach().cpu(), W.detach().cpu(), batch.entities_length, Na, Nb, Ne, 10.0, 4.13, 'train')
    e_loss = abs(float(L) - float(L_o)) / abs(float(L_o))
    ind_ok = bool((D.cpu() == D_o).all())
    # and one full training iteration (forward + backward + clip + Adam)
    loss, _, _, _ = train_step(model, opt, crit, batch, args)
    torch.cuda.synchronize()
    if verbose:
        print("smoke: %s | base_feat rel.err %.2e | identical rois %.1f%% | fc7 rel.err %.2e | loss %.5f vs oracle %.5f "
              "(rel %.1e) | D_ind exact: %s | train-step loss %.5f"
              % (_lib.version(), e_base, 100 * float(same.float().mean()), e_fc7, float(L), float(L_o), e_loss, ind_ok,
                 float(loss)))
    assert e_base < 1e-4, e_base
    assert float(same.float().mean()) > 0.9
    assert e_fc7 < 1e-4, e_fc7
    assert e_loss < 1e-4 and ind_ok
    assert torch.isfinite(loss).item()
    return float(loss)
