import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from nafae_amd import synthetic as syn
from nafae_amd.config import cfg, cfg_from_file, reset_cfg
from nafae_amd.model import default_args
from nafae_amd.train import make_batch, setup_training
from oracle import detector as OD
reset_cfg(); cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
Na, Ns, Ne, Nb = 2, 2, 8, 32
cfg.TEST.RPN_POST_NMS_TOP_N = Nb
args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
model, opt, crit, _ = setup_training(args, device='cuda:0', seed=1234)
batch = make_batch(Na, Ns, Ne, seed=1234, device='cuda:0', lens=[3, 5])
fr = model.fasterRCNN
print("anchor scales", fr.RCNN_rpn.anchor_scales, cfg.ANCHOR_SCALES, "training", fr.training)
sd = syn.detector_state(seed=1234, heads=False)
k='RCNN_rpn.RPN_Conv.weight'
print("w equal", torch.equal(fr.state_dict()[k].cpu(), sd[k]), torch.equal(fr.state_dict()['RCNN_base.0.weight'].cpu(), sd['RCNN_base.0.weight']))
rois, roi_scores, pooled, fc7 = fr(batch.im_data, batch.im_info, batch.gt_boxes, batch.num_boxes)
ocfg = dict(FEAT_STRIDE=16, ANCHOR_SCALES=cfg.ANCHOR_SCALES, ANCHOR_RATIOS=cfg.ANCHOR_RATIOS, RPN_PRE_NMS_TOP_N=cfg.TEST.RPN_PRE_NMS_TOP_N, RPN_POST_NMS_TOP_N=Nb, RPN_NMS_THRESH=cfg.TEST.RPN_NMS_THRESH, POOLING_SIZE=7)
r_o, s_o, pooled_o, fc7_o = OD.detector_forward(batch.im_data.cpu(), batch.im_info.cpu(), sd, ocfg)
print(rois[0,:4].cpu(), r_o[0,:4], roi_scores[0,:4].cpu(), s_o[0,:4])
print(batch.im_info.cpu())
