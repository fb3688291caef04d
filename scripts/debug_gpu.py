import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
from nafae_amd.config import cfg, cfg_from_file, reset_cfg
from nafae_amd import synthetic as syn, ops
from oracle import detector as OD
reset_cfg(); cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
# ---- embed
from nafae_amd.model import DVSA, VisEbd, WordEbd, default_args
g = np.load(os.path.join(G, "embed.npz"))
Na, Ns, Nb, Ne, D, FC, Gd = [int(x) for x in g["shape"]]
cfg.TEST.RPN_POST_NMS_TOP_N = Nb
args = default_args(batch_size=Na, batch_size_val=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13, word_ebd_dim=D, vis_fc_dim=FC, glove_dim=Gd)
ve, we, dv = VisEbd(args).cuda(), WordEbd(args).cuda(), DVSA(args, cfg).cuda()
with torch.no_grad():
    for p, k in ((ve.fc1.weight, "ve_w"), (ve.fc1.bias, "ve_b"), (we.fc1.weight, "we_w"), (we.fc1.bias, "we_b"), (we.bn.weight, "bn_w"), (we.bn.bias, "bn_b")):
        p.copy_(torch.from_numpy(g[k]))
fc7, glove = torch.from_numpy(g["fc7"]).cuda(), torch.from_numpy(g["glove"]).cuda()
lens = g["lens"].tolist()
ve.train(); we.train(); dv.init_train()
V, W = ve(fc7), we(glove)
print("V", relerr(V.detach().cpu(), g["V_train"]), "W", relerr(W.detach().cpu(), g["W_train"]))
Di, Ds, L = dv(V, W, lens)
loss = torch.nn.L1Loss()(L, torch.zeros_like(L)); loss.backward()
print("loss", float(L), float(g["loss_train"]), "Dind eq", np.array_equal(Di.cpu().numpy(), g["D_ind_train"]))
for p, k in ((ve.fc1.weight, "g_ve_w"), (ve.fc1.bias, "g_ve_b"), (we.fc1.weight, "g_we_w"), (we.fc1.bias, "g_we_b"), (we.bn.weight, "g_bn_w"), (we.bn.bias, "g_bn_b")):
    print(k, relerr(p.grad.cpu(), g[k]), float(np.abs(g[k]).max()))
print("rm", relerr(we.bn.running_mean.cpu(), g["run_mean"]), "rv", relerr(we.bn.running_var.cpu(), g["run_var"]))
ve.eval(); we.eval(); dv.init_eval()
with torch.no_grad():
    V, W = ve(fc7), we(glove); Di, Ds, L = dv(V, W, lens)
print("eval W", relerr(W.cpu(), g["W_eval"]), "Dind", np.array_equal(Di.cpu().numpy(), g["D_ind_eval"]), "Dsim", relerr(Ds.cpu(), g["D_sim_eval"]), "loss", float(L), float(g["loss_eval"]))
# ---- detector
from nafae_amd.detector import vgg16
g = np.load(os.path.join(G, "detector.npz"))
cfg.TEST.RPN_POST_NMS_TOP_N = int(g["post_nms_topN"])
fr = vgg16(np.array([''] * 2501)); fr.create_architecture()
print(fr.load_state_dict(syn.detector_state(seed=77, heads=False), strict=False))
fr = fr.eval().cuda()
im, im_info = syn.frames(2, 64, 48, seed=77)
base = fr.base_features(im.cuda())
print("base", relerr(base.permute(0, 3, 1, 2).cpu(), g["base_feat"]))
sd = syn.detector_state(seed=77, heads=False)
rp = {k[len('RCNN_rpn.'):]: v for k, v in sd.items() if k.startswith('RCNN_rpn.')}
basec = base.permute(0, 3, 1, 2).cpu().contiguous()
import torch.nn.functional as F
x_o = F.relu(F.conv2d(basec, rp['RPN_Conv.weight'], rp['RPN_Conv.bias'], padding=1))
P = fr._pack()
Fr, h, w, _ = base.shape
x = ops.conv3x3_relu(base, P['rpn_w'], P['rpn_b'], relu=True)
print("rpn conv", relerr(x.permute(0, 3, 1, 2).cpu(), x_o))
head = ops.gemm_nt(x.view(Fr * h * w, 512), P['head_w'], P['head_b'])
cls_o = F.conv2d(x_o, rp['RPN_cls_score.weight'], rp['RPN_cls_score.bias'])
del_o = F.conv2d(x_o, rp['RPN_bbox_pred.weight'], rp['RPN_bbox_pred.bias'])
head_o = torch.cat([cls_o, del_o], 1).permute(0, 2, 3, 1).reshape(Fr * h * w, -1)
print("head", relerr(head.cpu(), head_o), head.shape)
prob, deltas = OD.rpn_head(basec, rp)
s_o, p_o = OD.decode_proposals(prob, deltas, im_info, 16, [4, 8, 16, 32], [0.5, 1, 2])
A = 12
s, p = ops.rpn_decode(head, P['anchors'], im_info.cuda(), Fr, h, w, A, 16)
print("scores", relerr(s.cpu(), s_o), "boxes", relerr(p.cpu(), p_o), "anchors", P['anchors'][:2].cpu().tolist())
order = ops.sort_desc(s)
order_o = OD.sort_desc(s_o)
print("order same frac", float((order.cpu().long() == order_o).float().mean()))
rois, rs, nk = ops.proposals(p, s, order, s.shape[1], 0.7, 8)
r_o, rs_o, nk_o = OD.select_proposals(s_o, p_o, order_o, 6000, 8, 0.7)
print("rois hip\n", rois.cpu()[0], "\noracle\n", r_o[0], "\ngolden\n", g["rois"][0])
r2, rs2, _, _ = fr(im.cuda(), im_info.cuda(), None, None)
print("forward rois\n", r2.cpu()[0], fr.n_keep)
