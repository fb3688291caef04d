"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement in plain PyTorch fp32 (+ numpy for anchors, + oracle/native.c for NMS / ROI-Align)
of the reference's frozen detector forward:

  generate_anchors      <- lib/model/rpn/generate_anchors.py:45-105
  all_anchors           <- lib/model/rpn/proposal_layer.py:79-93
  bbox_transform_inv    <- lib/model/rpn/bbox_transform.py:77-103
  clip_boxes            <- lib/model/rpn/bbox_transform.py:125-133
  rpn_head              <- lib/model/rpn/rpn.py:58-79
  proposal_layer        <- lib/model/rpn/proposal_layer.py:49-171
  vgg16_features        <- lib/model/faster_rcnn/vgg16_rpn.py:28-46 (torchvision cfg "D" minus pool5)
  head_to_tail          <- lib/model/faster_rcnn/vgg16_rpn.py:56-61
  detector_forward      <- lib/model/faster_rcnn/rpn.py:39-87

Pinned by tests/golden/detector_*.npz, generated from the imported reference with the oracle's C
NMS / ROI-Align plugged in where the reference is CUDA-only (tests/golden/ref_harness.py).
The only known-answer vector the reference itself carries is the anchor table
(generate_anchors.py:12-37), checked in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import native

VGG_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
# indices of the conv modules inside RCNN_base (nn.Sequential of conv,relu,...,pool): state-dict keys
VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]


# ----------------------------------------------------------------------------- anchors
def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(ws, hs, xc, yc):
    ws = np.asarray(ws, dtype=np.float64)[:, None]
    hs = np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """generate_anchors.py:45-105 (float64): ratio enumeration (rounded w,h) then scale enumeration."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    base = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, xc, yc = _whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mk(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, xc, yc = _whctrs(ratio_anchors[i])
        out.append(_mk(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def all_anchors(feat_h, feat_w, feat_stride, scales, ratios):
    """proposal_layer.py:79-93: [K*A,4] float32, ordered (K = h*W + w, A) row-major."""
    base = torch.from_numpy(generate_anchors(scales=np.array(scales), ratios=np.array(ratios))).float()
    sx = np.arange(0, feat_w) * feat_stride
    sy = np.arange(0, feat_h) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = torch.from_numpy(np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()).contiguous().float()
    A, K = base.shape[0], shifts.shape[0]
    return (base.view(1, A, 4) + shifts.view(K, 1, 4)).view(K * A, 4)


# ----------------------------------------------------------------------------- box transforms
def bbox_transform_inv(boxes, deltas):
    """bbox_transform.py:77-103; boxes/deltas [B,N,4]."""
    widths = boxes[:, :, 2] - boxes[:, :, 0] + 1.0
    heights = boxes[:, :, 3] - boxes[:, :, 1] + 1.0
    ctr_x = boxes[:, :, 0] + 0.5 * widths
    ctr_y = boxes[:, :, 1] + 0.5 * heights
    dx, dy, dw, dh = deltas[:, :, 0], deltas[:, :, 1], deltas[:, :, 2], deltas[:, :, 3]
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = torch.exp(dw) * widths
    ph = torch.exp(dh) * heights
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), 2)


def clip_boxes(boxes, im_info):
    """bbox_transform.py:125-133; im_info[i] = (h, w, scale)."""
    boxes = boxes.clone()
    for i in range(boxes.shape[0]):
        boxes[i, :, 0].clamp_(0, float(im_info[i, 1]) - 1)
        boxes[i, :, 1].clamp_(0, float(im_info[i, 0]) - 1)
        boxes[i, :, 2].clamp_(0, float(im_info[i, 1]) - 1)
        boxes[i, :, 3].clamp_(0, float(im_info[i, 0]) - 1)
    return boxes


# ----------------------------------------------------------------------------- RPN
def rpn_head(base_feat, p):
    """rpn/rpn.py:63-72.  p: dict with RPN_Conv/RPN_cls_score/RPN_bbox_pred .weight/.bias.
    Returns fg/bg probs [F,2A,H,W] and deltas [F,4A,H,W]."""
    x = F.relu(F.conv2d(base_feat, p['RPN_Conv.weight'], p['RPN_Conv.bias'], padding=1))
    cls = F.conv2d(x, p['RPN_cls_score.weight'], p['RPN_cls_score.bias'])
    B, C2, H, W = cls.shape
    r = cls.view(B, 2, (C2 * H) // 2, W)                 # rpn.py:47-56 reshape(x, 2)
    prob = F.softmax(r, dim=1).view(B, C2, H, W)         # implicit dim for 4-D input is 1
    deltas = F.conv2d(x, p['RPN_bbox_pred.weight'], p['RPN_bbox_pred.bias'])
    return prob, deltas


def decode_proposals(prob, deltas, im_info, feat_stride, scales, ratios):
    """proposal_layer.py:67-109: fg scores and clipped boxes in (H,W,A) order.
    Returns scores [F,K*A], proposals [F,K*A,4]."""
    A = len(scales) * len(ratios)
    B, _, H, W = deltas.shape
    scores = prob[:, A:, :, :]
    anchors = all_anchors(H, W, feat_stride, scales, ratios).view(1, -1, 4).expand(B, -1, 4)
    d = deltas.permute(0, 2, 3, 1).contiguous().view(B, -1, 4)
    s = scores.permute(0, 2, 3, 1).contiguous().view(B, -1)
    props = clip_boxes(bbox_transform_inv(anchors, d), im_info)
    return s, props


def sort_desc(scores):
    """proposal_layer.py:125 ``torch.sort(scores, 1, True)``.  The oracle pins ties to ascending
    original index (a stable descending sort), which is what torch's CPU sort yields."""
    return torch.sort(scores, dim=1, descending=True, stable=True)[1]


def select_proposals(scores, props, order, pre_nms_topN, post_nms_topN, nms_thresh):
    """proposal_layer.py:127-165: per-frame NMS + top-N + zero padding.
    Returns rois [F,Nb,5], roi_scores [F,Nb], n_keep [F] (number of valid rows, <= Nb)."""
    B = scores.shape[0]
    rois = scores.new_zeros(B, post_nms_topN, 5)
    roi_scores = scores.new_zeros(B, post_nms_topN)
    n_keep = []
    for i in range(B):
        o = order[i]
        if 0 < pre_nms_topN < scores.numel():            # :140 (guard is on the BATCH numel)
            o = o[:pre_nms_topN]
        p = props[i][o, :]
        s = scores[i][o].view(-1, 1)
        keep = native.nms(torch.cat((p, s), 1).numpy(), nms_thresh)
        keep = torch.from_numpy(keep.astype(np.int64))
        if post_nms_topN > 0:
            keep = keep[:post_nms_topN]
        n = keep.numel()
        rois[i, :, 0] = i
        rois[i, :n, 1:] = p[keep, :]
        roi_scores[i, :n] = s[keep, 0]
        n_keep.append(n)
    return rois, roi_scores, n_keep


# ----------------------------------------------------------------------------- VGG16
def vgg16_features(im_data, p, prefix='RCNN_base.'):
    """vgg16_rpn.py:38: torchvision VGG16 "D" features without the last max-pool."""
    x = im_data
    li = 0
    for v in VGG_CFG_D:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
        else:
            k = prefix + str(VGG_CONV_IDX[li])
            x = F.relu(F.conv2d(x, p[k + '.weight'], p[k + '.bias'], padding=1))
            li += 1
    return x


def head_to_tail(pooled, p, prefix='RCNN_top.'):
    """vgg16_rpn.py:56-61 (Dropout = identity, detector always in eval)."""
    x = pooled.reshape(pooled.shape[0], -1)
    x = F.relu(F.linear(x, p[prefix + '0.weight'], p[prefix + '0.bias']))
    x = F.relu(F.linear(x, p[prefix + '3.weight'], p[prefix + '3.bias']))
    return x


def roi_align_avg(base_feat, rois2d, pooled=7, scale=1.0 / 16.0):
    out = native.roi_align_avg(base_feat.detach().numpy(), rois2d.detach().numpy(), pooled, scale)
    return torch.from_numpy(out)


def detector_forward(im_data, im_info, p, cfg):
    """faster_rcnn/rpn.py:39-87 -> (rois, roi_scores, pooled_feat, fc7).  ``p`` is the fasterRCNN
    state dict (keys without the 'fasterRCNN.' prefix); cfg carries TEST.* / ANCHOR_* / FEAT_STRIDE."""
    with torch.no_grad():
        base = vgg16_features(im_data, p)
        rp = {k[len('RCNN_rpn.'):]: v for k, v in p.items() if k.startswith('RCNN_rpn.')}
        prob, deltas = rpn_head(base, rp)
        s, props = decode_proposals(prob, deltas, im_info, cfg['FEAT_STRIDE'], cfg['ANCHOR_SCALES'],
                                    cfg['ANCHOR_RATIOS'])
        order = sort_desc(s)
        rois, roi_scores, _ = select_proposals(s, props, order, cfg['RPN_PRE_NMS_TOP_N'],
                                               cfg['RPN_POST_NMS_TOP_N'], cfg['RPN_NMS_THRESH'])
        pooled = roi_align_avg(base, rois.view(-1, 5), cfg.get('POOLING_SIZE', 7), 1.0 / 16.0)
        fc7 = head_to_tail(pooled, p)
    return rois, roi_scores, pooled, fc7
