"""Deterministic synthetic weights and inputs (no datasets / checkpoints exist offline).

Shapes and distributions follow SURVEY.md section 8(d): frames are uint8 U[0,255) minus 127.5 in BGR NCHW
(reference preprocessing: lib/datasets/youcook2.py:212-214), VGG convs are Kaiming-normal, fc and RPN /
head layers N(0, 0.01) (lib/model/faster_rcnn/rpn.py:89-105), GloVe rows N(0, 0.4^2) with zero rows for
padded query slots (model.py:730-747), entity counts drawn from the train-split histogram.

Every tensor has its own generator seeded from (seed, name) so that a test can regenerate exactly the
weights a golden fixture was produced with, on any machine running the same torch build.
"""
import zlib

import numpy as np
import torch

VGG_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]
VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]

# P(len = k) on the YouCookII train split, k = 0..13 (mean 2.07, P(0) = 0.108)
_LEN_HIST = np.array([0.108, 0.262, 0.289, 0.186, 0.088, 0.038, 0.016, 0.007, 0.003, 0.001, 0.001, 0.0005,
                      0.0003, 0.0002])


def _gen(seed, name):
    g = torch.Generator(device='cpu')
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def randn(seed, name, shape, std=1.0):
    return torch.randn(shape, generator=_gen(seed, name), dtype=torch.float32) * std


def detector_state(seed=1234, n_classes=2501, anchors=12, heads=True, bias_std=0.01):
    """State dict of the frozen detector, keys as in the reference checkpoint minus the 'fasterRCNN.'
    prefix.  ``heads=False`` skips the two unused 2501-class heads (62 MB)."""
    sd = {}
    cin = 3
    li = 0
    for v in VGG_CFG_D:
        if v == 'M':
            continue
        k = 'RCNN_base.%d' % VGG_CONV_IDX[li]
        # Kaiming-normal; the first layer is scaled by 1/64 so that +-127.5 pixel inputs give O(1)
        # activations (and hence un-saturated RPN scores / sane box deltas) with random weights
        std = float(np.sqrt(2.0 / (cin * 9))) * (1.0 / 64 if cin == 3 else 1.0)
        sd[k + '.weight'] = randn(seed, k + '.w', (v, cin, 3, 3), std=std)
        sd[k + '.bias'] = randn(seed, k + '.b', (v,), std=bias_std)
        cin = v
        li += 1
    for name, shape in (('RCNN_rpn.RPN_Conv', (512, 512, 3, 3)),
                        ('RCNN_rpn.RPN_cls_score', (2 * anchors, 512, 1, 1)),
                        ('RCNN_rpn.RPN_bbox_pred', (4 * anchors, 512, 1, 1))):
        sd[name + '.weight'] = randn(seed, name + '.w', shape, std=0.01)
        sd[name + '.bias'] = randn(seed, name + '.b', (shape[0],), std=bias_std)
    sd['RCNN_top.0.weight'] = randn(seed, 'fc6.w', (4096, 25088), std=0.01)
    sd['RCNN_top.0.bias'] = randn(seed, 'fc6.b', (4096,), std=bias_std)
    sd['RCNN_top.3.weight'] = randn(seed, 'fc7.w', (4096, 4096), std=0.3)   # fc7 ~ O(30): VisEbd divides by 100
    sd['RCNN_top.3.bias'] = randn(seed, 'fc7.b', (4096,), std=bias_std)
    if heads:
        sd['RCNN_cls_score.weight'] = randn(seed, 'cls.w', (n_classes, 4096), std=0.01)
        sd['RCNN_cls_score.bias'] = torch.zeros(n_classes)
        sd['RCNN_bbox_pred.weight'] = randn(seed, 'bbox.w', (4 * n_classes, 4096), std=0.001)
        sd['RCNN_bbox_pred.bias'] = torch.zeros(4 * n_classes)
    return sd


def frames(F, H=224, W=224, seed=1234):
    """[F,3,H,W] float32: uint8 U[0,255) - 127.5 (BGR NCHW), and im_info [F,3] = (h, w, 1)."""
    g = _gen(seed, 'frames')
    im = torch.randint(0, 255, (F, 3, H, W), generator=g, dtype=torch.int32).float() - 127.5
    im_info = torch.tensor([[H, W, 1.0]] * F, dtype=torch.float32)
    return im, im_info


def entity_lengths(Na, Ne, seed=1234):
    """Entity count per segment from the train-split histogram, clipped to [0, Ne], at least one > 0."""
    rs = np.random.RandomState((seed * 7919 + Na * 31 + Ne) & 0x7FFFFFFF)
    p = _LEN_HIST / _LEN_HIST.sum()
    lens = np.minimum(rs.choice(len(p), size=Na, p=p), Ne).tolist()
    if max(lens) == 0:
        lens[0] = min(2, Ne)
    return [int(x) for x in lens]


def glove(Na, Ne, lens, dim=200, seed=1234):
    """[Na*Ne, dim]: N(0, 0.4^2) rows for real entities, zero rows for padded slots."""
    g = randn(seed, 'glove', (Na, Ne, dim), std=0.4)
    for a, l in enumerate(lens):
        g[a, l:] = 0
    return g.view(Na * Ne, dim)


def embeddings(R, Q, D=512, seed=1234):
    """tanh(N(0,1)) stand-ins for V [R,D], W [Q,D] (sim+loss-only runs)."""
    return torch.tanh(randn(seed, 'V', (R, D))), torch.tanh(randn(seed, 'W', (Q, D)))
