"""Import harness for the *reference* implementation (test infrastructure only).

This module exists ONLY in this build container: it loads ``/root/reference/model.py`` and the
``lib/model`` package unmodified, with stand-in modules for the third-party packages that are
not installed here (easydict, tensorboardX, torchtext, cv2, nltk, torchvision, torch.utils.ffi,
torch._six and the five compiled ``_ext`` CUDA extensions).  It is used by
``make_golden.py`` to generate the fixtures under ``tests/golden/*.npz``; nothing under
``nafae_amd/`` and no ``-m gpu`` test imports it, and the reference sources never travel to the
GPU box.

What is injected (and why it does not weaken the pin):
  * ``torchvision.models.vgg16``  -- only the *topology* (cfg "D") is third-party; arithmetic is
    torch ``conv2d/linear/max_pool2d``.  (reference call site: lib/model/faster_rcnn/vgg16_rpn.py:29)
  * ``proposal_layer.nms``         -- the reference has no CPU NMS (lib/model/nms/nms_wrapper.py:18);
    the harness plugs the oracle's C restatement so that the *glue* (sort, top-N, padding,
    roi_scores: proposal_layer.py:125-171) is the reference's own code.
  * ``RoIAlignFunction``           -- raises on CPU (lib/model/roi_align/functions/roi_align.py:29);
    the oracle's C restatement is plugged the same way.
  * ``Tensor.masked_fill_`` accepts the uint8 masks model.py:537,540 builds (torch 2.x needs bool).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("NAFAE_REFERENCE", "/root/reference")


class _EasyDict(dict):
    """Minimal easydict: attribute access, recursive dict wrapping, tuple->list coercion
    (old easydict behaviour the reference relies on for TEST.SCALES, config.py:168)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, (list, tuple)):
            v = [type(self)(x) if isinstance(x, dict) else x for x in v]
        elif isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = type(self)(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _vgg16_topology():
    """torchvision.models.vgg16() stand-in: configuration "D" (13 conv3x3 + 5 maxpool, classifier
    25088->4096->4096->1000).  Only the module list matters to vgg16_rpn.py:29-46."""
    import torch.nn as nn

    class VGG(nn.Module):
        def __init__(self):
            super().__init__()
            cfgD = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
            layers, cin = [], 3
            for v in cfgD:
                if v == 'M':
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                    cin = v
            self.features = nn.Sequential(*layers)
            self.classifier = nn.Sequential(
                nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
                nn.Linear(4096, 1000))
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                    nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0, 0.01)
                    nn.init.constant_(m.bias, 0)

    return VGG()


_loaded = None


def load_reference():
    """Returns the reference ``model.py`` module object (named ``nafae_ref_model``)."""
    global _loaded
    if _loaded is not None:
        return _loaded

    _stub("easydict", EasyDict=_EasyDict)
    _stub("tensorboardX", SummaryWriter=object)
    tt = _stub("torchtext")
    tt.vocab = _stub("torchtext.vocab")
    _stub("cv2")
    nl = _stub("nltk")
    nl.stem = _stub("nltk.stem", WordNetLemmatizer=object)
    nl.corpus = _stub("nltk.corpus", wordnet=object)
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models", vgg16=_vgg16_topology)
    _stub("torch.utils.ffi", _wrap_function=lambda *a, **k: None, create_extension=None)
    _stub("torch._six", int_classes=int, string_classes=str)
    _stub("_init_paths")
    for op in ("nms", "roi_align", "roi_pooling", "roi_crop"):
        base = "model.%s._ext" % op
        # populated after `model` package is importable; registered lazily below
    sys.path.insert(0, os.path.join(REF, "lib"))
    sys.path.insert(0, os.path.join(REF, "lib", "model"))

    # compiled extension packages: model.<op>._ext.<op> with the C entry points as attributes
    import model  # noqa: F401  (the reference's lib/model package)
    for op, names in (("nms", ["nms_cuda"]),
                      ("roi_align", ["roi_align_forward_cuda", "roi_align_backward_cuda"]),
                      ("roi_pooling", ["roi_pooling_forward", "roi_pooling_backward",
                                       "roi_pooling_forward_cuda", "roi_pooling_backward_cuda"]),
                      ("roi_crop", [])):
        pkg = _stub("model.%s._ext" % op)
        pkg.__path__ = []
        sub = _stub("model.%s._ext.%s" % (op, op))
        for n in names:
            setattr(sub, n, None)
        setattr(pkg, op, sub)
    # crop_resize (roi_crop) has a second ext name
    pkg = sys.modules["model.roi_crop._ext"]
    sub2 = _stub("model.roi_crop._ext.roi_crop")
    pkg.roi_crop = sub2
    _stub("model.roi_crop._ext.crop_resize")

    # PIL font file is absent in this image (net_utils.py:49)
    import PIL.ImageFont as ImageFont
    _orig_tt = ImageFont.truetype

    def _tt(*a, **k):
        try:
            return _orig_tt(*a, **k)
        except Exception:
            return ImageFont.load_default()
    ImageFont.truetype = _tt

    # yaml.load without Loader (config.py:374)
    import yaml
    _orig_load = yaml.load
    yaml.load = lambda f, Loader=None: _orig_load(f, Loader=Loader or yaml.FullLoader)

    # uint8 masks -> bool (model.py:537,540,551,575)
    _orig_mf = torch.Tensor.masked_fill_

    def _mf(self, mask, value):
        if mask.dtype == torch.uint8:
            mask = mask.bool()
        return _orig_mf(self, mask, value)
    torch.Tensor.masked_fill_ = _mf
    # init.xavier_normal was removed (transformer/SubLayers.py:34-36)
    import torch.nn.init as init
    if not hasattr(init, "xavier_normal"):
        init.xavier_normal = init.xavier_normal_

    spec = importlib.util.spec_from_file_location("nafae_ref_model", os.path.join(REF, "model.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nafae_ref_model"] = mod
    spec.loader.exec_module(mod)
    mod.device = torch.device("cpu")
    _loaded = mod
    return mod


def plug_native_ops(nms_fn, roi_align_fn):
    """Plug CPU callables where the reference calls its CUDA-only natives.
    nms_fn(dets[n,5] tensor, thresh) -> int tensor [n_keep,1]   (nms_wrapper.nms signature)
    roi_align_fn(features, rois, AH, AW, scale) -> tensor [N,C,AH,AW]
    """
    load_reference()
    import model.rpn.proposal_layer as pl
    import model.roi_align.modules.roi_align as ram
    pl.nms = nms_fn

    class _Fn(object):
        def __init__(self, ah, aw, scale):
            self.ah, self.aw, self.scale = int(ah), int(aw), float(scale)

        def __call__(self, features, rois):
            return roi_align_fn(features, rois, self.ah, self.aw, self.scale)
    ram.RoIAlignFunction = _Fn


def make_args(**over):
    """argparse.Namespace with the reference defaults (model.py:35-264) for the attributes the
    model constructors consume."""
    import argparse
    d = dict(n_head=8, word_ebd_dim=512, d_k=64, d_v=64, dropout_rate=0.1, n_position=100,
             sample_num=5, batch_size=8, batch_size_val=1, max_ent_len=13, Delta=1.0, vis_lam=1.0,
             vis_fc_dim=4096, glove_dim=200, class_agnostic=False)
    d.update(over)
    return argparse.Namespace(**d)
