"""Config surface of the reference kept verbatim: a module-global attribute-dict ``cfg`` whose detector keys are
read at construction AND call time, a yaml overlay (``cfg_from_file``) and a ``--set k v ...`` overlay
(``cfg_from_list``).

Mirrors lib/model/utils/config.py: defaults :19-302 (same key names and values), merge rules :337-367 (a yaml key
must already exist and its type must match), :370-376 (yaml loader), :379-399 (CLI list).  Two deliberate
differences, both noted in SURVEY.md section 5: a safe yaml loader is used (the reference's bare ``yaml.load(f)``
fails on PyYAML >= 6), and list <-> tuple are accepted for each other (``TEST.SCALES: [224]`` in cfgs/vgg16.yml
only merged in the reference because old ``easydict`` coerced the tuple default to a list).
"""
import os
from ast import literal_eval

import numpy as np


class AttrDict(dict):
    """Attribute-style dict (stand-in for easydict.EasyDict, which is not a dependency)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        elif isinstance(v, tuple):
            v = list(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _defaults():
    return AttrDict({
        "TRAIN": {
            "LEARNING_RATE": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "GAMMA": 0.1, "STEPSIZE": [30000],
            "DISPLAY": 10, "DOUBLE_BIAS": True, "TRUNCATED": False, "BIAS_DECAY": False, "USE_GT": False,
            "ASPECT_GROUPING": False, "SNAPSHOT_KEPT": 3, "SUMMARY_INTERVAL": 180, "SCALES": [600], "MAX_SIZE": 1000,
            "TRIM_HEIGHT": 600, "TRIM_WIDTH": 600, "IMS_PER_BATCH": 1, "BATCH_SIZE": 128, "FG_FRACTION": 0.25,
            "FG_THRESH": 0.5, "BG_THRESH_HI": 0.5, "BG_THRESH_LO": 0.1, "USE_FLIPPED": True, "BBOX_REG": True,
            "BBOX_THRESH": 0.5, "SNAPSHOT_ITERS": 5000, "SNAPSHOT_PREFIX": "res101_faster_rcnn",
            "BBOX_NORMALIZE_TARGETS": True, "BBOX_INSIDE_WEIGHTS": [1.0, 1.0, 1.0, 1.0],
            "BBOX_NORMALIZE_TARGETS_PRECOMPUTED": True, "BBOX_NORMALIZE_MEANS": [0.0, 0.0, 0.0, 0.0],
            "BBOX_NORMALIZE_STDS": [0.1, 0.1, 0.2, 0.2], "PROPOSAL_METHOD": "gt", "HAS_RPN": True,
            "RPN_POSITIVE_OVERLAP": 0.7, "RPN_NEGATIVE_OVERLAP": 0.3, "RPN_CLOBBER_POSITIVES": False,
            "RPN_FG_FRACTION": 0.5, "RPN_BATCHSIZE": 256, "RPN_NMS_THRESH": 0.7, "RPN_PRE_NMS_TOP_N": 12000,
            "RPN_POST_NMS_TOP_N": 2000, "RPN_MIN_SIZE": 8, "RPN_BBOX_INSIDE_WEIGHTS": [1.0, 1.0, 1.0, 1.0],
            "RPN_POSITIVE_WEIGHT": -1.0, "USE_ALL_GT": True, "BN_TRAIN": False,
        },
        "TEST": {
            "SCALES": [600], "MAX_SIZE": 1000, "NMS": 0.3, "SVM": False, "BBOX_REG": True, "HAS_RPN": False,
            "PROPOSAL_METHOD": "gt", "RPN_NMS_THRESH": 0.7, "RPN_PRE_NMS_TOP_N": 6000, "RPN_POST_NMS_TOP_N": 300,
            "RPN_MIN_SIZE": 16, "MODE": "nms", "RPN_TOP_N": 5000,
        },
        "RESNET": {"MAX_POOL": False, "FIXED_BLOCKS": 1},
        "MOBILENET": {"REGU_DEPTH": False, "FIXED_LAYERS": 5, "WEIGHT_DECAY": 4e-05, "DEPTH_MULTIPLIER": 1.0},
        "DEDUP_BOXES": 0.0625,
        "PIXEL_MEANS": np.array([[[102.9801, 115.9465, 122.7717]]]),
        "RNG_SEED": 3, "EPS": 1e-14, "ROOT_DIR": _ROOT, "DATA_DIR": os.path.join(_ROOT, "data"), "MATLAB": "matlab",
        "EXP_DIR": "default", "USE_GPU_NMS": True, "GPU_ID": 0, "POOLING_MODE": "crop", "POOLING_SIZE": 7,
        "MAX_NUM_GT_BOXES": 20, "ANCHOR_SCALES": [8, 16, 32], "ANCHOR_RATIOS": [0.5, 1, 2], "FEAT_STRIDE": [16],
        "CUDA": False, "CROP_RESIZE_WITH_MAX_POOL": True,
    })


cfg = _defaults()


def reset_cfg():
    """Restore the defaults in place (the global object identity is kept, as modules hold references to it)."""
    d = _defaults()
    cfg.clear()
    for k, v in d.items():
        cfg[k] = v
    return cfg


def _same_type(old, new):
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return True
    return type(old) is type(new)


def _merge_a_into_b(a, b):
    """config.py:337-367: keys of ``a`` must exist in ``b`` with a matching type."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old = b[k]
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        if not _same_type(old, v):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(v), k))
        if isinstance(v, AttrDict):
            try:
                _merge_a_into_b(v, old)
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a yaml config file and merge it into the global options (config.py:370-376)."""
    import yaml
    with open(filename, 'r') as f:
        y = yaml.safe_load(f)
    _merge_a_into_b(AttrDict(y), cfg)


def cfg_from_list(cfg_list):
    """Set config keys from ``--set K V K V ...`` (config.py:379-399)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = cfg
        for subkey in key_list[:-1]:
            assert subkey in d
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        assert _same_type(d[subkey], value), 'type {} does not match original type {}'.format(type(value),
                                                                                             type(d[subkey]))
        d[subkey] = value
