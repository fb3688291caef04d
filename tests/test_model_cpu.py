"""CPU-only checks of the host-side boundary: config surface, CLI surface, state-dict keys, anchors."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_cfg_from_file_and_list():
    from nafae_amd.config import cfg, cfg_from_file, cfg_from_list, reset_cfg
    reset_cfg()
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 300 and cfg.POOLING_MODE == 'crop' and cfg.ANCHOR_SCALES == [8, 16, 32]
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    # SURVEY.md section 3.3 [probed on the reference]: (20, [4,8,16,32], 'align')
    assert (cfg.TEST.RPN_POST_NMS_TOP_N, cfg.ANCHOR_SCALES, cfg.POOLING_MODE) == (20, [4, 8, 16, 32], 'align')
    assert cfg.TEST.SCALES == [224] and cfg.CROP_RESIZE_WITH_MAX_POOL is False and cfg.TRAIN.BATCH_SIZE == 256
    cfg_from_list(['TEST.RPN_POST_NMS_TOP_N', '128', 'POOLING_MODE', 'align'])
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 128
    with pytest.raises(AssertionError):
        cfg_from_list(['TEST.RPN_POST_NMS_TOP_N', '1.5'])          # type must match (config.py:395-398)
    with pytest.raises(AssertionError):
        cfg_from_list(['TEST.NO_SUCH_KEY', '1'])
    from nafae_amd.config import _merge_a_into_b, AttrDict
    with pytest.raises(KeyError):
        _merge_a_into_b(AttrDict({'NOT_A_KEY': 1}), cfg)            # config.py:346-347
    with pytest.raises(ValueError):
        _merge_a_into_b(AttrDict({'POOLING_SIZE': 'seven'}), cfg)   # config.py:350-357
    reset_cfg()


def test_parse_args_surface():
    from nafae_amd.model import parse_args
    a = parse_args([])
    # defaults of model.py:35-264
    assert (a.batch_size, a.sample_num, a.max_ent_len, a.word_ebd_dim, a.glove_dim, a.vis_fc_dim) == (8, 5, 13, 512, 200, 4096)
    assert (a.Delta, a.vis_lam, a.dropout_rate, a.clip, a.lr, a.weight_decay) == (1, 1, 0.1, 100, 0.001, 0.00001)
    assert a.cfg_file == 'cfgs/vgg16.yml' and a.n_head == 8 and a.n_position == 100 and a.batch_size_val == 1
    # the reference's train_model.sh command line parses unchanged
    a = parse_args("--cuda --phase train --checksession 0 --checkepoch 0 --checkbatch 1290 --shuffle_train --fix_seg_len "
                   "--Delta 10 --vis_lam 4.13 --workers 4 --epoch 30 --train_vis_freq 10000 --val_vis_freq 10000 "
                   "--statement train".split())
    assert a.cuda and a.phase == 'train' and a.Delta == 10 and a.vis_lam == 4.13 and a.shuffle_train and a.fix_seg_len
    a = parse_args("--phase test --bs 4 --set TEST.RPN_POST_NMS_TOP_N 128".split())
    assert a.set_cfgs == ['TEST.RPN_POST_NMS_TOP_N', '128'] and a.batch_size == 4


def test_state_dict_keys_match_reference():
    from nafae_amd.config import cfg, cfg_from_file, reset_cfg
    from nafae_amd.model import GroundModel, default_args
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    m = GroundModel(default_args(), cfg)
    ref_keys = np.load(os.path.join(G, "detector.npz"))["state_keys"].tolist()
    assert sorted(m.state_dict().keys()) == ref_keys and len(ref_keys) == 59
    sd = m.state_dict()
    assert tuple(sd['fasterRCNN.RCNN_top.0.weight'].shape) == (4096, 25088)
    assert tuple(sd['fasterRCNN.RCNN_rpn.RPN_cls_score.weight'].shape) == (24, 512, 1, 1)
    assert tuple(sd['fasterRCNN.RCNN_bbox_pred.weight'].shape) == (10004, 4096)
    assert tuple(sd['DVSA.slf_attn.w_qs'].shape) == (8, 512, 64) and tuple(sd['DVSA.ffn.weight'].shape) == (5, 1024)
    assert tuple(sd['DVSA.position_enc.weight'].shape) == (100, 512)
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert frozen == ['fasterRCNN.RCNN_base.%d.%s' % (i, k) for i in (0, 2, 5, 7) for k in ('weight', 'bias')]
    assert not m.fasterRCNN.training
    reset_cfg()


def test_anchors_match_reference_known_answer():
    from nafae_amd.detector import generate_anchors
    g = np.load(os.path.join(G, "anchors.npz"))
    assert np.array_equal(generate_anchors(), g["default"])
    assert np.array_equal(generate_anchors(scales=(4, 8, 16, 32)), g["vgg16_yml"])


def test_postprocess_matches_reference():
    from nafae_amd.model import postprocess
    g = np.load(os.path.join(G, "dvsa_ragged.npz"))
    Na, Ns, Nb, Ne, D = [int(x) for x in g["shape"]]
    Dp, Sp = postprocess(g["D_ind_eval"], g["D_sim_eval"], Na, Ns, Nb, Ne)
    assert np.array_equal(Dp, g["post_D"]) and np.array_equal(Sp, g["post_sim"])
