"""Checkpoint I/O in the reference's formats.

  * detector init:  torch.load('models/vgg16/pretrain/faster_rcnn_gnome.pth')['model'] -> GroundModel.fasterRCNN
                    (model.py:1056-1064); an optional 'pooling_mode' key overrides cfg.POOLING_MODE.
  * grounding ckpt: {'session', 'epoch', 'model': GroundModel.state_dict(), 'optimizer', 'pooling_mode'} saved as
                    output/models/<net>/<dataset>/vis_ground_{session}_{epoch}_{checkbatch}.pth (model.py:1115-1126,
                    net_utils.py:232-233); --resume reloads model + epoch but NOT the optimiser state (:1039-1046).
The state-dict keys and shapes of nafae_amd.model.GroundModel equal the reference's (tests/test_model_cpu.py), so files
written by either side load in the other.  Kernel-layout copies of the weights are rebuilt lazily after a load.
"""
import os

import torch

from .config import cfg


def checkpoint_name(save_dir, net, dataset, session, epoch, checkbatch):
    return os.path.join(save_dir, net, dataset, 'vis_ground_{}_{}_{}.pth'.format(session, epoch, checkbatch))


def save_checkpoint(state, filename):
    """net_utils.py:232-233."""
    d = os.path.dirname(filename)
    if d and not os.path.exists(d):
        os.makedirs(d)
    torch.save(state, filename)


def save_ground_checkpoint(ground_model, optimizer, session, epoch, filename):
    opt_state = optimizer.state_dict() if hasattr(optimizer, 'state_dict') else {}
    save_checkpoint({'session': session, 'epoch': epoch, 'model': ground_model.state_dict(), 'optimizer': opt_state,
                     'pooling_mode': cfg.POOLING_MODE}, filename)


def load_detector_checkpoint(ground_model, filename, map_location='cpu'):
    """model.py:1056-1064: detector-only initialisation from the Visual-Genome Faster-RCNN checkpoint."""
    ck = torch.load(filename, map_location=map_location)
    ground_model.fasterRCNN.load_state_dict(ck['model'])
    if 'pooling_mode' in ck:
        cfg.POOLING_MODE = ck['pooling_mode']
    return ck


def load_ground_checkpoint(ground_model, filename, resume=False, map_location='cpu'):
    """model.py:1039-1054.  Returns the start epoch (epoch + 1 when resuming training, epoch for val/test)."""
    ck = torch.load(filename, map_location=map_location)
    ground_model.load_state_dict(ck['model'])
    if 'pooling_mode' in ck:
        cfg.POOLING_MODE = ck['pooling_mode']
    return ck['epoch'] + 1 if resume else ck['epoch']


def adjust_learning_rate(optimizer, base_lr, epoch, drop_rate, step):
    """model.py:1084-1088: lr = base_lr * drop_rate ** (epoch // step) on every param group."""
    lr = base_lr * (drop_rate ** (epoch // step))
    for g in optimizer.param_groups:
        g['lr'] = lr
    return lr
