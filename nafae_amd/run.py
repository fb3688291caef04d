"""`python -m nafae_amd.run --cuda --phase train ...` -- the reference's entry point (model.py:994-1141: `main()`, what
train_model.sh / test_model.sh / eval_model.sh call) over this package's pieces: parse_args (same flags), cfg_from_file /
cfg_from_list, GroundModel + the three checkpoint branches (resume / val-test / detector init), Adam + L1Loss, the epoch
loop with lr decay, validation every `eval_freq` epochs, best-accuracy checkpoint, `.optm/<name>.best`.

    python -m nafae_amd.run --cuda --phase train --checksession 0 --checkepoch 0 --checkbatch 1290 --shuffle_train \\
        --fix_seg_len --Delta 10 --vis_lam 4.13 --workers 4 --epoch 30 --train_vis_freq 10000 --val_vis_freq 10000 \\
        --statement train                                             (= train_model.sh's command line, unchanged)

What is NOT here is the reference's data side (YouCookII frames, annotation json, the GloVe table, faster_rcnn_gnome.pth):
there is no network and no dataset in this environment.  When `<root>/<dataset>` does not exist the run switches -- loudly -- to
SYNTHETIC loader tuples of the same structure (train.combine_batches_synthetic: the 8-tuple of youcook2.py:254-308), a seeded
synthetic GloVe table over their vocabulary and, if `<load_dir>/faster_rcnn_gnome.pth` is missing too, the seeded synthetic
detector weights of nafae_amd.synthetic.  With the dataset present, pass `loaders=` / `glove=` (any iterables of loader tuples
and any object with torchtext's GloVe interface): the dataset classes themselves are out of scope.

The HIP path has no CPU fallback: without --cuda and a visible GPU the run stops before building the model.  Under
`torch.distributed.run` (WORLD_SIZE > 1) every rank trains on its own loader tuples and the flat gradient buffer is
all-reduced over RCCL (parallel.GradAllReducer) -- the reference parses --mGPUs and ignores it (model.py:91-99).
"""
import os
import sys

import numpy as np
import torch

from .config import cfg, cfg_from_file, cfg_from_list
from .model import parse_args

SYN_VOCAB = ('bowl', 'egg', 'pan', 'oil', 'salt', 'water')


class SyntheticGloVe:
    """torchtext.vocab.GloVe's interface (`.stoi`, `.itos`, `.vectors`) over a small seeded table (model.py:1019-1020)."""

    def __init__(self, words=SYN_VOCAB, dim=200, seed=1234):
        g = torch.Generator().manual_seed(seed)
        self.itos = list(words)
        self.stoi = {w: i for i, w in enumerate(self.itos)}
        self.vectors = torch.randn(len(self.itos), dim, generator=g) * 0.4


def plan(argv=None):
    """Everything main() decides before it touches the GPU: parsed args, merged cfg, directories, checkpoint to load and why,
    whether the data side is synthetic.  Pure host logic (the CPU test of the flag plumbing calls this)."""
    args = parse_args(argv)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file if os.path.exists(args.cfg_file)
                      else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), args.cfg_file))
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs)
    cfg.USE_GPU_NMS = True
    output_dir = os.path.join(args.save_dir, args.net, args.dataset)
    ground_ckpt = os.path.join(output_dir, 'vis_ground_{}_{}_{}.pth'.format(args.checksession, args.checkepoch, args.checkbatch))
    detector_ckpt = os.path.join(args.load_dir, 'faster_rcnn_gnome.pth')
    synthetic_data = not os.path.isdir(os.path.join(args.root, args.dataset))
    if args.resume:
        load = ('resume', ground_ckpt)
    elif args.phase in ('val', 'test'):
        load = ('eval', ground_ckpt)
    elif os.path.exists(detector_ckpt):
        load = ('detector', detector_ckpt)
    elif synthetic_data:
        load = ('synthetic-detector', None)
    else:
        load = ('detector', detector_ckpt)          # (missing: main() raises like model.py:1032-1033)
    return dict(args=args, output_dir=output_dir, ground_ckpt=ground_ckpt, detector_ckpt=detector_ckpt, load=load,
                synthetic_data=synthetic_data, summary_path=os.path.join('runs', 'sess_{}_{}'.format(args.checksession, args.statement)))


def synthetic_loader(args, phase, n_batches, seed0):
    """`n_batches` loader tuples of the reference's collate format for one epoch (different content per batch)."""
    from .train import combine_batches_synthetic
    Na = args.batch_size if phase == 'train' else args.batch_size_val
    Ns = args.sample_num if phase == 'train' else max(args.sample_num, 6)
    for b in range(n_batches):
        yield combine_batches_synthetic(Na, Ns, args.max_ent_len, H=args.img_h, W=args.img_w, seed=seed0 + b, vocab=SYN_VOCAB)


def main(argv=None, loaders=None, glove=None, synthetic_batches=None, log=print):
    """Returns the best validation accuracy (synthetic data: minus the best mean validation loss, there is no ground truth)."""
    from . import synthetic as syn
    from .checkpoint import (adjust_learning_rate, load_detector_checkpoint, load_ground_checkpoint, save_ground_checkpoint)
    from .model import GroundModel
    from .parallel import FusedClipAdam, GradAllReducer, broadcast_parameters, trainable_parameters
    from .train import train_epoch, validate_epoch
    p = plan(argv)
    args = p['args']
    log('Called with args:')
    log(args)
    if not (args.cuda and torch.cuda.is_available()):
        raise SystemExit("nafae_amd.run: needs --cuda and a visible GPU -- the HIP path has no CPU fallback "
                         "(torch.cuda.is_available() = %s, --cuda = %s)" % (torch.cuda.is_available(), bool(args.cuda)))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    device = torch.device('cuda', torch.cuda.current_device())
    np.random.seed(cfg.RNG_SEED)
    n_syn = int(synthetic_batches if synthetic_batches is not None else os.environ.get('NAFAE_SYNTHETIC_BATCHES', '4'))
    if p['synthetic_data'] and loaders is None:
        log('[nafae_amd.run] %s not found: SYNTHETIC loader tuples (%d per epoch) and a synthetic GloVe table stand in for the '
            'data set' % (os.path.join(args.root, args.dataset), n_syn))
    glove = glove if glove is not None else SyntheticGloVe(dim=args.glove_dim, seed=cfg.RNG_SEED)
    log('load {} word'.format(len(glove.itos)))
    if rank == 0 and not os.path.exists(p['output_dir']):
        os.makedirs(p['output_dir'])
    ground_model = GroundModel(args, cfg)
    start_epoch = 0
    kind, path = p['load']
    if kind == 'resume':
        log("resume checkpoint %s" % path)
        start_epoch = load_ground_checkpoint(ground_model, path, resume=True)
    elif kind == 'eval':
        log("load checkpoint %s" % path)
        start_epoch = load_ground_checkpoint(ground_model, path, resume=False)
    elif kind == 'detector':
        if not os.path.exists(args.load_dir):
            raise Exception('There is no input directory for loading network from ' + args.load_dir)      # model.py:1032-1033
        log("load checkpoint %s" % path)
        load_detector_checkpoint(ground_model, path)
    else:
        log('[nafae_amd.run] %s not found: seeded synthetic detector weights' % p['detector_ckpt'])
        ground_model.fasterRCNN.load_state_dict(syn.detector_state(seed=cfg.RNG_SEED, heads=False), strict=False)
    log('load model successfully!')
    ground_model.to(device)
    if world > 1:
        broadcast_parameters(ground_model, src=0)
    criterion = torch.nn.L1Loss()
    reducer = GradAllReducer(trainable_parameters(ground_model))
    optimizer = FusedClipAdam(reducer, lr=args.lr, weight_decay=args.weight_decay, max_norm=args.clip)
    optimizer.set_reference_layout(ground_model)

    def loader(phase, epoch):
        if loaders is not None:
            return loaders[phase]() if callable(loaders[phase]) else loaders[phase]
        return synthetic_loader(args, phase, n_syn, seed0=1000 * (epoch + 1) + 17 * rank + {'train': 0, 'val': 500, 'test': 700}[phase])

    def validate(phase, epoch):
        acc, vloss, dets = validate_epoch(loader(phase, epoch), ground_model, glove, args, device=device)
        ground_model.train()
        ground_model.DVSA.init_train()
        ground_model.fasterRCNN.eval()
        log('[%s] epoch %d: mean loss %.5f, accuracy %s' % (phase, epoch, vloss, 'n/a (no ground truth)' if acc is None else '%.4f' % acc))
        return acc if acc is not None else -vloss

    best_accuracy = -float('inf') if p['synthetic_data'] else 0
    for epoch in range(start_epoch, args.epoch):
        if args.phase == 'train':
            ground_model.train()                                     # model.py:669-673
            ground_model.DVSA.init_train()
            ground_model.fasterRCNN.eval()
            mean_loss, steps = train_epoch(loader('train', epoch), ground_model, glove, criterion, optimizer, reducer, args,
                                           device=device)
            log('[train] epoch %d: %d steps, mean loss %.5f, lr %g' % (epoch, steps, mean_loss, optimizer.param_groups[0]['lr']))
            adjust_learning_rate(optimizer, args.lr, epoch, args.lr_decay_gamma, args.lr_decay_step)
            if (epoch + 1) % args.eval_freq == 0 or epoch == args.epoch - 1:
                accuracy = validate('val', epoch)
                is_best = accuracy > best_accuracy
                best_accuracy = max(accuracy, best_accuracy)
                if is_best and rank == 0:
                    save_name = os.path.join(p['output_dir'], 'vis_ground_{}_{}_{}.pth'.format(args.checksession, epoch, args.checkbatch))
                    save_ground_checkpoint(ground_model, optimizer, args.checksession, epoch, save_name)
                    log('saved %s' % save_name)
        elif args.phase == 'val':
            best_accuracy = validate('val', epoch)
            break
        elif args.phase == 'test':
            best_accuracy = validate('test', epoch)
            break
        else:
            raise SystemExit("nafae_amd.run: --phase must be train, val or test (detvis is the reference's visualiser: out of scope)")
    if rank == 0:
        if not os.path.exists('.optm'):
            os.makedirs('.optm')
        with open('.optm/{}.best'.format('model'), 'w') as f:        # model.py:1136-1141 (python_file = 'model')
            f.write('{}'.format(best_accuracy))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return best_accuracy


if __name__ == '__main__':
    main(sys.argv[1:])
