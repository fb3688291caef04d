"""Worker of tests/test_gpu_exact_dp.py: one rank of a frame-sharded ("exact global batch") training run.
usage: RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/exact_dp_worker.py OUT.pt STEPS
All ranks share GPU 0 (the test box has one GPU), so the process group is gloo and nafae_amd.parallel stages the
collectives through host memory; with one GPU per rank the same code runs on nccl (= RCCL)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nafae_amd.config import cfg_from_file, cfg_from_list, reset_cfg
    from nafae_amd.model import default_args
    from nafae_amd.train import make_batch, setup_training, shard_frames, train_step_exact
    reset_cfg()
    cfg_from_file(os.path.join(ROOT, "cfgs", "vgg16.yml"))
    cfg_from_list(["TEST.RPN_POST_NMS_TOP_N", "32"])
    Na, Ns, Ne = 4, 4, 8
    args = default_args(batch_size=Na, sample_num=Ns, max_ent_len=Ne, dropout_rate=0.0, Delta=10.0, vis_lam=4.13)
    model, opt, crit, red = setup_training(args, seed=21, distributed=True)
    model.fasterRCNN.conv_stream_k = False      # detector bit-identical for any number of frames per rank (see the test)
    losses = []
    for k in range(steps):
        gb = make_batch(Na, Ns, Ne, seed=100 + k, lens=[3, 0, 8, 5])
        lb = shard_frames(gb, rank, world)
        loss, D, D_sim, rois = train_step_exact(model, opt, crit, lb, args, red)
        losses.append(float(loss))
        if k == 0:
            grads0, norm0 = red.flat.clone().cpu(), float(opt.total_norm)     # (clipped) global gradient of step 0
    torch.cuda.synchronize()
    torch.save({"losses": losses, "params": opt.flat_params.cpu(), "grads0": grads0, "norm0": norm0, "D": D.cpu(), "D_sim": D_sim.cpu(),
                "bn_mean": model.word_ebd.bn.running_mean.cpu()}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
